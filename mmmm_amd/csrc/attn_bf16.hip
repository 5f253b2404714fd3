// Variable-length (block-diagonal) flash attention for gfx950, bf16 MFMA + fp32 softmax.
// Replaces the xformers memory_efficient_attention call sites of the reference:
//   modeling_cogvlm.py:113-128  BlockDiagonalCausalMask, head_dim 128 (LM)
//   visual.py:91-99             BlockDiagonalMask, head_dim 112 (EVA ViT)
//
// FORWARD: attn32_fwd.hpp — 8 waves x 32 queries on v_mfma_f32_32x32x16_bf16, four barrier-separated clusters per 64-key tile with
// the two waves of every SIMD one cluster apart (guide T16); its header has the structure and the measurements behind it.
// BACKWARD (this file): delta + two kernels, no atomics — dQ is query-stationary, dK/dV key-stationary; a wave owns 16 queries / keys
// on v_mfma_f32_16x16x32_bf16 (~110-164 VGPRs). Scores are computed TRANSPOSED (guide §3 "An accumulator tile as the next MFMA's
// operand"), S^T[kv][q] = K Q^T, so the query index sits on the lane and two stacked accumulator tiles, converted to bf16, are
// directly the B operand of the next product; the transposed operands come from the row-major tiles through ds_read_b64_tr_b16.
#include <mutex>
#include <type_traits>
#include <utility>
#include "vm_common.hpp"
#include "vm_tile.hpp"

extern "C" int vm_prof_begin_(int kind, void* stream, void** tok);
extern "C" int vm_prof_end_(int kind, void* stream, void* tok, double flops);

namespace {

constexpr float NEG_BIG = -1.0e30f;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

// registers 8s..8s+7 of a 32x32 accumulator -> bf16 operand fragment
__device__ __forceinline__ bf16x8_t pack8(const f32x16_t& a, int s) {
  u16x8_t r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = f2bf(a[8 * s + j]);
  return __builtin_bit_cast(bf16x8_t, r);
}

// kv/q index (inside a 32-row accumulator tile) of register r for lane half h
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

struct AttnP {
  const unsigned short* q; const unsigned short* k; const unsigned short* v; unsigned short* out;
  int64_t ldq, ldk, ldv, ldo;
  float* lse;
  const int32_t* cu; const int32_t* row_of_pos;
  int total_pos_max, n_heads;
  int n_seq, n_tiles;      // sequences, workgroup tiles per sequence (1-D XCD-aware grid)
  int q_block;             // 32-wide kernels: positions per workgroup tile (a multiple of 32: the sequence split evenly)
  float scale; int causal;
  const unsigned short* dout; int64_t lddo;
  unsigned short* dq; unsigned short* dk; unsigned short* dv; int64_t lddq, lddk, lddv;
  float* delta;
  unsigned short* ds; int ds_pitch;   // backward with a workspace: dS^T [head][position of the key][query position, pitch ds_pitch] bf16, or null
  unsigned long long* dbg;   // diagnostic builds only (-DA32_STAMPS)
};

__device__ __forceinline__ int phys_row(const AttnP& p, int gpos) {
  return p.row_of_pos ? p.row_of_pos[gpos] : gpos;
}

// ----------------------------------------------------------------------------- delta = rowsum(dO * O)
template <int HD>
__global__ __launch_bounds__(256) void attn_delta_k(const AttnP p, int n_seq) {
  // one wave per position; a lane owns one 8-element chunk (16-byte loads) and a pass covers as many WHOLE heads as fit
  // in 64 lanes, so the per-head sum is a gather over HD/8 neighbouring lanes
  constexpr int CPH = HD / 8;                       // chunks per head
  constexpr int HPP = 64 / CPH;                     // heads per pass
  const int lane = threadIdx.x & 63;
  const int gpos = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int total = p.cu[n_seq];
  if (gpos >= total) return;
  const int64_t r = phys_row(p, gpos);
  const unsigned short* dop = p.dout + r * p.lddo;
  const unsigned short* op = p.out + r * p.ldo;
  const int hl = lane / CPH;                        // head within the pass (lanes >= HPP*CPH idle)
  const int first = hl * CPH;
  for (int h0 = 0; h0 < p.n_heads; h0 += HPP) {
    const int head = h0 + hl;
    const bool live = hl < HPP && head < p.n_heads;
    float acc = 0.f;
    if (live) {
      const int off = head * HD + (lane - first) * 8;
      const u16x8_t a = *reinterpret_cast<const u16x8_t*>(dop + off);
      const u16x8_t b = *reinterpret_cast<const u16x8_t*>(op + off);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += bf2f(a[e]) * bf2f(b[e]);
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < CPH; ++j) sum += __shfl(acc, (first + j) & 63, 64);
    if (live && lane == first) p.delta[(int64_t)head * p.total_pos_max + gpos] = sum;
  }
}

// =====================================================================================================================
// A wave owns 16 queries (forward, dQ) or 16 keys (dK/dV) and works with v_mfma_f32_16x16x32_bf16: half the register
// footprint of a 32-wide design (the first generation of these kernels: 4-8 waves per CU, 2.3x slower), 16 waves per CU,
// and the MFMA, VALU (softmax) and LDS phases of different waves overlap by thread-level parallelism.
// Operand tiles are staged by LDS-DMA (buffer_load ... lds, swizzle applied to the source chunk) into a 2-stage ring,
// one barrier per 64-position tile; the tile after next's physical rows (packed-layout indirection) are fetched one
// iteration ahead. Score tiles are kept transposed (key on the MFMA row, query on the lane) exactly as above.
constexpr int OOB_OFF = 0x7FFFFFF0;
constexpr int A16_STAGE = 2 * 64 * ROWB;         // two 64-row operand tiles per stage
constexpr int A16_LDS = 2 * A16_STAGE;
constexpr int A16_LDS_DKV = A16_LDS + 2 * 2 * 64 * 4;   // + [stage][lse|delta][64] floats

typedef __attribute__((address_space(3))) void* lds_vptr_t;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t whole_rsrc(const void* base) {
  const uint64_t a = (uint64_t)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, OOB_OFF, 0x00020000);
}

// Chunk swizzle of the 256-byte-row tiles read by the 16-wide kernels: slot = chunk ^ ((row & 7) << 1).
// Conflict-free for BOTH access patterns (bank rule: 16-byte slot s of a row occupies banks 4s..4s+3 of the 256-B bank row):
//  * ds_read_b128 row fragments, lane (ln, g) -> (row 16j + ln, chunk 4s + g): inside each of the instruction's four
//    16-lane groups the slots (4s + g) ^ 2(ln & 7) are 16 distinct values (the groups pair g = 0 with 1 and 2 with 3);
//  * ds_read_b64_tr_b16, whose 32-lane half touches rows rbase + 0..7 (or 8..15), 32 contiguous bytes each at slot pair
//    (2b ^ 2(row & 7)): eight distinct pairs.
// (The 32-wide image swz() is 2-way conflicted under both of these patterns: measured SQ_LDS_BANK_CONFLICT 2.3x the
// LDS instruction cycles.)
__device__ __forceinline__ int swz16(int row) { return (row & 7) << 1; }
__device__ __forceinline__ int tile_off16(int row, int chunk) { return row * ROWB + ((chunk ^ swz16(row)) << 4); }

// A/B operand of a 16x16x32 MFMA read along rows: lane -> row r0 + (lane & 15), k = 32 s + 8 (lane >> 4) + 0..7
__device__ __forceinline__ bf16x8_t frag16_row(const char* tile, int r0, int s, int lane) {
  return *reinterpret_cast<const bf16x8_t*>(tile + tile_off16(r0 + (lane & 15), 4 * s + (lane >> 4)));
}
// transposed operand: lane (i = lane & 15, g = lane >> 4) gets tile[rbase + 4 g + 0..3][16 b + i] and
// tile[rbase + 16 + 4 g + 0..3][16 b + i] — the k order in which two stacked 16x16 accumulator tiles
// (registers 0..3 of rows 4g.. and of rows 16 + 4g..) appear as the other operand.
__device__ __forceinline__ bf16x8_t frag16_tr(const char* tile, int rbase, int b, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const int rowA = rbase + 4 * g + (i >> 2), rowB = rowA + 16;
  const int chunk = 2 * b + ((i & 3) >> 1);
  const int sub = (i & 1) << 3;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(tile + tile_off16(rowA, chunk) + sub));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(tile + tile_off16(rowB, chunk) + sub));
  return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
// frag16_tr through inline assembly. hipcc cannot see what a ds_read_b64_tr_b16 BUILTIN reads: it drains the LDS-DMA queue (vmcnt(0)) in front
// of the first one after a DMA issue — i.e. in the middle of every tile, under half of the DMA's flight — and waits for each pair of
// reads on its own (lgkmcnt(0) per MFMA pair, nothing in flight behind it). The assembly form is invisible to it: a whole batch is
// issued, lands under the exponentials, and is complete only behind tr_wait() (a wait-only statement + sched_barrier: guide §5.7 (iii)).
// tr_lane_base(): the lane's address for column block b of tile rows 4 g + (i >> 2) (+ 16: immediate 16 ROWB); 32-row groups, the
// second operand tile and the stage are added by the caller (lane-uniform) or ride as immediates.
// Column block b sits at bits 5..7 of the address as b ^ (row & 7): one lane base (block 0), "^ (b << 5)" per block (the stage base is
// 256-byte aligned and every other term leaves those bits alone).
__device__ __forceinline__ unsigned tr_lane_base(const char* smem, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const int rowA = 4 * g + (i >> 2);
  return (unsigned)(size_t)smem + rowA * ROWB + ((rowA & 7) << 5) + (((i & 3) >> 1) << 4) + ((i & 1) << 3);
}
template <int IMM>
__device__ __forceinline__ bf16x8_t tr_asm(unsigned addr) {
  s16x4_t lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "i"(IMM));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "i"(IMM + 16 * ROWB));
  return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ void tr_wait() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ bf16x8_t pack2(const f32x4_t& a, const f32x4_t& b) {
  u16x8_t r = {f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3]), f2bf(b[0]), f2bf(b[1]), f2bf(b[2]), f2bf(b[3])};
  return __builtin_bit_cast(bf16x8_t, r);
}
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // x <= 0 here; denormal results flush to 0

// physical rows of the two 4-row groups this lane stages for the 64-position tile starting at pos0
// NW = waves per workgroup (8: 128 positions per block, 16: 256): a 64-row tile is 16 wave-instructions of 4 rows, 16 / NW per wave
template <int NW>
__device__ __forceinline__ void a16_rows(const AttnP& p, int seq0, int seqlen, int pos0, int wave, int lane, int (&pr)[2]) {
  constexpr int PW = 16 / NW;
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int pos = pos0 + (wave * PW + i) * 4 + (lane >> 4);
    pr[i] = pos < seqlen ? phys_row(p, seq0 + pos) : 0;
  }
}
// LDS-DMA of one 64-row tile of a [pos][H*HD] operand (8 waves x 2 wave-instructions of 4 rows x 256 B).
// The lane's tile row and source column are loop invariants (StageLane); per tile only the row's physical index changes.
struct StageLane {
  int row[2];    // tile row of wave-instruction i
  int col[2];    // byte offset of the lane's source chunk inside a row of the operand, or -1 past the head dimension
  template <int HD, int NW>
  __device__ __forceinline__ void init(int head, int wave, int lane) {
    constexpr int PW = 16 / NW;
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      row[i] = (wave * PW + i) * 4 + (lane >> 4);
      const int chunk = (lane & 15) ^ swz16(row[i]);
      col[i] = chunk * 8 < HD ? (head * HD + chunk * 8) * 2 : -1;
    }
  }
};
template <int NW>
__device__ __forceinline__ void a16_stage(__amdgpu_buffer_rsrc_t rs, int ld_b, const StageLane& sl, int rows_left, const int (&pr)[2],
                                          char* tile, int wave) {
  constexpr int PW = 16 / NW;
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const bool valid = sl.row[i] < rows_left && sl.col[i] >= 0;
    const int voff = valid ? (int)__umul24(pr[i], ld_b) + sl.col[i] : OOB_OFF;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr_t)(tile + (wave * PW + i) * 1024), 16, voff, 0, 0, 0);
  }
}

#define A16_WAIT_ALL() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

// 1-D grid -> (tile, head, sequence). Workgroups b and b + 8 run on the same XCD (one L2 each), so all tiles of one
// (head, sequence) pair — which stream the same K/V (or Q/dO) rows — are given ids of one residue mod 8 and adjacent
// slots: the pair's operands are fetched into that XCD's L2 once instead of once per tile. (With the plain 3-D grid the
// 7 query tiles of a ViT head land on 7 different XCDs and the kernel is bound by the resulting 7x operand traffic.)
__device__ __forceinline__ bool a16_block(const AttnP& p, int& tile, int& head, int& seq) {
  const int id = blockIdx.x;
  const int xcd = id & 7, slot = id >> 3;
  tile = slot % p.n_tiles;
  const int hs = (slot / p.n_tiles) * 8 + xcd;
  if (hs >= p.n_heads * p.n_seq) return false;
  head = hs % p.n_heads;
  seq = hs / p.n_heads;
  return true;
}

}  // namespace
#include "attn32_fwd.hpp"
namespace {

// ----------------------------------------------------------------------------- backward: dQ (16 queries per wave)
template <int HD, int NW, int TRV>
__global__ __launch_bounds__(NW * 64, 4) void attn16_dq_k(const AttnP p) {
  constexpr int QB = NW * 16;
  constexpr int KS = (HD + 31) / 32;
  constexpr int ND = HD / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 15, g = lane >> 4;
  int tile_, head, seq;
  if (!a16_block(p, tile_, head, seq)) return;
  const int seq0 = p.cu[seq];
  const int seqlen = p.cu[seq + 1] - seq0;
  const int q0 = tile_ * QB;
  if (q0 >= seqlen) return;
  const int qpos = q0 + wave * 16 + ln;
  const bool qvalid = qpos < seqlen;
  const bool wave_on = q0 + wave * 16 < seqlen;        // a wave whose 16 queries all lie past the sequence only stages and meets the barriers
  const int64_t qrow = qvalid ? phys_row(p, seq0 + qpos) : 0;

  bf16x8_t qf[KS], dof[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    i32x4_t a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
    if (qvalid && 32 * s + 8 * g < HD) {
      a = *reinterpret_cast<const i32x4_t*>(p.q + qrow * p.ldq + head * HD + 32 * s + 8 * g);
      b = *reinterpret_cast<const i32x4_t*>(p.dout + qrow * p.lddo + head * HD + 32 * s + 8 * g);
    }
    qf[s] = __builtin_bit_cast(bf16x8_t, a);
    dof[s] = __builtin_bit_cast(bf16x8_t, b);
  }
  // an invalid query gets lse = +inf-like so that its probabilities are exactly zero
  const float lse2 = qvalid ? p.lse[(int64_t)head * p.total_pos_max + seq0 + qpos] * LOG2E : 1.0e30f;
  const float dlt = qvalid ? p.delta[(int64_t)head * p.total_pos_max + seq0 + qpos] : 0.f;
  const float sc = p.scale * LOG2E;
  f32x4_t dq[ND];
#pragma unroll
  for (int b = 0; b < ND; ++b) dq[b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int kv_end = p.causal ? min(seqlen, q0 + QB) : seqlen;
  const int nt = (kv_end + 63) / 64;
  const __amdgpu_buffer_rsrc_t rK = whole_rsrc(p.k), rV = whole_rsrc(p.v);
  const int ldk_b = (int)p.ldk * 2, ldv_b = (int)p.ldv * 2;
  StageLane sl;
  sl.init<HD, NW>(head, wave, lane);
  const unsigned trb = tr_lane_base(smem, lane);
  int pr[2];
  a16_rows<NW>(p, seq0, seqlen, 0, wave, lane, pr);
  a16_stage<NW>(rK, ldk_b, sl, seqlen - (0), pr, smem, wave);
  a16_stage<NW>(rV, ldv_b, sl, seqlen - (0), pr, smem + 64 * ROWB, wave);
  if (nt > 1) a16_rows<NW>(p, seq0, seqlen, 64, wave, lane, pr);
  A16_WAIT_ALL();
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    if (t + 1 < nt) {
      char* nb = smem + (buf ^ 1) * A16_STAGE;
      a16_stage<NW>(rK, ldk_b, sl, seqlen - ((t + 1) * 64), pr, nb, wave);
      a16_stage<NW>(rV, ldv_b, sl, seqlen - ((t + 1) * 64), pr, nb + 64 * ROWB, wave);
      if (t + 2 < nt) a16_rows<NW>(p, seq0, seqlen, (t + 2) * 64, wave, lane, pr);
    }
    const char* sK = smem + buf * A16_STAGE;
    const char* sV = sK + 64 * ROWB;
    const int kv0 = t * 64;
    const bool edge = kv0 + 64 > seqlen || (p.causal && kv0 + 64 > q0 + wave * 16);
    const int lim = p.causal ? min(qpos, seqlen - 1) : seqlen - 1;
    if (!wave_on) {
    } else if constexpr (TRV != 0) {
      // per 32-key half: S^T / dP^T (16 MFMAs), the half's transposed K fragments issued under the exponentials, dQ (ND MFMAs)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        f32x4_t sa[2], dp[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
          const int j = 2 * c + jj;
          sa[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag16_row(sK, 16 * j, 0, lane), qf[0], (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          dp[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag16_row(sV, 16 * j, 0, lane), dof[0], (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
          for (int s = 1; s < KS; ++s) {
            sa[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag16_row(sK, 16 * j, s, lane), qf[s], sa[jj], 0, 0, 0);
            dp[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag16_row(sV, 16 * j, s, lane), dof[s], dp[jj], 0, 0, 0);
          }
        }
        // (two batches: the 128-VGPR budget of 16 waves per CU has no room for ND fragments beside Q, dO and dQ)
        constexpr int NB0 = ND < 4 ? ND : 4;
        bf16x8_t tk[NB0], tk1[ND - NB0 + 1];
        const unsigned ad = trb + buf * A16_STAGE + c * 32 * ROWB;
#pragma unroll
        for (int b = 0; b < NB0; ++b) tk[b] = tr_asm<0>(ad ^ (b << 5));
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float pr_ = fast_exp2(__builtin_fmaf(sa[jj][r], sc, -lse2));
            if (edge) pr_ = (kv0 + 32 * c + 16 * jj + 4 * g + r <= lim) ? pr_ : 0.f;
            sa[jj][r] = pr_ * (dp[jj][r] - dlt);
          }
        const bf16x8_t df = pack2(sa[0], sa[1]);
        tr_wait();
#pragma unroll
        for (int b = NB0; b < ND; ++b) tk1[b - NB0] = tr_asm<0>(ad ^ (b << 5));
#pragma unroll
        for (int b = 0; b < NB0; ++b) dq[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tk[b], df, dq[b], 0, 0, 0);
        if constexpr (ND > NB0) {
          tr_wait();
#pragma unroll
          for (int b = NB0; b < ND; ++b) dq[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tk1[b - NB0], df, dq[b], 0, 0, 0);
        }
      }
    } else {
    f32x4_t sa[4], dp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sa[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag16_row(sK, 16 * j, 0, lane), qf[0], (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      dp[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag16_row(sV, 16 * j, 0, lane), dof[0], (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
      for (int s = 1; s < KS; ++s) {
        sa[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag16_row(sK, 16 * j, s, lane), qf[s], sa[j], 0, 0, 0);
        dp[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag16_row(sV, 16 * j, s, lane), dof[s], dp[j], 0, 0, 0);
      }
    }
    // dS^T = P ∘ (dP^T − delta)   (the softmax scale is applied once, on dQ)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float pr_ = fast_exp2(__builtin_fmaf(sa[j][r], sc, -lse2));
        if (edge) pr_ = (kv0 + 16 * j + 4 * g + r <= lim) ? pr_ : 0.f;
        sa[j][r] = pr_ * (dp[j][r] - dlt);
      }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const bf16x8_t df = pack2(sa[2 * c], sa[2 * c + 1]);
#pragma unroll
      for (int b = 0; b < ND; ++b)
        dq[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag16_tr(sK, 32 * c, b, lane), df, dq[b], 0, 0, 0);
    }
    }
    A16_WAIT_ALL();
    __syncthreads();
  }
  if (!qvalid) return;
  unsigned short* drow = p.dq + qrow * p.lddq + head * HD;
#pragma unroll
  for (int b = 0; b < ND; ++b) {
    const u16x4_t w = {f2bf(dq[b][0] * p.scale), f2bf(dq[b][1] * p.scale), f2bf(dq[b][2] * p.scale), f2bf(dq[b][3] * p.scale)};
    *reinterpret_cast<u16x4_t*>(drow + 16 * b + 4 * g) = w;
  }
}

// ----------------------------------------------------------------------------- backward: dQ = dS K from the stored dS^T (16 queries per wave)
// Same blocks, staging and epilogue as attn16_dq_k; per 64-key tile the K tile and the [64 keys x 128 query positions] tile of dS^T come in by
// LDS-DMA and a wave's work is 2 (ND + 1) transposed fragments and 2 ND MFMAs (the recomputing kernel: 32 + 2 ND MFMAs, 32 row fragments, the
// exponentials). The transposed read of the dS^T tile at the wave's 16-query column block IS the operand the recomputing kernel packs in registers.
template <int HD, int NW>
__global__ __launch_bounds__(NW * 64, 4) void attn16_dq_ds_k(const AttnP p) {
  constexpr int QB = NW * 16;
  constexpr int ND = HD / 16;
  constexpr int PW = 16 / NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 15, g = lane >> 4;
  int tile_, head, seq;
  if (!a16_block(p, tile_, head, seq)) return;
  const int seq0 = p.cu[seq];
  const int seqlen = p.cu[seq + 1] - seq0;
  const int q0 = tile_ * QB;
  if (q0 >= seqlen) return;
  const int qpos = q0 + wave * 16 + ln;
  const bool qvalid = qpos < seqlen;
  const bool wave_on = q0 + wave * 16 < seqlen;
  const int64_t qrow = qvalid ? phys_row(p, seq0 + qpos) : 0;
  f32x4_t dq[ND];
#pragma unroll
  for (int b = 0; b < ND; ++b) dq[b] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int kv_end = p.causal ? min(seqlen, q0 + QB) : seqlen;
  const int nt = (kv_end + 63) / 64;
  const __amdgpu_buffer_rsrc_t rK = whole_rsrc(p.k);
  const __amdgpu_buffer_rsrc_t rS = whole_rsrc(p.ds + ((int64_t)head * p.total_pos_max + seq0) * p.ds_pitch);
  const int ldk_b = (int)p.ldk * 2, lds_b = p.ds_pitch * 2;
  StageLane sl;
  sl.init<HD, NW>(head, wave, lane);
  int srow[PW], scol[PW];                        // dS^T tile: the lane's tile row and its source byte column (query positions q0 ..)
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    srow[i] = (wave * PW + i) * 4 + (lane >> 4);
    scol[i] = q0 * 2 + (((lane & 15) ^ swz16(srow[i])) << 4);
  }
  auto stage_ds = [&](int kv0, char* tile) {
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      const int key = kv0 + srow[i];
      const int voff = key < seqlen ? (int)__umul24(key, lds_b) + scol[i] : OOB_OFF;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rS, (lds_vptr_t)(tile + (wave * PW + i) * 1024), 16, voff, 0, 0, 0);
    }
  };
  const unsigned trb = tr_lane_base(smem, lane);
  int pr[2];
  a16_rows<NW>(p, seq0, seqlen, 0, wave, lane, pr);
  a16_stage<NW>(rK, ldk_b, sl, seqlen - (0), pr, smem, wave);
  stage_ds(0, smem + 64 * ROWB);
  if (nt > 1) a16_rows<NW>(p, seq0, seqlen, 64, wave, lane, pr);
  A16_WAIT_ALL();
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    if (t + 1 < nt) {
      char* nb = smem + (buf ^ 1) * A16_STAGE;
      a16_stage<NW>(rK, ldk_b, sl, seqlen - ((t + 1) * 64), pr, nb, wave);
      stage_ds((t + 1) * 64, nb + 64 * ROWB);
      if (t + 2 < nt) a16_rows<NW>(p, seq0, seqlen, (t + 2) * 64, wave, lane, pr);
    }
    if (wave_on) {
      bf16x8_t tk[2][ND], td[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const unsigned ad = trb + buf * A16_STAGE + c * 32 * ROWB;
        td[c] = tr_asm<64 * ROWB>(ad ^ (wave << 5));
#pragma unroll
        for (int b = 0; b < ND; ++b) tk[c][b] = tr_asm<0>(ad ^ (b << 5));
      }
      tr_wait();
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int b = 0; b < ND; ++b) dq[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tk[c][b], td[c], dq[b], 0, 0, 0);
    }
    A16_WAIT_ALL();
    __syncthreads();
  }
  if (!qvalid) return;
  unsigned short* drow = p.dq + qrow * p.lddq + head * HD;
#pragma unroll
  for (int b = 0; b < ND; ++b) {
    const u16x4_t w = {f2bf(dq[b][0] * p.scale), f2bf(dq[b][1] * p.scale), f2bf(dq[b][2] * p.scale), f2bf(dq[b][3] * p.scale)};
    *reinterpret_cast<u16x4_t*>(drow + 16 * b + 4 * g) = w;
  }
}

// ----------------------------------------------------------------------------- backward: dK, dV (16 keys per wave)
// DS: the kernel also writes dS^T = P^T ∘ (dP^T − delta) (bf16, exactly the operand of its own dK product) to the workspace, one row of
// query positions per key: dQ = dS K is then a plain product over it (attn16_dq_ds_k) instead of a second recomputation of S and dP.
template <int HD, int NW, int TRV, bool DS = false>
__global__ __launch_bounds__(NW * 64, 1) void attn16_dkv_k(const AttnP p) {
  constexpr int QB = NW * 16;
  constexpr int KS = (HD + 31) / 32;
  constexpr int ND = HD / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* stats = reinterpret_cast<float*>(smem + A16_LDS);     // [stage][lse | delta][64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ln = lane & 15, g = lane >> 4;
  int tile_, head, seq;
  if (!a16_block(p, tile_, head, seq)) return;
  const int seq0 = p.cu[seq];
  const int seqlen = p.cu[seq + 1] - seq0;
  const int k0 = tile_ * QB;
  if (k0 >= seqlen) return;
  const int kpos = k0 + wave * 16 + ln;
  const bool kvalid = kpos < seqlen;
  // a wave whose 16 keys all lie past the sequence (the last key block of a 785-key sequence holds 17 keys: six of its eight waves)
  // stages its share of the Q / dO tiles and meets the barriers, nothing else: its MFMAs and exponentials competed with the live
  // waves of the same SIMDs for nothing
  const bool wave_on = k0 + wave * 16 < seqlen;
  const int64_t krow = kvalid ? phys_row(p, seq0 + kpos) : 0;

  bf16x8_t kf[KS], vf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    i32x4_t a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
    if (kvalid && 32 * s + 8 * g < HD) {
      a = *reinterpret_cast<const i32x4_t*>(p.k + krow * p.ldk + head * HD + 32 * s + 8 * g);
      b = *reinterpret_cast<const i32x4_t*>(p.v + krow * p.ldv + head * HD + 32 * s + 8 * g);
    }
    kf[s] = __builtin_bit_cast(bf16x8_t, a);
    vf[s] = __builtin_bit_cast(bf16x8_t, b);
  }
  const float sc = p.scale * LOG2E;
  f32x4_t dk[ND], dv[ND];
#pragma unroll
  for (int b = 0; b < ND; ++b) { dk[b] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dv[b] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }

  const int q_begin = p.causal ? (k0 / 64) * 64 : 0;     // queries before the key block see none of it
  const int nt = (seqlen - q_begin + 63) / 64;
  const __amdgpu_buffer_rsrc_t rQ = whole_rsrc(p.q), rDO = whole_rsrc(p.dout);
  const __amdgpu_buffer_rsrc_t rL = whole_rsrc(p.lse), rD = whole_rsrc(p.delta);
  const int ldq_b = (int)p.ldq * 2, lddo_b = (int)p.lddo * 2;
  const int stat_base = (head * p.total_pos_max + seq0) * 4;
  auto stage_stats = [&](int qq0, int buf) {               // waves 0 / 1: 64 lse / delta values, zero past the sequence
    if (wave < 2) {
      const int qp = qq0 + lane;
      const int voff = qp < seqlen ? stat_base + qp * 4 : OOB_OFF;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wave == 0 ? rL : rD, (lds_vptr_t)(stats + buf * 128 + wave * 64), 4, voff, 0, 0, 0);
    }
  };
  StageLane sl;
  sl.init<HD, NW>(head, wave, lane);
  const unsigned trb = tr_lane_base(smem, lane);
  // dS^T rows of this (head, sequence): the lane's key row, 4 consecutive query positions per store (an invalid key stores out of bounds:
  // the store count per tile stays wave-uniform for the counted wait below)
  __amdgpu_buffer_rsrc_t rDS = rQ;
  int ds_off = OOB_OFF;
  if constexpr (DS) {
    rDS = whole_rsrc(p.ds + ((int64_t)head * p.total_pos_max + seq0) * p.ds_pitch);
    if (kvalid) ds_off = (kpos * p.ds_pitch + 4 * g) * 2;
  }
  int pr[2];
  a16_rows<NW>(p, seq0, seqlen, q_begin, wave, lane, pr);
  a16_stage<NW>(rQ, ldq_b, sl, seqlen - (q_begin), pr, smem, wave);
  a16_stage<NW>(rDO, lddo_b, sl, seqlen - (q_begin), pr, smem + 64 * ROWB, wave);
  stage_stats(q_begin, 0);
  if (nt > 1) a16_rows<NW>(p, seq0, seqlen, q_begin + 64, wave, lane, pr);
  A16_WAIT_ALL();
  __syncthreads();

#ifdef A32_STAMPS
  unsigned long long acc_[3] = {0, 0, 0}, last_ = a32::stamp();
#define A16_STAMP(i) { const unsigned long long t_ = a32::stamp(); acc_[i] += t_ - last_; last_ = t_; }
#else
#define A16_STAMP(i)
#endif
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    const int qq0 = q_begin + t * 64;
    if (t + 1 < nt) {
      char* nb = smem + (buf ^ 1) * A16_STAGE;
      a16_stage<NW>(rQ, ldq_b, sl, seqlen - (qq0 + 64), pr, nb, wave);
      a16_stage<NW>(rDO, lddo_b, sl, seqlen - (qq0 + 64), pr, nb + 64 * ROWB, wave);
      stage_stats(qq0 + 64, buf ^ 1);
      if (t + 2 < nt) a16_rows<NW>(p, seq0, seqlen, qq0 + 128, wave, lane, pr);
    }
    // (DS: the counted wait at the end of the tile assumes that NOTHING of this block — the next tile's DMA, the row-index loads — is
    // issued after the tile's dS^T stores; pinned, not left to the scheduler)
    if constexpr (DS) __builtin_amdgcn_sched_barrier(0);
    A16_STAMP(0)
    const char* sQ = smem + buf * A16_STAGE;
    const char* sDO = sQ + 64 * ROWB;
    const float* sL = stats + buf * 128;
    const float* sD = sL + 64;
    const bool edge = qq0 + 64 > seqlen || (p.causal && qq0 < k0 + QB);
    if (wave_on) {
#pragma unroll
    for (int hq = 0; hq < 2; ++hq) {
      // S[q][key], dP[q][key] for 32 queries: lane (key = ln, g) holds q = qq0 + 32 hq + 16 jj + 4 g + r
      f32x4_t sa[2], dp[2];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        sa[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag16_row(sQ, 32 * hq + 16 * jj, 0, lane), kf[0], (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        dp[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag16_row(sDO, 32 * hq + 16 * jj, 0, lane), vf[0], (f32x4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int s = 1; s < KS; ++s) {
          sa[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag16_row(sQ, 32 * hq + 16 * jj, s, lane), kf[s], sa[jj], 0, 0, 0);
          dp[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag16_row(sDO, 32 * hq + 16 * jj, s, lane), vf[s], dp[jj], 0, 0, 0);
        }
      }
      // the transposed dO / Q fragments of this half's dV / dK products: issued here, they land under the exponentials
      bf16x8_t tq[ND], tdo[ND];
      if constexpr (TRV != 0) {
        const unsigned ad = trb + buf * A16_STAGE + hq * 32 * ROWB;
#pragma unroll
        for (int b = 0; b < ND; ++b) {
          tq[b] = tr_asm<0>(ad ^ (b << 5));
          tdo[b] = tr_asm<64 * ROWB>(ad ^ (b << 5));
        }
      }
      f32x4_t pa[2];
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(sL + 32 * hq + 16 * jj + 4 * g);
        const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(sD + 32 * hq + 16 * jj + 4 * g);
        if (edge) {                                    // wave-uniform BRANCH around a second copy of the loop: as a flag inside one loop the compares and selects run on every tile
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float pr_ = fast_exp2(__builtin_fmaf(sa[jj][r], sc, -l4[r] * LOG2E));
            const int qp = qq0 + 32 * hq + 16 * jj + 4 * g + r;
            pr_ = (qp < seqlen && (!p.causal || kpos <= qp)) ? pr_ : 0.f;
            pa[jj][r] = pr_;
            sa[jj][r] = pr_ * (dp[jj][r] - d4[r]);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pr_ = fast_exp2(__builtin_fmaf(sa[jj][r], sc, -l4[r] * LOG2E));
            pa[jj][r] = pr_;
            sa[jj][r] = pr_ * (dp[jj][r] - d4[r]);
          }
        }
      }
      const bf16x8_t pf = pack2(pa[0], pa[1]);
      const bf16x8_t df = pack2(sa[0], sa[1]);
      if constexpr (DS) {
        typedef int i32x2_t __attribute__((ext_vector_type(2)));
        const i32x4_t d4 = __builtin_bit_cast(i32x4_t, df);
        const int o = kvalid ? ds_off + (qq0 + 32 * hq) * 2 : OOB_OFF;
        __builtin_amdgcn_raw_buffer_store_b64((i32x2_t){d4[0], d4[1]}, rDS, o, 0, 0);                       // q = qq0 + 32 hq + 4 g + 0..3
        __builtin_amdgcn_raw_buffer_store_b64((i32x2_t){d4[2], d4[3]}, rDS, kvalid ? o + 32 : OOB_OFF, 0, 0);   // ... + 16
      }
      if constexpr (TRV != 0) {
        tr_wait();
#pragma unroll
        for (int b = 0; b < ND; ++b) {
          dv[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tdo[b], pf, dv[b], 0, 0, 0);
          dk[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tq[b], df, dk[b], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int b = 0; b < ND; ++b) {
          dv[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag16_tr(sDO, 32 * hq, b, lane), pf, dv[b], 0, 0, 0);
          dk[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag16_tr(sQ, 32 * hq, b, lane), df, dk[b], 0, 0, 0);
        }
      }
    }
    }
    A16_STAMP(1)
    if constexpr (DS) {
      // the tile's four dS^T stores are this wave's youngest memory operations: everything older — the next tile's DMA — has landed at
      // vmcnt(4), and the stores keep flying (vmcnt counts stores on gfx950; __syncthreads() would drain them with its fence)
      constexpr int DS_STORES_PER_TILE = 2 /* query halves */ * 2 /* 8-byte stores per half */;
      static_assert(DS_STORES_PER_TILE == 4, "the wait below leaves exactly the tile's dS^T stores in flight: keep it equal to the stores issued above");
      if (wave_on) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(DS_STORES_PER_TILE) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");             // (a wave without keys issued no stores)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    } else {
      A16_WAIT_ALL();
      __syncthreads();
    }
    A16_STAMP(2)
  }
#ifdef A32_STAMPS
  if (p.dbg && lane == 0) {
    unsigned long long* d = p.dbg + ((size_t)blockIdx.x * NW + wave) * 4;
    d[0] = acc_[0]; d[1] = acc_[1]; d[2] = acc_[2]; d[3] = (unsigned long long)nt;
  }
#endif
  if (!kvalid) return;
  unsigned short* dkrow = p.dk + krow * p.lddk + head * HD;
  unsigned short* dvrow = p.dv + krow * p.lddv + head * HD;
#pragma unroll
  for (int b = 0; b < ND; ++b) {
    const u16x4_t wk = {f2bf(dk[b][0] * p.scale), f2bf(dk[b][1] * p.scale), f2bf(dk[b][2] * p.scale), f2bf(dk[b][3] * p.scale)};
    const u16x4_t wv = {f2bf(dv[b][0]), f2bf(dv[b][1]), f2bf(dv[b][2]), f2bf(dv[b][3])};
    *reinterpret_cast<u16x4_t*>(dkrow + 16 * b + 4 * g) = wk;
    *reinterpret_cast<u16x4_t*>(dvrow + 16 * b + 4 * g) = wv;
  }
}

AttnP to_params(const vm_attn_args* a) {
  AttnP p;
  p.q = (const unsigned short*)a->q; p.k = (const unsigned short*)a->k; p.v = (const unsigned short*)a->v;
  p.out = (unsigned short*)a->out;
  p.ldq = a->ldq; p.ldk = a->ldk; p.ldv = a->ldv; p.ldo = a->ldo;
  p.lse = a->lse; p.cu = a->cu_seqlens; p.row_of_pos = a->row_of_pos;
  p.total_pos_max = a->total_pos_max; p.n_heads = a->n_heads;
  p.n_seq = a->n_seq; p.n_tiles = (a->max_seqlen + 127) / 128;      // (the launchers overwrite n_tiles with their block size)
  p.scale = a->scale; p.causal = a->causal;
  p.dout = (const unsigned short*)a->dout; p.lddo = a->lddo;
  p.dq = (unsigned short*)a->dq; p.dk = (unsigned short*)a->dk; p.dv = (unsigned short*)a->dv;
  p.lddq = a->lddq; p.lddk = a->lddk; p.lddv = a->lddv;
  p.delta = a->delta;
  p.ds = nullptr; p.ds_pitch = 0;
  p.dbg = nullptr;
  return p;
}

bool args_ok(const vm_attn_args* a) {
  if (!a || !a->q || !a->k || !a->v || !a->out || !a->cu_seqlens || !a->lse) return false;
  if (a->n_seq <= 0 || a->n_heads <= 0 || a->max_seqlen <= 0 || a->total_pos_max <= 0) return false;
  if (a->ldq % 8 || a->ldk % 8 || a->ldv % 8 || a->ldo % 4) return false;
  return true;
}

// dynamic LDS above 64 KiB needs the attribute once per kernel
bool lds_ok(const void* fn, int bytes) {
  return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
}
// the LDS-DMA staging addresses operands with 32-bit byte offsets from the tensor base
bool fits32(const vm_attn_args* a) {
  const int64_t rows = 2 * (int64_t)a->total_pos_max;     // physical rows of the packed layout: positions + slack
  return rows * a->ldq * 2 < OOB_OFF && rows * a->ldk * 2 < OOB_OFF && rows * a->ldv * 2 < OOB_OFF &&
         (!a->dout || rows * a->lddo * 2 < OOB_OFF) && (int64_t)a->n_heads * a->total_pos_max * 4 < OOB_OFF;
}

dim3 grid16(const vm_attn_args* a, int qb) {
  const int pairs8 = (a->n_heads * a->n_seq + 7) / 8 * 8;
  return dim3((unsigned)(pairs8 * ((a->max_seqlen + qb - 1) / qb)));
}
// (Measured in round 3 and removed: 256-position blocks of 16 waves — the forward 83 vs 79 us, the step 314.3 vs 314.0 ms; dK / dV would spill
// 26-49 registers at 16 waves. The backward kernels use 128-position blocks of 8 waves.)
double attn_flops(const vm_attn_args* a, double mult) {
  // upper bound with every sequence at max_seqlen; bench uses equal-length sequences so it is exact
  const double L = a->max_seqlen;
  double f = mult * 2.0 * L * L * a->head_dim * a->n_heads * a->n_seq;
  return a->causal ? 0.5 * f : f;
}


#ifdef A32_STAMPS
unsigned long long* g_a32_dbg = nullptr;
#endif
// ---- forward launch
template <int HD>
int fwd_launch32(const vm_attn_args* a, hipStream_t st) {
  constexpr int NW = a32::NW;
  AttnP p = to_params(a);
#ifdef A32_STAMPS
  p.dbg = g_a32_dbg;
#endif
  p.n_tiles = (a->max_seqlen + NW * 32 - 1) / (NW * 32);
  // full 256-query workgroups and ONE short one per sequence (785 = 3 x 256 + 17) rather than an even split (4 x 224): the short
  // workgroup's single live wave has its SIMD to itself and is done in about half the time, and the dispatcher hands its CU the
  // next workgroup — 1.5 + 0.25 rounds of full workgroups instead of 2 on the 8 x 785 ViT-E shape
  p.q_block = NW * 32;
  // rings / epilogue slabs, plus the sequence's slice of row_of_pos when the layout is indirect
  const int lds = a32::RING_LDS + (a->row_of_pos ? (a->max_seqlen + 3) / 4 * 16 : 0);
  if (lds > 160 * 1024) return VM_ERR_UNSUPPORTED;
  const int pairs8 = (a->n_heads * a->n_seq + 7) / 8 * 8;
  const dim3 grid((unsigned)(pairs8 * p.n_tiles)), block(NW * 64);
  static std::once_flag once;
  static bool ok = false;
  std::call_once(once, [] {
    ok = lds_ok((const void*)a32::fwd_k<HD, false>, 160 * 1024) && lds_ok((const void*)a32::fwd_k<HD, true>, 160 * 1024);
  });
  if (!ok) return VM_ERR_LAUNCH;
  if (a->causal) hipLaunchKernelGGL((a32::fwd_k<HD, true>), grid, block, lds, st, p);
  else hipLaunchKernelGGL((a32::fwd_k<HD, false>), grid, block, lds, st, p);
  return VM_OK;
}

int fwd_default_variant(const vm_attn_args* a) { return 8; }

int fwd_launch(const vm_attn_args* a, hipStream_t st, int variant) {
  switch (a->head_dim) {
#define VM_A32_CASE(hd) case hd: return fwd_launch32<hd>(a, st);
    VM_A32_CASE(128) VM_A32_CASE(112) VM_A32_CASE(96) VM_A32_CASE(64) VM_A32_CASE(32) VM_A32_CASE(16)
#undef VM_A32_CASE
    default: return VM_ERR_UNSUPPORTED;
  }
}

// ---- backward launch: delta (which & 1), dQ (& 2), dK / dV (& 4); variant 0 = transposed reads through the builtin (the round-3
// kernels, kept for tools/ubench/attn_bench's A/B), 1 = batched through assembly (what ships)
// bytes of the dS^T workspace: [head][position][query position, pitch = max_seqlen rounded up to the 128-query block] bf16
int64_t bwd_ds_pitch(const vm_attn_args* a) { return ((int64_t)a->max_seqlen + 127) / 128 * 128; }
int64_t bwd_ds_bytes(const vm_attn_args* a) { return (int64_t)a->n_heads * a->total_pos_max * bwd_ds_pitch(a) * 2; }

template <int HD, int TRV>
int bwd_launch16(const vm_attn_args* a, hipStream_t st, int which) {
  AttnP p = to_params(a);
#ifdef A32_STAMPS
  p.dbg = g_a32_dbg;
#endif
  p.n_tiles = (a->max_seqlen + 127) / 128;
  static std::once_flag once;
  static bool ok = false;
  std::call_once(once, [] {
    ok = lds_ok((const void*)attn16_dq_k<HD, 8, TRV>, A16_LDS) && lds_ok((const void*)attn16_dkv_k<HD, 8, TRV>, A16_LDS_DKV) &&
         lds_ok((const void*)attn16_dkv_k<HD, 8, TRV, true>, A16_LDS_DKV) && lds_ok((const void*)attn16_dq_ds_k<HD, 8>, A16_LDS);
  });
  if (!ok) return VM_ERR_LAUNCH;
  const int64_t items = (int64_t)a->total_pos_max;     // delta: one wave per position
  if (which & 1) hipLaunchKernelGGL(attn_delta_k<HD>, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, p, a->n_seq);
  // With a workspace (TRV != 0; one sequence's rows must stay inside 32-bit byte offsets): dK / dV first — it leaves dS^T behind — then dQ as a
  // product over it. Without: dQ and dK / dV each recompute S and dP.
  const bool ds = TRV != 0 && a->workspace && a->workspace_bytes >= bwd_ds_bytes(a) && (int64_t)a->max_seqlen * bwd_ds_pitch(a) * 2 < OOB_OFF;
  if (ds) {
    p.ds = (unsigned short*)a->workspace; p.ds_pitch = (int)bwd_ds_pitch(a);
    if (which & 4) hipLaunchKernelGGL((attn16_dkv_k<HD, 8, TRV, true>), grid16(a, 128), dim3(512), A16_LDS_DKV, st, p);
    if (which & 2) hipLaunchKernelGGL((attn16_dq_ds_k<HD, 8>), grid16(a, 128), dim3(512), A16_LDS, st, p);
    return VM_OK;
  }
  if (which & 2) hipLaunchKernelGGL((attn16_dq_k<HD, 8, TRV>), grid16(a, 128), dim3(512), A16_LDS, st, p);
  // dK / dV: two accumulator sets, one 8-wave workgroup per CU
  if (which & 4) hipLaunchKernelGGL((attn16_dkv_k<HD, 8, TRV>), grid16(a, 128), dim3(512), A16_LDS_DKV, st, p);
  return VM_OK;
}
int bwd_launch(const vm_attn_args* a, hipStream_t st, int variant, int which) {
  switch (a->head_dim) {
#ifdef VM_ATTN_BENCH_BUILD
#define VM_A16_CASE(hd) case hd: return variant ? bwd_launch16<hd, 1>(a, st, which) : bwd_launch16<hd, 0>(a, st, which);
#else
#define VM_A16_CASE(hd) case hd: return bwd_launch16<hd, 1>(a, st, which);
#endif
    VM_A16_CASE(128) VM_A16_CASE(112) VM_A16_CASE(96) VM_A16_CASE(64) VM_A16_CASE(32) VM_A16_CASE(16)
#undef VM_A16_CASE
    default: return VM_ERR_UNSUPPORTED;
  }
}

}  // namespace

extern "C" {

int vm_attn_fwd_bf16(const vm_attn_args* a, void* stream) {
  if (!args_ok(a)) return VM_ERR_BAD_ARG;
  if (!fits32(a)) return VM_ERR_UNSUPPORTED;
  void* tok = nullptr;
  vm_prof_begin_(VM_PROF_ATTN, stream, &tok);
  const int rc = fwd_launch(a, (hipStream_t)stream, fwd_default_variant(a));
  vm_prof_end_(VM_PROF_ATTN, stream, tok, attn_flops(a, 2.0));
  if (rc != VM_OK) return rc;
  VM_LAUNCH_CHECK();
  return VM_OK;
}

int vm_attn_bwd_workspace_bytes(const vm_attn_args* a, int64_t* bytes) {
  if (!a || !bytes || a->n_heads <= 0 || a->total_pos_max <= 0 || a->max_seqlen <= 0) return VM_ERR_BAD_ARG;
  *bytes = bwd_ds_bytes(a);
  return VM_OK;
}

int vm_attn_bwd_bf16(const vm_attn_args* a, void* stream) {
  if (!args_ok(a) || !a->dout || !a->dq || !a->dk || !a->dv || !a->delta) return VM_ERR_BAD_ARG;
  if (a->lddo % 8 || a->ldo % 8 || a->lddq % 4 || a->lddk % 4 || a->lddv % 4) return VM_ERR_BAD_ARG;
  if (!fits32(a)) return VM_ERR_UNSUPPORTED;
  void* tok = nullptr;
  vm_prof_begin_(VM_PROF_ATTN, stream, &tok);
  const int rc = bwd_launch(a, (hipStream_t)stream, 1, 7);
  vm_prof_end_(VM_PROF_ATTN, stream, tok, attn_flops(a, 5.0));
  if (rc != VM_OK) return rc;
  VM_LAUNCH_CHECK();
  return VM_OK;
}

}  // extern "C"
