// LoRA-side kernels of the fused LoRA linear (peft lora.Linear, conf/lora.yaml r = 64) that do not fit the big
// NT GEMM tile:
//
//  vm_lora_down : t[M,64] = drop(x)[M,K] · A[64,K]^T           (forward A-projection, and u = dy · B in the backward)
//                 skinny output => one 16-row slab per workgroup, K split over its 4 waves, dropout fused on the
//                 activation fragment (no dropped copy of x in HBM), per-row-segment weights for the gated experts.
//  vm_gemm_tn   : C[P,Q] (+)= X[M,P]^T · Y[M,Q]  contracted over token rows (dA, dB, and full weight gradients).
//                 Both operands are staged row-major exactly as they sit in HBM and fed to the MFMA through
//                 ds_read_b64_tr_b16 transposed reads, so no transposed copies of activations are ever written.
#include <atomic>
#include <cstdlib>
#include <mutex>
#include "vm_common.hpp"
#include "vm_tile.hpp"

extern "C" int vm_prof_begin_(int kind, void* stream, void** tok);
extern "C" int vm_prof_end_(int kind, void* stream, void* tok, double flops);

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, int64_t byte_off, int bytes) {
  const uint64_t a = (uint64_t)((const char*)base + byte_off);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  const int n = __builtin_amdgcn_readfirstlane(bytes);
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, n, 0x00020000);
}

// ============================================================================ vm_lora_down
struct DownP {
  const unsigned short* x; int64_t ldx;
  const unsigned short* A0; const unsigned short* A1; int64_t lda;
  unsigned short* t; int64_t ldt;
  int M, K;
  const int32_t* counts_dev; int split;
  float drop_p; uint64_t seed;
  float* ws; int ksplits;                   // fp32 partials [ksplits][M][64] when ksplits > 1
  unsigned* tickets;                        // != NULL: one arrival counter per row block (zero between launches) — the workgroup whose add comes
                                            // last sums the partials itself (no lora_reduce_k launch)
};
// Streaming skinny GEMM: a workgroup owns 64 rows x one K range. x (the HBM stream) and the 64 x K factor slice are
// staged per 128-wide K-tile through LDS-DMA (whole 256-byte row segments, 2 stages, 2 workgroups per CU), so the factor
// is fetched from L2 once per 64 rows instead of once per 16 (the earlier one-slab-per-workgroup form moved 4 bytes of A
// through L2 for every byte of x and sat at ~1.1 TB/s). Wave w multiplies rows 16w..16w+15 by all 64 factor rows with
// v_mfma_f32_16x16x32_bf16; the dropout mask is applied to the x fragment in registers, once per element.
// K is split over blockIdx.y to fill the chip; partial sums go to an fp32 workspace and are reduced in a FIXED order by
// lora_reduce_k (deterministic, no atomics).
// [r3] measured and NOT kept: (a) a 16-row single-pass form for K <= 4096 (four waves own 16 output columns each, 393 / 229
// workgroups, no reduce launch): 317.5 vs 317.8 ms at K <= 2048, 319.2 / 319.7 vs 317.8 ms at K <= 4096 — the factor is re-read per
// 16 rows; (b) a four-stage DMA ring for the single-pass grids: 320.3 vs 320.5 ms. Kept: always split K (4 x 99 workgroups + reduce
// at K = 1792 instead of 99 workgroups): -0.9 ms.
// With dropout the projection is VALU-bound, not memory-bound ([6280 x 15360]: 35 us without the mask, 64 us with it): ~100 VALU
// instructions per 8 elements, a third of them the hash's 32-bit multiplies (v_mul_lo_u32 issues at quarter rate). Eight waves per
// workgroup (two per 16-row group, half a K-tile each) to overlap mask generation with the DMA: 64.6 vs 64.1 us — it is issue-bound,
// not latency-bound; only a cheaper hash would help, and the hash is shared with the oracle and every other consumer of the mask.
constexpr int DN_BM = 64;

template <int AUX = 0>
__device__ __forceinline__ void stage_rows(const unsigned short* base, int64_t ld, int row_begin, int rows_valid,
                                           int col0, int cols_total, char* tile, int wave, int lane);

// STAGES = depth of the LDS-DMA ring (STAGES - 1 K-tiles of loads in flight per workgroup). 2: 64 KiB, two workgroups per CU — the
// K-split form, whose grid has more workgroups than CUs. 4: 128 KiB, one workgroup per CU — the single-pass form of a short K
// ([6280 x 1792]: 99 workgroups of 14 K-tiles; with one tile in flight each of them paid the full memory latency 14 times in a
// row: 19-23 us for 22.5 MB in situ).
// (round 5: the "element-wise producer + projection in one pass" modes of this kernel — GELU, GELU backward, SiLU * up computed on the
// staged tile, vm_lora_down_fused — are out of the library: bit-identical to the two-kernel form but slower inside the step in all three
// places, round 3's A B A: 322.5 / 327.6 / 321.9 vs 321.6 ms; DESIGN.md section 3 dead ends)
template <int STAGES>
__global__ __launch_bounds__(256, STAGES == 2 ? 2 : 1) void lora_down_k(const DownP p) {
  constexpr int NIN = 1;
  constexpr int DN_STAGE = (NIN + 1) * DN_BM * ROWB;                   // input tile + factor tile
  extern __shared__ __attribute__((aligned(16))) char smem[];          // STAGES * DN_STAGE
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int M = p.M, split = p.split;
  if (p.counts_dev) {
    split = __builtin_amdgcn_readfirstlane(p.counts_dev[0]);
    M = min(p.M, __builtin_amdgcn_readfirstlane(p.counts_dev[1]));
  }
  int row0, nrows, seg = 0;
  const int tb = blockIdx.x;
  if (split < 0) { row0 = tb * DN_BM; nrows = min(DN_BM, M - row0); }
  else {
    split = min(split, M);
    const int t0 = (split + DN_BM - 1) / DN_BM;
    if (tb < t0) { row0 = tb * DN_BM; nrows = min(DN_BM, split - row0); }
    else { seg = 1; row0 = split + (tb - t0) * DN_BM; nrows = min(DN_BM, M - row0); }
  }
  if (nrows <= 0) return;
  const unsigned short* A = seg ? p.A1 : p.A0;
  const int frow = lane & 15, fq = lane >> 4;
  const int kt_total = (p.K + 127) / 128;
  const int per = (kt_total + p.ksplits - 1) / p.ksplits;
  const int kt0 = blockIdx.y * per, kt1 = min(kt_total, kt0 + per);
  if (kt0 >= kt1) return;      // (the host sizes ksplits so that every split is non-empty)

  const bool drop = p.drop_p > 0.f;
  const unsigned thr = vm_drop_threshold(p.drop_p);
  const float inv_keep = drop ? 1.0f / (1.0f - p.drop_p) : 1.0f;
  const int64_t m = row0 + wave * 16 + frow;
  // mask index of the lane's 8 elements = m K + kk; below 2^34 elements the hash takes the 32-bit group index m (K / 4) + kk / 4
  const bool idx32 = vm_fits32(p.M, p.K) && (p.K & 3) == 0;
  const VmSeed sd = vm_seed(p.seed);
  const unsigned g_row = (unsigned)m * (unsigned)(p.K >> 2);

  auto stage = [&](int kt, int buf) {        // K-tiles past the range stage zero rows: every wave issues 8 DMA instructions per call
    char* sx = smem + buf * DN_STAGE;
    char* sa = sx + NIN * DN_BM * ROWB;
    const bool live = kt < kt1;
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf) {
      stage_rows(p.x, p.ldx, row0 + 32 * hlf, live ? max(0, nrows - 32 * hlf) : 0, kt * 128, p.K, sx + hlf * 32 * ROWB, wave, lane);
      stage_rows(A, p.lda, 32 * hlf, live ? 32 : 0, kt * 128, p.K, sa + hlf * 32 * ROWB, wave, lane);
    }
  };
  constexpr int DMA_PER_TILE = 4 * (NIN + 1);        // DMA instructions per wave and K-tile

  f32x4_t acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int i = 0; i < STAGES - 1; ++i) stage(kt0 + i, i);
  for (int kt = kt0; kt < kt1; ++kt) {
    const int buf = (kt - kt0) % STAGES;
    // this wave's part of K-tile `kt` has landed (STAGES - 2 tiles stay in flight: 8 DMA instructions per wave and tile) ...
    if (STAGES == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (STAGES == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                       // ... everybody's has, and tile kt - 1 has been consumed: its buffer is free
    stage(kt + STAGES - 1, (buf + STAGES - 1) % STAGES);
    const char* sx = smem + buf * DN_STAGE;
    const char* sa = sx + NIN * DN_BM * ROWB;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      u16x8_t xv = *reinterpret_cast<const u16x8_t*>(sx + tile_off(wave * 16 + frow, 4 * s + fq));
      if (drop) {
        const int kk = kt * 128 + 32 * s + 8 * fq;
        if (idx32) {
          const unsigned g = g_row + (unsigned)(kk >> 2);
          vm_mask8w(xv, vm_hash4w(sd, g), vm_hash4w(sd, g + 1), thr);           // 1/(1-p) is applied to the accumulators below
        } else {
          const uint64_t idx = (uint64_t)m * (uint64_t)p.K + (uint64_t)kk;       // multiple of 8
          const uint64_t h0 = vm_hash4(p.seed, idx >> 2), h1 = vm_hash4(p.seed, (idx >> 2) + 1);
          vm_mask8(xv, h0, h1, thr);
        }
      }
      const bf16x8_t xb = __builtin_bit_cast(bf16x8_t, xv);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bf16x8_t wa = *reinterpret_cast<const bf16x8_t*>(sa + tile_off(16 * i + frow, 4 * s + fq));
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb, acc[i], 0, 0, 0);
      }
    }
  }
  (void)DMA_PER_TILE;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the zero-row tail stages
  // D[row = r][col = m_local]: lane holds r = 16 i + 4 fq + 0..3 for row m
  const bool row_ok = wave * 16 + frow < nrows;
  if (drop) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] *= inv_keep;
  }
  if (p.ksplits <= 1) {
    if (!row_ok) return;
    unsigned short* o = p.t + m * p.ldt;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const u16x4_t v = {f2bf(acc[i][0]), f2bf(acc[i][1]), f2bf(acc[i][2]), f2bf(acc[i][3])};
      *reinterpret_cast<u16x4_t*>(o + 16 * i + 4 * fq) = v;
    }
    return;
  }
  if (!p.tickets) {                      // partials for lora_reduce_k
    if (!row_ok) return;
    float* w = p.ws + ((int64_t)blockIdx.y * p.M + m) * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4_t*>(w + 16 * i + 4 * fq) = acc[i];
    return;
  }
  // ---- one launch: partials out with write-through (sc1) stores, every wave drains its stores, the workgroup meets, ONE lane takes a
  // ticket; the workgroup that drew the last one sums all `ksplits` partials of its row block IN SPLIT ORDER with sc1 loads — the order and
  // the arithmetic of lora_reduce_k, so the two forms agree bit for bit (guide "Workgroup dispatch ... inter-workgroup visibility": sc1 payload,
  // vmcnt(0) of every storing wave, barrier, agent-scope counter add; the consumer loads after its add has returned, the other waves after
  // a barrier it joins; no fence: a release would write back the whole L2 — that form cost 45 -> 117 us in round 1). The counter is back at
  // zero when the last workgroup leaves.
  {
    const int64_t ws_rows = (int64_t)p.ksplits * p.M;                     // rows of 256 bytes
    const __amdgpu_buffer_rsrc_t rW = make_rsrc(p.ws, 0, (int)(ws_rows * 256));
    const int woff = (int)(((int64_t)blockIdx.y * p.M + m) * 256);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, acc[i]), rW, row_ok ? woff + 64 * i + 16 * fq : 0x7FFFFFF0, 0, 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flag = reinterpret_cast<int*>(smem);                            // (every wave is past its last LDS read: the barrier above)
    if (tid == 0) {
      unsigned* tk = p.tickets + blockIdx.x;
      const unsigned got = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = got + 1 == (unsigned)p.ksplits;
      if (last) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *flag = last;
    }
    __syncthreads();
    if (!*flag) return;
    // 64 rows x 16 groups of 4 columns: thread -> 4 (row, group) items, 16 lanes per 256-byte row
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int item = tid + 256 * j, r = item >> 4, c = (item & 15) * 4;
      if (r >= nrows) continue;
      const int64_t row = row0 + r;
      f32x4_t a = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rW, (int)(row * 256) + c * 4, 0, 16));
      for (int sp = 1; sp < p.ksplits; ++sp)
        a += __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rW, (int)(((int64_t)sp * p.M + row) * 256) + c * 4, 0, 16));
      const u16x4_t v = {f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3])};
      *reinterpret_cast<u16x4_t*>(p.t + row * p.ldt + c) = v;
    }
  }
}

// t[m][0..63] = bf16(sum over splits, in split order, of ws[s][m][0..63]); one thread per 4 consecutive outputs
__global__ __launch_bounds__(256) void lora_reduce_k(const float* __restrict__ ws, unsigned short* __restrict__ t, int64_t ldt,
                                                     int M, int ksplits, const int32_t* counts_dev) {
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t m = g >> 4;
  const int c = (int)(g & 15) * 4;
  int rows = M;
  if (counts_dev) rows = min(M, counts_dev[1]);
  if (m >= rows) return;
  f32x4_t a = *reinterpret_cast<const f32x4_t*>(ws + m * 64 + c);
  for (int s = 1; s < ksplits; ++s) a += *reinterpret_cast<const f32x4_t*>(ws + ((int64_t)s * M + m) * 64 + c);
  const u16x4_t v = {f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3])};
  *reinterpret_cast<u16x4_t*>(t + m * ldt + c) = v;
}

// ============================================================================ vm_gemm_tn
struct TnP {
  const unsigned short* X; int64_t ldx; int P;
  const unsigned short* Y; int64_t ldy; int Q;
  void* C; int64_t ldc;
  int M;
  const int32_t* counts_dev; int segment;   // row range from device counts: -1 = [0, M), 0 = [0,c0), 1 = [c0,c1)
  const int32_t* nrows_dev;                 // optional device row count when segment < 0
  int splits;                               // > 1: fp32 atomic accumulation into C (pre-zeroed)
  int out_f32;
  float drop_p; uint64_t seed; int drop_cols;   // inverted dropout on Y: element (m, q) of a [*, drop_cols] tensor
  float alpha;
  int tiles_p, tiles_q;
};

// stage 32 rows x 128 bf16 columns (256 B per row) of a row-major operand: LDS image is lane-linear, the swizzle is
// applied to the SOURCE chunk (rule 21); rows / columns outside the operand read as zero (buffer bounds + ld check)
template <int AUX>
__device__ __forceinline__ void stage_rows(const unsigned short* base, int64_t ld, int row_begin, int rows_valid,
                                           int col0, int cols_total, char* tile, int wave, int lane) {
  // the buffer covers rows [row_begin, row_begin + rows_valid); per-lane offset = r * ld*2 + col bytes
  const int ld_b = (int)ld * 2;
  __amdgpu_buffer_rsrc_t rs = make_rsrc(base, (int64_t)row_begin * ld_b, rows_valid > 0 ? rows_valid * ld_b : 0);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int inst = wave * 2 + i;                 // 8 wave-instructions of 1 KiB = 4 rows each
    const int row = inst * 4 + (lane >> 4);
    const int slot = lane & 15;
    const int chunk = slot ^ swz(row);
    const int col = col0 + chunk * 8;
    // columns past the operand width must read zero: push the offset out of the buffer
    const int voff = (col + 8 <= cols_total) ? row * ld_b + col * 2 : 0x7FFFFFF0;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(tile + inst * 1024), 16, voff, 0, 0, AUX);
  }
}

__global__ __launch_bounds__(256, 2) void gemm_tn_k(const TnP p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * 32 * ROWB];   // [buf][X|Y][32 rows][256 B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave >> 1, wq = wave & 1;
  const int tile = blockIdx.x;
  const int tp = tile / p.tiles_q, tq = tile % p.tiles_q;
  const int p0 = tp * 128, q0 = tq * 128;

  int rb = 0, re = p.M;
  if (p.counts_dev) {
    const int c0 = __builtin_amdgcn_readfirstlane(p.counts_dev[0]);
    const int c1 = min(p.M, __builtin_amdgcn_readfirstlane(p.counts_dev[1]));
    if (p.segment == 0) { rb = 0; re = min(c0, c1); }
    else if (p.segment == 1) { rb = min(c0, c1); re = c1; }
    else { rb = 0; re = c1; }
  } else if (p.nrows_dev) {
    re = min(p.M, __builtin_amdgcn_readfirstlane(p.nrows_dev[0]));
  }
  // split the row range over blockIdx.y in multiples of 32 rows
  const int total_steps = (re - rb + 31) / 32;
  const int per = (total_steps + p.splits - 1) / p.splits;
  const int s_begin = blockIdx.y * per, s_end = min(total_steps, s_begin + per);
  if (s_begin >= s_end && !(p.splits == 1)) return;

  f32x16_t acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  auto stage = [&](int step, int buf) {
    const int r0 = rb + step * 32;
    const int valid = min(32, re - r0);
    char* sx = smem + buf * (2 * 32 * ROWB);
    char* sy = sx + 32 * ROWB;
    stage_rows(p.X, p.ldx, r0, valid, p0, p.P, sx, wave, lane);
    stage_rows(p.Y, p.ldy, r0, valid, q0, p.Q, sy, wave, lane);
  };
  const bool drop = p.drop_p > 0.f;
  const unsigned thr = vm_drop_threshold(p.drop_p);

  if (s_begin < s_end) stage(s_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int st = s_begin; st < s_end; ++st) {
    const int buf = (st - s_begin) & 1;
    if (st + 1 < s_end) stage(st + 1, buf ^ 1);
    char* sx = smem + buf * (2 * 32 * ROWB);
    char* sy = sx + 32 * ROWB;
    if (drop) {
      // apply the inverted-dropout mask to the Y tile in place: thread -> 2 chunks of 8 consecutive columns
      const int r0 = rb + st * 32;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = tid + i * 256;
        const int row = c >> 4, chunk = c & 15;
        const int col = q0 + chunk * 8;
        char* addr = sy + tile_off(row, chunk);
        u16x8_t v = *reinterpret_cast<u16x8_t*>(addr);
        const uint64_t idx = (uint64_t)(r0 + row) * (uint64_t)p.drop_cols + (uint64_t)col;
        const uint64_t h0 = vm_hash4(p.seed, idx >> 2), h1 = vm_hash4(p.seed, (idx >> 2) + 1);
        vm_mask8(v, h0, h1, thr);            // 1/(1-p) is folded into alpha by the launcher
        *reinterpret_cast<u16x8_t*>(addr) = v;
      }
      __syncthreads();
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8_t xa[2], yb[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) xa[a] = frag_tr(sx, 16 * s, 2 * wp + a, lane);
#pragma unroll
      for (int b = 0; b < 2; ++b) yb[b] = frag_tr(sy, 16 * s, 2 * wq + b, lane);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[a], yb[b], acc[a][b], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // D[i = p_local][j = q_local]: col = lane & 31 -> q, rows (r&3) + 8 (r>>2) + 4 h -> p
  const int h = lane >> 5;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int qq = q0 + wq * 64 + b * 32 + (lane & 31);
      if (qq >= p.Q) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pp = p0 + wp * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (pp >= p.P) continue;
        const float v = acc[a][b][r] * p.alpha;
        if (p.splits > 1) atomicAdd((float*)p.C + (int64_t)pp * p.ldc + qq, v);
        else if (p.out_f32) ((float*)p.C)[(int64_t)pp * p.ldc + qq] = v;
        else ((unsigned short*)p.C)[(int64_t)pp * p.ldc + qq] = f2bf(v);
      }
    }
}

// ============================================================================ vm_gemm_tn_f32 (fp32 weight gradients, TN form)
// C[P, Q] += X[M, P]^T Y[M, Q] for fp32 operands as they sit in HBM (dW = dy^T x of every unfrozen nn.Linear of the SAM / iSAM
// islands): round 2 transposed BOTH operands into K-contiguous copies for the NT kernel (618 transpose launches per step).
// Split-bf16 arithmetic as vm_gemm_f32 (NS = 2: three products, NS = 3: six), but the split happens ONCE, in the threads that
// stage a tile: 32-row steps are loaded row-major with 16-byte loads, split into NS bf16 planes and written to LDS in the
// transposed-read image of vm_tile.hpp; the MFMA operands are then ds_read_b64_tr_b16 fragments exactly as in gemm_tn_k.
// Tile 64 (P) x 128 (Q), one workgroup walks all rows when the grid fills the chip (no atomics: C += is a plain
// read-modify-write of an exclusively owned tile), else rows are split over blockIdx.y with fp32 atomics whose access shape is
// the full-rate one (a 32x32 accumulator register = two 128-byte row segments per wave instruction).
// `colsum`: the column sums of X (the bias gradient) are accumulated by the workgroups of the first Q tile from the values they
// stage anyway.
struct TnF32P {
  const float* X; int64_t ldx; int P;
  const float* Y; int64_t ldy; int Q;
  float* C; int64_t ldc;
  float* colsum;
  int M, splits, tiles_q;
};

template <int NS>
__device__ __forceinline__ void split_store8(const f32x4_t& a, const f32x4_t& b, char* planes, int plane_bytes, int off) {
  float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    u16x8_t h;
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = f2bf(x[e]);
    *reinterpret_cast<u16x8_t*>(planes + s * plane_bytes + off) = h;
    if (s + 1 < NS) {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] -= bf2f(h[e]);
    }
  }
}

template <int NS>
__global__ __launch_bounds__(256, 2) void gemm_tn_f32_k(const TnF32P p) {
  constexpr int PLANE = 32 * ROWB;                 // one bf16 plane of a 32-row step (128 columns wide; X uses the first 64)
  constexpr int STAGE = 2 * NS * PLANE;            // X planes, then Y planes
  extern __shared__ __attribute__((aligned(16))) char smem[];          // 2 * STAGE
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wp = wave >> 1, wq = wave & 1;
  const int tp = blockIdx.x / p.tiles_q, tq = blockIdx.x % p.tiles_q;
  const int p0 = tp * 64, q0 = tq * 128;
  const int total_steps = (p.M + 31) / 32;
  const int per = (total_steps + p.splits - 1) / p.splits;
  const int s_begin = blockIdx.y * per, s_end = min(total_steps, s_begin + per);
  if (s_begin >= s_end) return;

  f32x16_t acc[2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

  // staging map: thread -> row tid >> 3 of the step, 8 consecutive X columns and 16 consecutive Y columns
  const int srow = tid >> 3, sg = tid & 7;
  const int xcol = p0 + sg * 8, ycol = q0 + sg * 16;
  const bool xin = xcol + 8 <= p.P;                          // (P % 8 == 0, Q % 8 == 0: whole 8-column groups)
  const bool yin0 = ycol + 8 <= p.Q, yin1 = ycol + 16 <= p.Q;
  f32x4_t rx[2], ry[4];
  const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
  auto load = [&](int step) {
    const int m = step * 32 + srow;
    const bool rin = m < p.M;
    const float* xr = p.X + (int64_t)m * p.ldx + xcol;
    const float* yr = p.Y + (int64_t)m * p.ldy + ycol;
    rx[0] = (rin && xin) ? *reinterpret_cast<const f32x4_t*>(xr) : zero4;
    rx[1] = (rin && xin) ? *reinterpret_cast<const f32x4_t*>(xr + 4) : zero4;
    ry[0] = (rin && yin0) ? *reinterpret_cast<const f32x4_t*>(yr) : zero4;
    ry[1] = (rin && yin0) ? *reinterpret_cast<const f32x4_t*>(yr + 4) : zero4;
    ry[2] = (rin && yin1) ? *reinterpret_cast<const f32x4_t*>(yr + 8) : zero4;
    ry[3] = (rin && yin1) ? *reinterpret_cast<const f32x4_t*>(yr + 12) : zero4;
  };
  const bool do_colsum = p.colsum != nullptr && tq == 0;
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto split_write = [&](int buf) {
    char* sx = smem + buf * STAGE;
    char* sy = sx + NS * PLANE;
    split_store8<NS>(rx[0], rx[1], sx, PLANE, tile_off(srow, sg));
    split_store8<NS>(ry[0], ry[1], sy, PLANE, tile_off(srow, 2 * sg));
    split_store8<NS>(ry[2], ry[3], sy, PLANE, tile_off(srow, 2 * sg + 1));
    if (do_colsum) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { bsum[e] += rx[0][e]; bsum[4 + e] += rx[1][e]; }
    }
  };

  load(s_begin);
  split_write(0);
  __syncthreads();
  for (int st = s_begin; st < s_end; ++st) {
    const int buf = (st - s_begin) & 1;
    const bool more = st + 1 < s_end;
    if (more) load(st + 1);                                   // global loads in flight under the MFMAs of this step
    const char* sx = smem + buf * STAGE;
    const char* sy = sx + NS * PLANE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8_t xa[NS], yb[NS][2];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        xa[s] = frag_tr(sx + s * PLANE, 16 * ks, wp, lane);
        yb[s][0] = frag_tr(sy + s * PLANE, 16 * ks, 2 * wq, lane);
        yb[s][1] = frag_tr(sy + s * PLANE, 16 * ks, 2 * wq + 1, lane);
      }
      // cross terms x_s y_t with s + t < NS, smallest first
#pragma unroll
      for (int d = NS - 1; d >= 0; --d)
#pragma unroll
        for (int sw = 0; sw <= d; ++sw)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[sw], yb[d - sw][b], acc[b], 0, 0, 0);
    }
    if (more) split_write(buf ^ 1);
    __syncthreads();
  }
  // D[i = p_local][j = q_local]: col = lane & 31 -> q, rows (r&3) + 8 (r>>2) + 4 h -> p
  const int h = lane >> 5;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int qq = q0 + wq * 64 + b * 32 + (lane & 31);
    if (qq >= p.Q) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int pp = p0 + wp * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (pp >= p.P) continue;
      float* c = p.C + (int64_t)pp * p.ldc + qq;
      if (p.splits > 1) atomicAdd(c, acc[b][r]);
      else *c += acc[b][r];
    }
  }
  if (do_colsum) {
    // the 8 row-threads of a wave that share a column group: lanes sg, sg + 8, ..., sg + 56
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = bsum[e];
      v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
      if (lane < 8 && xcol + e < p.P) atomicAdd(p.colsum + xcol + e, v);
    }
  }
}

// ============================================================================ vm_tn_skinny (LoRA factor gradients)
// O[c][n] = sum_m W[m][c] * S[m][n] with a WIDE streamed operand W [M, C] and a rank-64 operand S [M, 64]:
//   dB[N, 64] = s * dy^T t      (W = dy, S = t)            -> out[c][n]
//   dA[64, K] = s * u^T drop(x) (W = x with the forward's LoRA dropout mask, S = u) -> out[n][c] (transpose_out)
// A workgroup owns 64 columns of W and a range of rows; 32-row steps run through an LDS-DMA ring (SK_STAGES - 1 steps of
// loads in flight, one barrier per step). Two stages = 32 KiB: measured as fast as four in isolation (4.8 vs 4.6 TB/s on
// [6280 x 15360]; several workgroups per CU hide the latency instead) and small enough to sit beside a 128 KiB workgroup
// of the 256x256 GEMM on the same CU, which matters because these kernels run on the side stream under the dgrad GEMMs. Row-range partial sums go to an fp32 workspace and tn_reduce_k adds them in a
// fixed order, scales, optionally accumulates into the existing gradient and writes bf16 / fp32 (deterministic: no
// atomics, no pre-zeroed buffer, no separate cast).
struct SkP {
  const unsigned short* W; int64_t ldw; int C;
  const unsigned short* S; int64_t lds;
  float* ws; int c_pad;
  int M;
  const int32_t* counts_dev; int segment;
  const int32_t* nrows_dev;
  int splits;
  float drop_p; uint64_t seed;
};
constexpr int SK_STAGE = 2 * 32 * ROWB;
#ifndef VM_SK_STAGES
#define VM_SK_STAGES 2
#endif
constexpr int SK_STAGES = VM_SK_STAGES;     // ring depth: SK_STAGES - 1 steps of loads in flight (4 DMA instructions per wave and step)

// BC = 64: waves as 2 (c) x 2 (n), one 32x32 accumulator each — many small tiles for narrow W (C of a few thousand);
// BC = 128: wave w owns columns 32w..32w+31 and both n halves — full 256-byte row segments for wide W.
template <int BC>
__global__ __launch_bounds__(256, 2) void tn_skinny_k(const SkP p) {
  __shared__ __attribute__((aligned(16))) char smem[SK_STAGES * SK_STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NB = BC == 128 ? 2 : 1;                 // 32-wide n blocks per wave
  const int c0 = blockIdx.x * BC;
  const int wc = BC == 128 ? wave : wave >> 1, wn = BC == 128 ? 0 : wave & 1;
  int rb = 0, re = p.M;
  if (p.counts_dev) {
    const int k0 = __builtin_amdgcn_readfirstlane(p.counts_dev[0]);
    const int k1 = min(p.M, __builtin_amdgcn_readfirstlane(p.counts_dev[1]));
    const int seg = p.segment == 2 ? (int)blockIdx.z : p.segment;      // 2: both segments in one launch (grid.z)
    if (seg == 0) { rb = 0; re = min(k0, k1); }
    else if (seg == 1) { rb = min(k0, k1); re = k1; }
    else { rb = 0; re = k1; }
  } else if (p.nrows_dev) {
    re = min(p.M, __builtin_amdgcn_readfirstlane(p.nrows_dev[0]));
  }
  const int total_steps = (re - rb + 31) / 32;
  const int per = (total_steps + p.splits - 1) / p.splits;
  const int s_begin = blockIdx.y * per, s_end = min(total_steps, s_begin + per);

  f32x16_t acc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

  const int c_end = min(p.C, c0 + BC);         // BC = 64: 128-byte row segments (half of the DMA lanes fetch zeros)
  auto stage = [&](int step, int buf) {      // steps past the range stage zero rows (uniform DMA count for s_waitcnt)
    const int r0 = rb + step * 32;
    const int valid = step < s_end ? min(32, re - r0) : 0;
    char* sw = smem + buf * SK_STAGE;
    stage_rows(p.W, p.ldw, r0, valid, c0, c_end, sw, wave, lane);
    stage_rows(p.S, p.lds, r0, valid, 0, 64, sw + 32 * ROWB, wave, lane);
  };
  const bool drop = p.drop_p > 0.f;
  const unsigned thr = vm_drop_threshold(p.drop_p);

#pragma unroll
  for (int i = 0; i < SK_STAGES - 1; ++i) stage(s_begin + i, i);
  for (int st = s_begin; st < s_end; ++st) {
    const int buf = (st - s_begin) % SK_STAGES;
    if (SK_STAGES == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // this wave's part of step `st` has landed (2 steps stay in flight)
    else if (SK_STAGES == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                        // ... everybody's has, and step st-1 has been consumed
    stage(st + SK_STAGES - 1, (buf + SK_STAGES - 1) % SK_STAGES);
    char* sw = smem + buf * SK_STAGE;
    const char* ss = sw + 32 * ROWB;
    if (drop) {
      const int r0 = rb + st * 32;
#pragma unroll
      for (int i = 0; i < BC / 64; ++i) {       // 32 rows x BC/8 chunks: one (BC = 64) or two 16-byte chunks per thread
        const int cc = tid + i * 256;
        const int row = BC == 128 ? cc >> 4 : cc >> 3, chunk = BC == 128 ? cc & 15 : cc & 7;
        const int col = c0 + chunk * 8;
        char* addr = sw + tile_off(row, chunk);
        u16x8_t v = *reinterpret_cast<u16x8_t*>(addr);
        const uint64_t idx = (uint64_t)(r0 + row) * (uint64_t)p.C + (uint64_t)col;
        const uint64_t h0 = vm_hash4(p.seed, idx >> 2), h1 = vm_hash4(p.seed, (idx >> 2) + 1);
        vm_mask8(v, h0, h1, thr);            // 1/(1-p) is folded into alpha by the launcher
        *reinterpret_cast<u16x8_t*>(addr) = v;
      }
      __syncthreads();
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8_t wa = frag_tr(sw, 16 * ks, wc, lane);
#pragma unroll
      for (int b = 0; b < NB; ++b)
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, frag_tr(ss, 16 * ks, wn + b, lane), acc[b], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // drain the zero-row tail stages before the LDS is released
  // partial tile: ws[split][c][n], n fastest
  const int h = lane >> 5;
  float* w = p.ws + ((int64_t)(blockIdx.z * p.splits + blockIdx.y) * p.c_pad + c0 + wc * 32) * 64 + 32 * wn + (lane & 31);
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) w[(int64_t)((r & 3) + 8 * (r >> 2) + 4 * h) * 64 + 32 * b] = acc[b][r];
}

// out = (accumulate ? out : 0) + alpha * sum_s ws[s]; block = 16 columns c x 64 n, one float4 of n per thread;
// the transposed form goes through LDS so that a thread writes 4 consecutive c
template <typename TO>
__global__ __launch_bounds__(256) void tn_reduce_k(const float* __restrict__ ws, int c_pad, int C, int splits, TO* __restrict__ out,
                                                   TO* __restrict__ out1, int64_t ldo, int transpose_out, int accumulate, float alpha) {
  __shared__ float tile[64][17];
  if (blockIdx.y) { out = out1; ws += (int64_t)splits * c_pad * 64; }      // second row segment of the merged form
  const int c0 = blockIdx.x * 16;
  const int tid = threadIdx.x;
  const int cl = tid >> 4, n4 = (tid & 15) * 4;
  const float* src = ws + (int64_t)(c0 + cl) * 64 + n4;
  const int64_t sstride = (int64_t)c_pad * 64;
  f32x4_t a = {0.f, 0.f, 0.f, 0.f};
  int s = 0;
  // the kernel is pure load latency (a few MB through 112-960 workgroups): keep eight loads in flight per thread; the summation
  // order stays s = 0, 1, 2, ... (deterministic)
  for (; s + 8 <= splits; s += 8) {
    f32x4_t v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x4_t*>(src + (s + j) * sstride);
#pragma unroll
    for (int j = 0; j < 8; ++j) a += v[j];
  }
  for (; s + 4 <= splits; s += 4) {
    const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(src + (s + 0) * sstride);
    const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(src + (s + 1) * sstride);
    const f32x4_t v2 = *reinterpret_cast<const f32x4_t*>(src + (s + 2) * sstride);
    const f32x4_t v3 = *reinterpret_cast<const f32x4_t*>(src + (s + 3) * sstride);
    a += v0; a += v1; a += v2; a += v3;
  }
  for (; s < splits; ++s) a += *reinterpret_cast<const f32x4_t*>(src + s * sstride);
  a *= alpha;
  if (!transpose_out) {
    if (c0 + cl < C) {
      TO* o = out + (int64_t)(c0 + cl) * ldo + n4;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = Elem<TO>::st(a[e] + (accumulate ? Elem<TO>::ld(o[e]) : 0.f));
    }
    return;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) tile[n4 + e][cl] = a[e];
  __syncthreads();
  const int n = tid >> 2, cb = (tid & 3) * 4;      // thread -> row n, 4 consecutive c
  TO* o = out + (int64_t)n * ldo + c0 + cb;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (c0 + cb + e < C) o[e] = Elem<TO>::st(tile[n][cb + e] + (accumulate ? Elem<TO>::ld(o[e]) : 0.f));
}


// ============================================================================ vm_tn_skinny_group (a batch of LoRA factor gradients)
// Up to VM_TN_GROUP_MAX independent factor gradients (vm_tn_group_item: out += alpha * W^T S over a row range) in ONE launch:
// the backward of a transformer layer produces 8 (ViT-E) or 20 (decoder: two experts) of them, each a 28-240 workgroup problem
// when launched alone — too small for 256 CUs without row splits, and with row splits each needs a second launch that reduces
// the fp32 partials (round 2: 16 launches per ViT layer, 236 us of kernel time). Here a workgroup owns 64 columns of ONE item and
// walks ALL of its rows (no partials, no reduce launch, no atomics: deterministic), and the items of a whole layer together are
// 700+ equally long workgroups. The result is added into `out` (a gradient-bucket slot) with one rounding.
struct GroupP { int n; vm_tn_group_item it[VM_TN_GROUP_MAX]; };
// A workgroup owns GR_BC = 256 columns of one item (wave w: columns 64 w .. 64 w + 63, both 32-wide halves of the 64 n). What bounds
// these kernels is the bytes a CU can take in through LDS-DMA (~25 GB/s per CU, 6.4 TB/s for the chip: MI355X_MICROARCH.md 'ldsdma-fill';
// the first form of this kernel, 64 columns per workgroup, moved a 128-byte row of the rank-64 operand for every 128 bytes of W and ran at
// exactly half that rate: 584 us for three ViT-E layers): with 256 columns the rank-64 operand is a quarter of the W bytes.
constexpr int GR_BC = 256;
constexpr int GR_STAGE_BYTES = 3 * 32 * ROWB;      // W columns 0..127, W columns 128..255, S (the first 128 bytes of each row are used)
#ifndef VM_GR_STAGES
#define VM_GR_STAGES 2
#endif
// ring depth. 2 (48 KiB, three workgroups per CU = 768 slots): the items of three ViT-E layers are 528 workgroups, and with the 512 slots
// of the three-stage ring the 16 left over ran a second round alone (533 us per launch; the chip-wide HBM rate would allow ~330)
constexpr int GR_STAGES = VM_GR_STAGES;

__global__ __launch_bounds__(256, GR_STAGES == 2 ? 3 : 2) void tn_group_k(const GroupP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];          // GR_STAGES * GR_STAGE_BYTES
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // item of this block: the table is tiny and wave-uniform (kernel arguments live in SGPRs / the scalar cache)
  int idx = 0;
#pragma unroll 1
  for (int i = 1; i < p.n; ++i) idx = ((int)blockIdx.x >= p.it[i].block0) ? i : idx;
  const vm_tn_group_item& q = p.it[idx];
  const unsigned short* W = (const unsigned short*)q.W;
  const unsigned short* S = (const unsigned short*)q.S;
  const int C = q.C;
  const int c0 = ((int)blockIdx.x - q.block0) * GR_BC;
  int rb = 0, re = q.M;
  if (q.counts_dev) {
    const int k0 = __builtin_amdgcn_readfirstlane(q.counts_dev[0]);
    const int k1 = min(q.M, __builtin_amdgcn_readfirstlane(q.counts_dev[1]));
    if (q.segment == 0) { rb = 0; re = min(k0, k1); }
    else if (q.segment == 1) { rb = min(k0, k1); re = k1; }
    else { rb = 0; re = k1; }
  }
  const int steps = (re - rb + 31) / 32;
  f32x16_t acc[2][2];          // [32-column block of the wave's 64][32-wide half of n]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  auto stage = [&](int step, int buf) {      // steps past the range stage zero rows (uniform DMA count for s_waitcnt: 6 per wave)
    const int r0 = rb + step * 32;
    const int valid = step < steps ? min(32, re - r0) : 0;
    char* sw = smem + buf * GR_STAGE_BYTES;
    // (W = the layer's activations or output gradients, read here for the last time in the step: non-temporal LDS-DMA (aux 2) —
    // 304.2 / 304.15 vs 303.6 / 303.95 ms per step, A B A B, gpurun_out/s2_ab_tnnt.log; S is re-read by every workgroup: default policy)
    stage_rows<2>(W, q.ldw, r0, valid, c0, C, sw, wave, lane);
    stage_rows<2>(W, q.ldw, r0, valid, c0 + 128, C, sw + 32 * ROWB, wave, lane);
    stage_rows(S, q.lds, r0, valid, 0, 64, sw + 2 * 32 * ROWB, wave, lane);
  };
  const bool drop = q.drop_p > 0.f;
  const unsigned thr = vm_drop_threshold(q.drop_p);
  const bool idx32 = vm_fits32(q.M, C) && (C & 3) == 0;
  const VmSeed sd = vm_seed(q.seed);
  const TrLane trl = tr_lane(smem, lane);
#pragma unroll
  for (int i = 0; i < GR_STAGES - 1; ++i) stage(i, i);
  for (int st = 0; st < steps; ++st) {
    const int buf = st % GR_STAGES;
    if (GR_STAGES == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // this wave's part of step `st` has landed (one step stays in flight)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                        // ... everybody's has, and step st-1 has been consumed
    stage(st + GR_STAGES - 1, (buf + GR_STAGES - 1) % GR_STAGES);
    char* sw = smem + buf * GR_STAGE_BYTES;
    if (drop) {
      const int r0 = rb + st * 32;
#pragma unroll
      for (int i = 0; i < 4; ++i) {               // 32 rows x 32 chunks of 8 columns: four per thread
        const int cc = tid + i * 256;
        const int row = cc >> 5, ch = cc & 31;    // chunk ch of the 256-column row: sub-tile ch >> 4, chunk ch & 15 inside it
        const int col = c0 + ch * 8;
        char* addr = sw + (ch >> 4) * 32 * ROWB + tile_off(row, ch & 15);
        u16x8_t v = *reinterpret_cast<u16x8_t*>(addr);
        if (idx32) {
          const unsigned g = (unsigned)(r0 + row) * (unsigned)(C >> 2) + (unsigned)(col >> 2);
          vm_mask8w(v, vm_hash4w(sd, g), vm_hash4w(sd, g + 1), thr);            // 1/(1-p) is folded into alpha by the launcher
        } else {
          const uint64_t e = (uint64_t)(r0 + row) * (uint64_t)C + (uint64_t)col;
          const uint64_t h0 = vm_hash4(q.seed, e >> 2), h1 = vm_hash4(q.seed, (e >> 2) + 1);
          vm_mask8(v, h0, h1, thr);
        }
        *reinterpret_cast<u16x8_t*>(addr) = v;
      }
      // (not __syncthreads(): its fence would drain the DMA of the next step that was issued above)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    // the step's eight transposed fragments as one batch of assembly reads (vm_tile.hpp: the builtin form makes hipcc wait for the
    // DMA issued above before the first read, i.e. nothing of the next step's flight would overlap these MFMAs)
    const unsigned o = buf * GR_STAGE_BYTES;
    const unsigned sA = trl.a + o + 2 * 32 * ROWB, sB = trl.b + o + 2 * 32 * ROWB;                    // S
    const unsigned wA = (trl.a + o + (wave >> 1) * 32 * ROWB) ^ ((wave & 1) << 7);                     // the wave's 128-column sub-tile,
    const unsigned wB = (trl.b + o + (wave >> 1) * 32 * ROWB) ^ ((wave & 1) << 7);                     // 32-column blocks 2 (wave & 1) + 0 / 1
    bf16x8_t sf[2][2], wf[2][2];
    sf[0][0] = tr_asm2<0>(sA, sB);                 sf[0][1] = tr_asm2<0>(sA ^ 64, sB ^ 64);
    wf[0][0] = tr_asm2<0>(wA, wB);                 wf[0][1] = tr_asm2<0>(wA ^ 64, wB ^ 64);
    sf[1][0] = tr_asm2<16 * ROWB>(sA, sB);         sf[1][1] = tr_asm2<16 * ROWB>(sA ^ 64, sB ^ 64);
    wf[1][0] = tr_asm2<16 * ROWB>(wA, wB);         wf[1][1] = tr_asm2<16 * ROWB>(wA ^ 64, wB ^ 64);
    tr_asm_wait();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][a], sf[ks][0], acc[a][0], 0, 0, 0);
        acc[a][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks][a], sf[ks][1], acc[a][1], 0, 0, 0);
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // drain the zero-row tail stages before the LDS is reused
  __syncthreads();
  // every wave parks one 32 (c) x 64 (n) half of its fp32 tile at a time in its own 8.1 KiB of LDS (pitch 65 floats) and adds
  // consecutive outputs of one output row per lane (same-wave LDS round trips: the compiler orders the reads behind the writes)
  float* tile = reinterpret_cast<float*>(smem) + wave * (32 * 65);
  const int h = lane >> 5;
  const float alpha = q.alpha;
#pragma unroll 1
  for (int a = 0; a < 2; ++a) {
    const int cw0 = c0 + wave * 64 + a * 32;                // first column of this half
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * h) * 65 + b * 32 + (lane & 31)] = (a ? acc[1][b][r] : acc[0][b][r]);
    if (!q.transpose_out) {
#pragma unroll 1
      for (int pass = 0; pass < 2; ++pass) {
        // out[cw0 + orow][o16 .. o16 + 15]
        const int orow = pass * 16 + (lane >> 2), o16 = (lane & 3) * 16;
        if (cw0 + orow >= C) continue;
        float v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = alpha * tile[orow * 65 + o16 + e];
        if (q.out_f32) {
          float* o = (float*)q.out + (int64_t)(cw0 + orow) * q.ldo + o16;
#pragma unroll
          for (int e = 0; e < 16; e += 4) {
            f32x4_t t = *reinterpret_cast<f32x4_t*>(o + e);
            t += (f32x4_t){v[e], v[e + 1], v[e + 2], v[e + 3]};
            *reinterpret_cast<f32x4_t*>(o + e) = t;
          }
        } else {
          unsigned short* o = (unsigned short*)q.out + (int64_t)(cw0 + orow) * q.ldo + o16;
#pragma unroll
          for (int e = 0; e < 16; e += 8) {
            u16x8_t t = *reinterpret_cast<u16x8_t*>(o + e);
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = f2bf(bf2f(t[k]) + v[e + k]);
            *reinterpret_cast<u16x8_t*>(o + e) = t;
          }
        }
      }
    } else {
#pragma unroll 1
      for (int pass = 0; pass < 4; ++pass) {
        // out[orow = n][cw0 + o8 .. + 7]: the transposed tile read column-wise out of LDS
        const int orow = pass * 16 + (lane >> 2), o8 = (lane & 3) * 8;
        const int cbase = cw0 + o8;
        if (cbase >= C) continue;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = alpha * tile[(o8 + e) * 65 + orow];
        if (q.out_f32) {
          float* o = (float*)q.out + (int64_t)orow * q.ldo + cbase;
          for (int e = 0; e < 8; ++e) if (cbase + e < C) o[e] += v[e];
        } else {
          unsigned short* o = (unsigned short*)q.out + (int64_t)orow * q.ldo + cbase;
          if (cbase + 8 <= C && (q.ldo % 8) == 0) {
            u16x8_t t = *reinterpret_cast<u16x8_t*>(o);
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = f2bf(bf2f(t[k]) + v[k]);
            *reinterpret_cast<u16x8_t*>(o) = t;
          } else {
            for (int e = 0; e < 8; ++e) if (cbase + e < C) o[e] = f2bf(bf2f(o[e]) + v[e]);
          }
        }
      }
    }
  }
}

// row splits: about 1.5 workgroups per CU, at least 8 steps each; the fp32 partials then stay a fraction of the input
static int tn_skinny_bc(int C) { return C >= 6144 ? 128 : 64; }
static int tn_skinny_splits(int M, int C) {
  const int bc = tn_skinny_bc(C);
  const int tiles = (C + bc - 1) / bc;
  const int steps = (M + 31) / 32;
  int want = (512 + tiles - 1) / tiles;        // two workgroups per CU (32 KiB LDS each): optimum of a 192..768 sweep in isolation;
                                               // inside the step 256 / 512 / 768 are within noise (375-376 ms), 1024 costs 2 ms
  want = max(1, min(want, steps / 8));
  const int per = (steps + want - 1) / want;
  return (steps + per - 1) / per;
}

}  // namespace

extern "C" {

// K-splits the kernel will use for (M, K). Measured on MI355X: in isolation (tools/bench_lora.py, kernel + reduce) about one
// workgroup per CU is the optimum — [6280 x 15360] 45.3 us with 6 splits (594 workgroups), 38.8 us with 3 (297): every extra
// split writes and re-reads M x 64 fp32 partials — but inside the training step, where the side stream's kernels compete for
// the CUs, 1.5 workgroups per CU is better (step time 376.6 ms vs 378.6 ms at one per CU, 377.4 ms at two). A short K
// (<= 16 K-tiles) is fastest in a single pass without the reduce launch ([6280 x 1792]: 11.0 us vs 12.8 us).
// zeroed counter pool of the one-launch K-split form, one per device, allocated on first use — never under stream capture (the decode
// step's hipGraph: a launch captured before the pool exists keeps the two-kernel form)
constexpr int LD_TICKET_REGION = 1024, LD_TICKET_REGIONS = 64;
static unsigned* lora_ticket_pool(hipStream_t stream) {
  static std::mutex mu;
  static unsigned* pools[16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  if (pools[dev]) return pools[dev];
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;
  unsigned* ptr = nullptr;
  const size_t bytes = (size_t)LD_TICKET_REGION * LD_TICKET_REGIONS * sizeof(unsigned);
  if (hipMalloc(&ptr, bytes) != hipSuccess) return nullptr;
  if (hipMemset(ptr, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(ptr); return nullptr; }
  pools[dev] = ptr;
  return ptr;
}

static int& lora_two_kernels() { static int v = 0; return v; }
/* internal (tests, A/B): 0 = by shape (one launch with the last-arriver reduction for <= 2 row blocks, else partials + lora_reduce_k),
 * 1 = always two kernels (the form before round 5), 2 = one launch whenever legal */
int vm_lora_down_two_kernels_(int mode) { lora_two_kernels() = (mode >= 0 && mode <= 2) ? mode : 0; return VM_OK; }

static int& lora_down_target() { static int t = 512; return t; }
/* internal (A/B): workgroup slots a lora_down launch may fill (two 64 KiB workgroups per CU x 256 CUs) */
int vm_lora_down_target_(int wgs) { if (wgs == 0 || wgs < -4096 || wgs > 4096) return VM_ERR_BAD_ARG; lora_down_target() = wgs; return VM_OK; }      // < 0: the rule before round 5b (split until ~|wgs| workgroups), for A/B runs
// K is always split (a single-pass form for short K measured 0.9 ms per step slower). Cost model of a launch (us): the chip holds `slots`
// workgroups at once (two 64 KiB workgroups per CU); a workgroup walks `per` K-tiles of 128, one memory round trip in flight behind the one it
// computes on: a ROUND of n workgroups takes max(latency-bound per x 1.25 us, bandwidth-bound n x per x 16 KiB / 5.5 TB/s) + 4 us of
// dispatch / cold first tile / partial store (calibrated on [6280 x 1792 / 5376 / 15360]: 12.5 / 25 / 35 us at 4 splits), and a last round
// of a few workgroups still costs a whole latency-bound workgroup. The split count with the
// cheapest sum wins, with a small charge per split for the partials the reduce kernel reads. What this replaces — "split until ~384
// workgroups" — put 514 workgroups on 512 slots at 16 392 rows (two stragglers doubling the launch) and 534 at 11 336; at 6 280 rows it
// gave 4 splits (396) where 5 (495 of 512) is one K-tile shorter per workgroup: 300.2 / 301.0 -> 299.9 / 299.2 ms per step, A B A B.
static int lora_down_ksplits(int M, int K, bool segmented) {
  const int m_tiles = (M + DN_BM - 1) / DN_BM + (segmented ? 1 : 0);
  const int kt = (K + 127) / 128;
  // few row blocks (decode steps, small models): the launch is one partial round whatever the split — keep the rule those paths were tuned
  // and pinned with (split until ~384 workgroups, at least two K-tiles each)
  if (lora_down_target() < 0 || m_tiles < 32) {
    int want = ((lora_down_target() < 0 ? -lora_down_target() : 384) + m_tiles - 1) / m_tiles;
    want = max(1, min(want, kt / 2));
    const int per = (kt + want - 1) / want;
    return (kt + per - 1) / per;
  }
  const int slots = lora_down_target();
  int best = 1;
  double best_cost = 1e30;
  for (int want = 1; want <= max(1, kt / 2) && want <= 64; ++want) {
    const int per = (kt + want - 1) / want;
    const int s = (kt + per - 1) / per;
    if (s != want) continue;                                   // (only split counts that are actually reached)
    const int64_t wgs = (int64_t)m_tiles * s;
    const int64_t full = wgs / slots, rest = wgs % slots;
    const double t_lat = 1.25 * per, t_bw_full = (double)slots * per * 0.00298;
    double cost = (double)full * ((t_lat > t_bw_full ? t_lat : t_bw_full) + 4.0);
    if (rest) { const double t_bw = (double)rest * per * 0.00298; cost += (t_lat > t_bw ? t_lat : t_bw) + 4.0; }
    cost += 0.15 * s;
    if (cost < best_cost) { best_cost = cost; best = s; }
  }
  return best;
}

int vm_lora_down_workspace(int M, int K, int segmented, int64_t* bytes_host) {
  if (!bytes_host || M < 0 || K <= 0) return VM_ERR_BAD_ARG;
  const int s = M > 0 ? lora_down_ksplits(M, K, segmented != 0) : 1;
  *bytes_host = s > 1 ? (int64_t)s * M * 64 * 4 : 0;
  return VM_OK;
}

int vm_lora_down(const void* x, int64_t ldx, const void* A0, const void* A1, int64_t lda, void* t, int64_t ldt,
                 int M, int K, int R, const int32_t* counts_dev, int split, float drop_p, uint64_t drop_seed,
                 void* workspace, int64_t workspace_bytes, void* stream) {
  if (!x || !A0 || !t) return VM_ERR_BAD_ARG;
  if (M <= 0) return VM_OK;
  if (R != 64 || K % 8 || K < 8 || ldx % 8 || lda % 8 || ldt % 4) return VM_ERR_UNSUPPORTED;
  if ((int64_t)32 * ldx * 2 + 256 >= (1ll << 31) || (int64_t)32 * lda * 2 + 256 >= (1ll << 31)) return VM_ERR_UNSUPPORTED;
  const bool segmented = counts_dev != nullptr || split >= 0;
  if (segmented && !A1) return VM_ERR_BAD_ARG;
  DownP p;
  p.x = (const unsigned short*)x; p.ldx = ldx;
  p.A0 = (const unsigned short*)A0; p.A1 = (const unsigned short*)(A1 ? A1 : A0); p.lda = lda;
  p.t = (unsigned short*)t; p.ldt = ldt;
  p.M = M; p.K = K;
  p.counts_dev = counts_dev;
  p.split = segmented ? (counts_dev ? 0 : split) : -1;
  p.drop_p = drop_p; p.seed = drop_seed;
  p.ksplits = lora_down_ksplits(M, K, segmented);
  p.ws = (float*)workspace;
  if (p.ksplits > 1 && (!workspace || workspace_bytes < (int64_t)p.ksplits * M * 64 * 4)) p.ksplits = 1;   // no workspace: single pass
  const int grid = (M + DN_BM - 1) / DN_BM + (segmented ? 1 : 0);
  // arrival counters of the one-launch form: a pool of zeroed regions handed out round robin (launches of one stream are ordered and every
  // launch leaves its region zeroed; two launches would have to be 64 apart AND in flight together to meet in one)
  // Measured (tools/bench_lora_down_ab.py, bit-identical results): the one-launch form wins where the launch boundary is the cost — the decode
  // step's M <= 16 rows, 9.9 vs 12.3 us x 160 projections per step — ties at [6280 x 1792] (12.0 vs 12.5) and loses 3 us where one workgroup
  // ends up summing 6-7 partials ([3648 x 4096]: 15.9 vs 12.5); inside the training step 312.0 vs 310.9 ms (A B A B). So: tiny M only.
  p.tickets = nullptr;
  unsigned* pool = lora_ticket_pool((hipStream_t)stream);          // (allocated by the first call outside a stream capture, whatever its shape)
  const int one_launch_blocks = lora_two_kernels() == 2 ? LD_TICKET_REGION : 2;
  if (pool && p.ksplits > 1 && lora_two_kernels() != 1 && grid <= one_launch_blocks && (int64_t)p.ksplits * M * 256 < (1ll << 31)) {
    static std::atomic<unsigned> seq{0};
    p.tickets = pool + (size_t)(seq++ % LD_TICKET_REGIONS) * LD_TICKET_REGION;
  }
  const int one = DN_BM * ROWB;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)lora_down_k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * one) != hipSuccess) return VM_ERR_LAUNCH;
    attr_set = true;
  }
  // (measured in round 3 and removed: a four-stage ring for single-pass grids — 320.3 vs 320.5 ms per step)
  void* tok = nullptr;
  vm_prof_begin_(VM_PROF_LORA, stream, &tok);
  hipLaunchKernelGGL((lora_down_k<2>), dim3(grid, p.ksplits), dim3(256), 2 * 2 * one, (hipStream_t)stream, p);
  if (p.ksplits > 1 && !p.tickets)
    hipLaunchKernelGGL(lora_reduce_k, dim3((unsigned)(((int64_t)M * 16 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)p.ws, p.t, p.ldt, M, p.ksplits, counts_dev);
  vm_prof_end_(VM_PROF_LORA, stream, tok, 2.0 * M * 64.0 * K);
  VM_LAUNCH_CHECK();
  return VM_OK;
}

int vm_gemm_tn_bf16(const void* X, int64_t ldx, int P, const void* Y, int64_t ldy, int Q, void* C, int64_t ldc,
                    int out_dtype, int M, const int32_t* counts_dev, int segment, const int32_t* nrows_dev, int splits,
                    float alpha, float drop_p, uint64_t drop_seed, int drop_cols, void* stream) {
  if (!X || !Y || !C) return VM_ERR_BAD_ARG;
  if (P <= 0 || Q <= 0) return VM_OK;
  if (ldx % 8 || ldy % 8 || P % 8 || Q % 8) return VM_ERR_BAD_ARG;
  if (splits < 1) splits = 1;
  if (splits > 1 && out_dtype != VM_F32) return VM_ERR_BAD_ARG;
  if ((int64_t)32 * ldx * 2 + 256 >= (1ll << 31) || (int64_t)32 * ldy * 2 + 256 >= (1ll << 31)) return VM_ERR_UNSUPPORTED;
  TnP p;
  p.X = (const unsigned short*)X; p.ldx = ldx; p.P = P;
  p.Y = (const unsigned short*)Y; p.ldy = ldy; p.Q = Q;
  p.C = C; p.ldc = ldc; p.M = M;
  p.counts_dev = counts_dev; p.segment = counts_dev ? segment : -1;
  p.nrows_dev = nrows_dev;
  p.splits = splits; p.out_f32 = out_dtype == VM_F32;
  p.drop_p = drop_p; p.seed = drop_seed; p.drop_cols = drop_cols;
  p.alpha = drop_p > 0.f ? alpha / (1.0f - drop_p) : alpha;      // the kernel only masks; inverted-dropout scale folded here
  p.tiles_p = (P + 127) / 128; p.tiles_q = (Q + 127) / 128;
  void* tok = nullptr;
  vm_prof_begin_(VM_PROF_LORA, stream, &tok);
  hipLaunchKernelGGL(gemm_tn_k, dim3(p.tiles_p * p.tiles_q, splits), dim3(256), 0, (hipStream_t)stream, p);
  vm_prof_end_(VM_PROF_LORA, stream, tok, 2.0 * (double)M * P * Q);
  VM_LAUNCH_CHECK();
  return VM_OK;
}

int vm_tn_skinny_workspace(int M, int C, int64_t* bytes_host) {
  if (!bytes_host || M <= 0 || C <= 0) return VM_ERR_BAD_ARG;
  const int c_pad = (C + 127) / 128 * 128;
  *bytes_host = 2 * (int64_t)tn_skinny_splits(M, C) * c_pad * 64 * 4;      // room for the two-segment form
  return VM_OK;
}

int vm_tn_skinny_bf16(const void* W, int64_t ldw, int C, const void* S, int64_t lds, void* out, void* out1, int64_t ldo, int out_dtype,
                      int transpose_out, int accumulate, int M, const int32_t* counts_dev, int segment,
                      const int32_t* nrows_dev, float alpha, float drop_p, uint64_t drop_seed, void* workspace,
                      int64_t workspace_bytes, void* stream) {
  if (!W || !S || !out || !workspace) return VM_ERR_BAD_ARG;
  if (C <= 0 || M <= 0) return VM_OK;
  if (ldw % 8 || lds % 8 || C % 8) return VM_ERR_BAD_ARG;
  if (out_dtype != VM_BF16 && out_dtype != VM_F32) return VM_ERR_BAD_ARG;
  if ((int64_t)32 * ldw * 2 + 256 >= (1ll << 31) || (int64_t)32 * lds * 2 + 256 >= (1ll << 31)) return VM_ERR_UNSUPPORTED;
  SkP p;
  p.W = (const unsigned short*)W; p.ldw = ldw; p.C = C;
  p.S = (const unsigned short*)S; p.lds = lds;
  p.M = M;
  p.counts_dev = counts_dev; p.segment = counts_dev ? segment : -1;
  const int nseg = p.segment == 2 ? 2 : 1;
  if (nseg == 2 && !out1) return VM_ERR_BAD_ARG;
  p.nrows_dev = nrows_dev;
  p.splits = tn_skinny_splits(M, C);
  p.c_pad = (C + 127) / 128 * 128;
  if (workspace_bytes < (int64_t)nseg * p.splits * p.c_pad * 64 * 4) return VM_ERR_BAD_ARG;
  p.ws = (float*)workspace;
  p.drop_p = drop_p; p.seed = drop_seed;
  if (drop_p > 0.f) alpha /= 1.0f - drop_p;                       // the kernel only masks; inverted-dropout scale folded here
  void* tok = nullptr;
  vm_prof_begin_(VM_PROF_LORA, stream, &tok);
  if (tn_skinny_bc(C) == 128) hipLaunchKernelGGL(tn_skinny_k<128>, dim3(p.c_pad / 128, p.splits, nseg), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(tn_skinny_k<64>, dim3(p.c_pad / 64, p.splits, nseg), dim3(256), 0, (hipStream_t)stream, p);
  const dim3 rg(p.c_pad / 16, nseg);
  if (out_dtype == VM_BF16)
    hipLaunchKernelGGL(tn_reduce_k<unsigned short>, rg, dim3(256), 0, (hipStream_t)stream, (const float*)p.ws, p.c_pad, C, p.splits,
                       (unsigned short*)out, (unsigned short*)out1, ldo, transpose_out, accumulate, alpha);
  else
    hipLaunchKernelGGL(tn_reduce_k<float>, rg, dim3(256), 0, (hipStream_t)stream, (const float*)p.ws, p.c_pad, C, p.splits,
                       (float*)out, (float*)out1, ldo, transpose_out, accumulate, alpha);
  vm_prof_end_(VM_PROF_LORA, stream, tok, 2.0 * (double)M * C * 64.0);
  VM_LAUNCH_CHECK();
  return VM_OK;
}

int vm_tn_skinny_group_bf16(const vm_tn_group_item* items_host, int n, void* stream) {
  if (!items_host || n < 0 || n > VM_TN_GROUP_MAX) return VM_ERR_BAD_ARG;
  if (n == 0) return VM_OK;
  GroupP p;
  p.n = n;
  int blocks = 0;
  double flops = 0;
  for (int i = 0; i < n; ++i) {
    vm_tn_group_item q = items_host[i];
    if (!q.W || !q.S || !q.out || q.C <= 0 || q.M < 0) return VM_ERR_BAD_ARG;
    if (q.ldw % 8 || q.lds % 8 || q.C % 8) return VM_ERR_BAD_ARG;
    if (q.out_f32 ? (!q.transpose_out && q.ldo % 4) : (!q.transpose_out && q.ldo % 8)) return VM_ERR_BAD_ARG;
    if ((int64_t)32 * q.ldw * 2 + 256 >= (1ll << 31) || (int64_t)32 * q.lds * 2 + 256 >= (1ll << 31)) return VM_ERR_UNSUPPORTED;
    if (!q.counts_dev) q.segment = -1;
    if (q.drop_p > 0.f) q.alpha /= 1.0f - q.drop_p;          // the kernel only masks; inverted-dropout scale folded here
    q.block0 = blocks;
    blocks += (q.C + GR_BC - 1) / GR_BC;
    flops += 2.0 * (double)q.M * q.C * 64.0;
    p.it[i] = q;
  }
  void* tok = nullptr;
  vm_prof_begin_(VM_PROF_LORA, stream, &tok);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)tn_group_k, hipFuncAttributeMaxDynamicSharedMemorySize, GR_STAGES * GR_STAGE_BYTES) != hipSuccess) return VM_ERR_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(tn_group_k, dim3(blocks), dim3(256), GR_STAGES * GR_STAGE_BYTES, (hipStream_t)stream, p);
  vm_prof_end_(VM_PROF_LORA, stream, tok, flops);
  VM_LAUNCH_CHECK();
  return VM_OK;
}

extern "C" int vm_gemm_f32_mode_get_(void);

int vm_gemm_tn_f32(const float* X, int64_t ldx, int P, const float* Y, int64_t ldy, int Q, float* C, int64_t ldc, int M, float* colsum,
                   int f32_split, void* stream) {
  if (!X || !Y || !C) return VM_ERR_BAD_ARG;
  if (P <= 0 || Q <= 0 || M <= 0) return VM_OK;
  if (P % 8 || Q % 8 || ldx % 4 || ldy % 4) return VM_ERR_UNSUPPORTED;
  if (f32_split < 0 || f32_split > 3) return VM_ERR_BAD_ARG;
  const int mode = f32_split == 0 ? vm_gemm_f32_mode_get_() : f32_split;
  if (mode != 2 && mode != 3) return VM_ERR_UNSUPPORTED;          // the exact f32 MFMA chain exists in the NT kernel only
  TnF32P p;
  p.X = X; p.ldx = ldx; p.P = P; p.Y = Y; p.ldy = ldy; p.Q = Q; p.C = C; p.ldc = ldc; p.M = M; p.colsum = colsum;
  const int tiles_p = (P + 63) / 64;
  p.tiles_q = (Q + 127) / 128;
  const int tiles = tiles_p * p.tiles_q;
  const int steps = (M + 31) / 32;
  // one workgroup per tile walks all rows when the tiles alone give every CU about a workgroup; otherwise split the rows
  constexpr int target = 800;
  int splits = 1;
  if (tiles < target / 2) splits = max(1, min(steps / 8, (target + tiles - 1) / tiles));
  p.splits = splits;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)gemm_tn_f32_k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * 2 * 32 * ROWB) != hipSuccess ||
        hipFuncSetAttribute((const void*)gemm_tn_f32_k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * 3 * 32 * ROWB) != hipSuccess)
      return VM_ERR_LAUNCH;
    attr_set = true;
  }
  void* tok = nullptr;
  vm_prof_begin_(VM_PROF_GEMM_F32, stream, &tok);
  if (mode == 2) hipLaunchKernelGGL(gemm_tn_f32_k<2>, dim3(tiles, splits), dim3(256), 2 * 2 * 2 * 32 * ROWB, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(gemm_tn_f32_k<3>, dim3(tiles, splits), dim3(256), 2 * 2 * 3 * 32 * ROWB, (hipStream_t)stream, p);
  vm_prof_end_(VM_PROF_GEMM_F32, stream, tok, 2.0 * (double)M * P * Q);
  VM_LAUNCH_CHECK();
  return VM_OK;
}

}  // extern "C"
