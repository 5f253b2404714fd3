// fp32 attention for the fp32 islands of VividMed (`sam`, `isam_model`: reference mmmm.py:137-138):
//   image_encoder.py:126-136   SAM ViT-B self-attention, packed var-len, head_dim 64
//   transformer.py:224-239     two-way transformer: token self-attention (head_dim 96), token->image and
//                              image->token cross-attention (head_dim 48), tiny Lq or tiny Lk
// Exact f32 arithmetic on v_mfma_f32_32x32x2_f32 (a k-ordered fmaf chain, guide §3 "FP32-input MFMA"), same
// transposed-score structure as attn_bf16.hip: S^T[kv][q] = K·Q^T puts the query on the lane, so softmax
// statistics are lane-local and the accumulator registers of S^T are directly the B operand of
// O^T[d][q] += V^T·P^T (one f32 per lane per k-step: no packing, no transposed LDS reads).
#include "vm_common.hpp"

extern "C" int vm_prof_begin_(int kind, void* stream, void** tok);
extern "C" int vm_prof_end_(int kind, void* stream, void* tok, double flops);

namespace {

constexpr float NEG_BIG = -1.0e30f;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---- split-bf16 products (NS = 2: 3 MFMAs, ~2^-17 per product; NS = 3: 6 MFMAs, fp32 products) on v_mfma_f32_32x32x16_bf16, same
// accumulator layout as v_mfma_f32_32x32x2_f32: lane (i = lane & 31, h = lane >> 5) supplies 8 consecutive k = 8 h + 0..7 of a 16-long
// k-step where the f32 instruction takes one k = h of a 2-long step. An fp32 operand x is split in registers into bf16 terms
// x = t0 + t1 (+ t2) (each subtraction exact) and the product summed from the leading cross terms, smallest first (gemm.hip, split_f32x8).
template <int NS>
__device__ __forceinline__ void split8(const float (&x0)[8], bf16x8_t (&t)[NS]) {
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = x0[e];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    bf16x8_t hh;
#pragma unroll
    for (int e = 0; e < 8; ++e) hh[e] = (__bf16)x[e];
    t[s] = hh;
    if (s + 1 < NS) {
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] -= (float)hh[e];
    }
  }
}
template <int NS>
__device__ __forceinline__ f32x16_t mma_split(const bf16x8_t (&a)[NS], const bf16x8_t (&b)[NS], f32x16_t acc) {
#pragma unroll
  for (int d = NS - 1; d >= 0; --d)
#pragma unroll
    for (int sw = 0; sw <= d; ++sw) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sw], b[d - sw], acc, 0, 0, 0);
  return acc;
}
// A-operand fragment of a k-step out of an fp32 LDS tile: 8 floats at `base + i * stride`, split
template <int NS>
__device__ __forceinline__ void lds_frag(const float* base, int stride, bf16x8_t (&t)[NS]) {
  float x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = base[i * stride];
  split8<NS>(x, t);
}
// B-operand fragment out of 8 accumulator registers acc[8 j .. 8 j + 7] (k = 8 h + i <-> row acc_row(8 j + i, h) of the tile)
template <int NS>
__device__ __forceinline__ void acc_frag(const f32x16_t& a, int j, bf16x8_t (&t)[NS]) {
  float x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = a[8 * j + i];
  split8<NS>(x, t);
}
// ---- split path, LDS side: a 32-row tile of an fp32 operand is split ONCE, by the threads that stage it, into NS bf16 planes, in the
// layouts the MFMA operands are read in (instead of every wave splitting every fragment it reads out of an fp32 tile):
//   row-major planes  [NS][32][PK]   (PK = HD + 8 bf16: 16-byte aligned rows, 144 B for HD 64): fragment = one ds_read_b128
//   transposed planes [NS][HD][PV]   (PV = 36 bf16: 8-byte aligned rows): a k-step's 8 tile rows are acc_row(8 j + i, h) = two runs of
//                                    four consecutive rows 16 j + 4 h + {0..3} and + 8: fragment = two ds_read_b64
constexpr int PV = 36;
// Two halves, so that the global loads of tile t + 1 are in flight while tile t is computed: tile_load (global -> registers, HD / 32
// float4 per thread) and tile_store (registers -> split -> LDS planes).
template <int HD>
__device__ __forceinline__ void tile_load(const float* base, int64_t ls, int head, int pos0, int len, f32x4_t (&v)[HD / 32], int tid) {
  constexpr int PER_ROW = HD / 4;
#pragma unroll
  for (int it = 0; it < HD / 32; ++it) {
    const int c = tid + it * 256;
    const int row = c / PER_ROW, ch = c % PER_ROW;
    const int pos = pos0 + row;
    v[it] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    if (pos < len) v[it] = *reinterpret_cast<const f32x4_t*>(base + (int64_t)pos * ls + head * HD + ch * 4);
  }
}
template <int HD, int NS, bool ROWM, bool TRANS>
__device__ __forceinline__ void tile_store(const f32x4_t (&v)[HD / 32], unsigned short* rowm, unsigned short* trans, int tid) {
  constexpr int PK = HD + 8, PER_ROW = HD / 4;
#pragma unroll
  for (int it = 0; it < HD / 32; ++it) {
    const int c = tid + it * 256;
    const int row = c / PER_ROW, ch = c % PER_ROW;
    float x[4] = {v[it][0], v[it][1], v[it][2], v[it][3]};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      u16x4_t t;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const __bf16 b = (__bf16)x[e];
        t[e] = __builtin_bit_cast(unsigned short, b);
        if (s + 1 < NS) x[e] -= (float)b;
      }
      if (ROWM) *reinterpret_cast<u16x4_t*>(rowm + s * 32 * PK + row * PK + ch * 4) = t;
      if (TRANS) {
#pragma unroll
        for (int e = 0; e < 4; ++e) trans[s * HD * PV + (ch * 4 + e) * PV + row] = t[e];
      }
    }
  }
}
// A fragment out of the row-major planes: row `row`, elements col0 .. col0 + 7
template <int HD, int NS>
__device__ __forceinline__ void rowm_frag(const unsigned short* rowm, int row, int col0, bf16x8_t (&t)[NS]) {
  constexpr int PK = HD + 8;
#pragma unroll
  for (int s = 0; s < NS; ++s) t[s] = *reinterpret_cast<const bf16x8_t*>(rowm + s * 32 * PK + row * PK + col0);
}
// A fragment out of the transposed planes: plane row `d` (a head-dim index), the k-step j's tile rows acc_row(8 j + i, h)
template <int HD, int NS>
__device__ __forceinline__ void trans_frag(const unsigned short* trans, int d, int j, int h, bf16x8_t (&t)[NS]) {
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const unsigned short* r = trans + s * HD * PV + d * PV + 16 * j + 4 * h;
    const u16x4_t lo = *reinterpret_cast<const u16x4_t*>(r), hi = *reinterpret_cast<const u16x4_t*>(r + 8);
    const u16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    t[s] = __builtin_bit_cast(bf16x8_t, v);
  }
}
constexpr int rowm_elems(int hd, int ns) { return ns * 32 * (hd + 8); }
constexpr int trans_elems(int hd, int ns) { return ns * hd * PV; }

// a lane's row operand (q, dO, k, v: HD floats of row `row`, this lane's half h of every 16-long k-step), split once per kernel
template <int HD, int NS>
__device__ __forceinline__ void row_frags(const float* row, bool valid, int h, bf16x8_t (&t)[HD / 16][NS]) {
#pragma unroll
  for (int s = 0; s < HD / 16; ++s) {
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = valid ? row[16 * s + 8 * h + i] : 0.f;
    split8<NS>(x, t[s]);
  }
}

// exp2 of a non-positive argument through v_exp_f32 alone: exp2f() wraps the same instruction in a range fix (v_ldexp, compares, selects)
// that only matters for results below 2^-126, which may flush to zero here (probabilities)
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }

struct AP {
  const float* q; const float* k; const float* v; float* out;
  int64_t q_bs, q_ls, k_bs, k_ls, v_bs, v_ls, o_bs, o_ls;
  float* lse;
  int Bn, Lq, Lk, n_heads;
  float scale;
  const int32_t* cu;
  const float* dout; int64_t do_bs, do_ls;
  float* dq; float* dk; float* dv;
  float* delta;
  // fp32 towers (exact form only): causal mask over sequence positions; position -> physical row of the packed operands
  int causal;
  const int32_t* rop;
};

struct Seq { int64_t qo, ko, vo, oo, doo; int lq, lk, stat0; const int32_t* rop; };

// per-(batch or sequence) base offsets and lengths
__device__ __forceinline__ Seq seq_of(const AP& p, int b) {
  Seq s;
  s.rop = nullptr;
  if (p.cu) {
    const int s0 = p.cu[b], s1 = p.cu[b + 1];
    s.lq = s.lk = s1 - s0;
    s.qo = (int64_t)s0 * p.q_ls; s.ko = (int64_t)s0 * p.k_ls; s.vo = (int64_t)s0 * p.v_ls;
    s.oo = (int64_t)s0 * p.o_ls; s.doo = (int64_t)s0 * p.do_ls;
    s.stat0 = s0;
    if (p.rop) {                 // indirect layout: position i of this sequence lives in row rop[s0 + i] of every operand
      s.rop = p.rop + s0;
      s.qo = s.ko = s.vo = s.oo = s.doo = 0;
    }
  } else {
    s.lq = p.Lq; s.lk = p.Lk;
    s.qo = (int64_t)b * p.q_bs; s.ko = (int64_t)b * p.k_bs; s.vo = (int64_t)b * p.v_bs;
    s.oo = (int64_t)b * p.o_bs; s.doo = (int64_t)b * p.do_bs;
    s.stat0 = b * p.Lq;
  }
  return s;
}
// statistics (lse, delta) are stored [H][total_q] with total_q = cu[n] or Bn*Lq
__device__ __forceinline__ int64_t stat_idx(const AP& p, int head, int stat0, int pos) {
  const int64_t total = p.cu ? (int64_t)p.cu[p.Bn] : (int64_t)p.Bn * p.Lq;
  return (int64_t)head * total + stat0 + pos;
}

__device__ __forceinline__ int64_t prow(const Seq& s, int pos) { return s.rop ? (int64_t)s.rop[pos] : (int64_t)pos; }

// Stage ROWS x HD floats of a [pos][H][HD] operand into LDS with pitch PITCH (odd => conflict-free column reads).
template <int HD, int HDP, int ROWS, int PITCH>
__device__ __forceinline__ void stage(const float* base, int64_t ls, int head, int pos0, int len, float* tile, int tid,
                                      const int32_t* rop = nullptr) {
  constexpr int PER_ROW = HDP / 4;
  for (int c = tid; c < ROWS * PER_ROW; c += 256) {
    const int row = c / PER_ROW, ch = c % PER_ROW;
    f32x4_t v = {0.f, 0.f, 0.f, 0.f};
    const int pos = pos0 + row;
    if (pos < len && ch * 4 < HD)
      v = *reinterpret_cast<const f32x4_t*>(base + (rop ? (int64_t)rop[pos] : (int64_t)pos) * ls + head * HD + ch * 4);
    float* d = tile + row * PITCH + ch * 4;
    d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
  }
}

// ----------------------------------------------------------------------------- forward
template <int HD, int NS = 0>
__global__ __launch_bounds__(256, NS == 2 ? 3 : 1) void attn_f32_fwd_k(const AP p) {
  constexpr int NSS = NS > 0 ? NS : 1, KB = NS > 0 ? HD / 16 : 1;
  constexpr int HDP = (HD + 31) / 32 * 32;
  constexpr int NB = HDP / 32;
  constexpr int KS = HD / 2;            // k-steps of the 32x32x2 MFMA over the head dimension
  constexpr int PITCH = HDP + 1;
  // NS == 0: fp32 tiles; NS > 0: K as row-major bf16 planes, V as transposed bf16 planes (stage_split)
  constexpr int LDS_A = NS > 0 ? rowm_elems(HD, NSS) * 2 : 32 * PITCH * 4, LDS_B = NS > 0 ? trans_elems(HD, NSS) * 2 : 32 * PITCH * 4;
  __shared__ __attribute__((aligned(16))) char lds_a[LDS_A];
  __shared__ __attribute__((aligned(16))) char lds_b[LDS_B];
  float* sK = reinterpret_cast<float*>(lds_a);
  float* sV = reinterpret_cast<float*>(lds_b);
  unsigned short* pK = reinterpret_cast<unsigned short*>(lds_a);
  unsigned short* pVt = reinterpret_cast<unsigned short*>(lds_b);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  const int head = blockIdx.y;
  const Seq sq = seq_of(p, blockIdx.z);
  const int q0 = blockIdx.x * 128;
  if (q0 >= sq.lq) return;
  const int qpos = q0 + wave * 32 + (lane & 31);
  const bool qvalid = qpos < sq.lq;
  const float* qrow = p.q + sq.qo + prow(sq, qvalid ? qpos : 0) * p.q_ls + head * HD;
  float qf[NS > 0 ? 1 : KS];
  bf16x8_t qs[KB][NSS];
  if constexpr (NS > 0) {
    row_frags<HD, NS>(qrow, qvalid, h, qs);
  } else {
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[s] = qvalid ? qrow[2 * s + h] : 0.f;
  }

  f32x16_t o[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[b][r] = 0.f;
  float m_run = NEG_BIG, l_run = 0.f;
  const float sc = p.scale * LOG2E;
  const bool causal = NS == 0 && p.causal;        // (the towers' fp32 mode; self-attention: key position <= query position)
  const int nt = causal ? (min(sq.lk, q0 + 128) + 31) / 32 : (sq.lk + 31) / 32;
  f32x4_t gk[NS > 0 ? HD / 32 : 1], gv[NS > 0 ? HD / 32 : 1];      // split path: the next tile's rows, in flight during the current tile
  if constexpr (NS > 0) {
    tile_load<HD>(p.k + sq.ko, p.k_ls, head, 0, sq.lk, gk, tid);
    tile_load<HD>(p.v + sq.vo, p.v_ls, head, 0, sq.lk, gv, tid);
  }
  for (int t = 0; t < nt; ++t) {
    const int kv0 = t * 32;
    __syncthreads();
    if constexpr (NS > 0) {
      tile_store<HD, NSS, true, false>(gk, pK, nullptr, tid);
      tile_store<HD, NSS, false, true>(gv, nullptr, pVt, tid);
    } else {
      stage<HD, HDP, 32, PITCH>(p.k + sq.ko, p.k_ls, head, kv0, sq.lk, sK, tid, sq.rop);
      stage<HD, HDP, 32, PITCH>(p.v + sq.vo, p.v_ls, head, kv0, sq.lk, sV, tid, sq.rop);
    }
    __syncthreads();
    if constexpr (NS > 0) {
      if (t + 1 < nt) {
        tile_load<HD>(p.k + sq.ko, p.k_ls, head, kv0 + 32, sq.lk, gk, tid);
        tile_load<HD>(p.v + sq.vo, p.v_ls, head, kv0 + 32, sq.lk, gv, tid);
      }
    }
    f32x16_t sa;
#pragma unroll
    for (int r = 0; r < 16; ++r) sa[r] = 0.f;
    if constexpr (NS > 0) {
#pragma unroll
      for (int s = 0; s < KB; ++s) {
        bf16x8_t ka[NSS];
        rowm_frag<HD, NSS>(pK, lane & 31, 16 * s + 8 * h, ka);
        sa = mma_split<NSS>(ka, qs[s], sa);
      }
    } else {
#pragma unroll
      for (int s = 0; s < KS; ++s)
        sa = __builtin_amdgcn_mfma_f32_32x32x2f32(sK[(lane & 31) * PITCH + 2 * s + h], qf[s], sa, 0, 0, 0);
    }
    float mx = NEG_BIG;
    if (causal && kv0 + 32 > q0 + wave * 32) {    // the tile reaches past this wave's first query: mask by position
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kvpos = kv0 + acc_row(r, h);
        const float x = (kvpos < sq.lk && kvpos <= qpos) ? sa[r] * sc : NEG_BIG;
        sa[r] = x;
        mx = fmaxf(mx, x);
      }
    } else if (kv0 + 32 > sq.lk) {                // only the last tile can reach past the keys (wave-uniform branch)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float x = kv0 + acc_row(r, h) < sq.lk ? sa[r] * sc : NEG_BIG;
        sa[r] = x;
        mx = fmaxf(mx, x);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float x = sa[r] * sc;
        sa[r] = x;
        mx = fmaxf(mx, x);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = fexp2(m_run - m_new);
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { const float e = fexp2(sa[r] - m_new); sa[r] = e; rs += e; }
    rs += __shfl_xor(rs, 32, 64);
    l_run = l_run * alpha + rs;
    m_run = m_new;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[b][r] *= alpha;
    }
    if constexpr (NS > 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bf16x8_t pb[NSS];
        acc_frag<NSS>(sa, j, pb);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          bf16x8_t va[NSS];
          trans_frag<HD, NSS>(pVt, 32 * b + (lane & 31), j, h, va);
          o[b] = mma_split<NSS>(va, pb, o[b]);
        }
      }
    } else {
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          o[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(sV[acc_row(r, h) * PITCH + 32 * b + (lane & 31)], sa[r], o[b], 0, 0, 0);
    }
  }
  if (!qvalid) return;
  const float inv_l = l_run > 0.f ? 1.0f / l_run : 0.f;
  if (h == 0 && p.lse) p.lse[stat_idx(p, head, sq.stat0, qpos)] = (m_run + log2f(l_run)) * LN2;
  float* orow = p.out + sq.oo + prow(sq, qpos) * p.o_ls + head * HD;
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = 32 * b + 8 * g + 4 * h;
      if (d < HD) *reinterpret_cast<f32x4_t*>(orow + d) =
          (f32x4_t){o[b][4 * g] * inv_l, o[b][4 * g + 1] * inv_l, o[b][4 * g + 2] * inv_l, o[b][4 * g + 3] * inv_l};
    }
}

// ----------------------------------------------------------------------------- delta
template <int HD>
__global__ __launch_bounds__(256) void attn_f32_delta_k(const AP p) {
  const int lane = threadIdx.x & 63;
  const int head = blockIdx.y;
  const Seq sq = seq_of(p, blockIdx.z);
  const int pos = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pos >= sq.lq) return;
  const float* a = p.dout + sq.doo + prow(sq, pos) * p.do_ls + head * HD;
  const float* b = p.out + sq.oo + prow(sq, pos) * p.o_ls + head * HD;
  float acc = 0.f;
  for (int d = lane; d < HD; d += 64) acc += a[d] * b[d];
  acc = wave_sum(acc);
  if (lane == 0) p.delta[stat_idx(p, head, sq.stat0, pos)] = acc;
}

// ----------------------------------------------------------------------------- backward dQ
template <int HD, int NS = 0>
__global__ __launch_bounds__(256, NS == 2 ? 2 : 1) void attn_f32_dq_k(const AP p) {
  constexpr int NSS = NS > 0 ? NS : 1, KB = NS > 0 ? HD / 16 : 1;
  constexpr int HDP = (HD + 31) / 32 * 32;
  constexpr int NB = HDP / 32;
  constexpr int KS = HD / 2;
  constexpr int PITCH = HDP + 1;
  // NS > 0: K as row-major AND transposed bf16 planes, V as row-major planes
  constexpr int LDS_A = NS > 0 ? (rowm_elems(HD, NSS) + trans_elems(HD, NSS)) * 2 : 32 * PITCH * 4;
  constexpr int LDS_B = NS > 0 ? rowm_elems(HD, NSS) * 2 : 32 * PITCH * 4;
  __shared__ __attribute__((aligned(16))) char lds_a[LDS_A];
  __shared__ __attribute__((aligned(16))) char lds_b[LDS_B];
  float* sK = reinterpret_cast<float*>(lds_a);
  float* sV = reinterpret_cast<float*>(lds_b);
  unsigned short* pK = reinterpret_cast<unsigned short*>(lds_a);
  unsigned short* pKt = pK + (NS > 0 ? rowm_elems(HD, NSS) : 0);
  unsigned short* pV = reinterpret_cast<unsigned short*>(lds_b);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  const int head = blockIdx.y;
  const Seq sq = seq_of(p, blockIdx.z);
  const int q0 = blockIdx.x * 128;
  if (q0 >= sq.lq) return;
  const int qpos = q0 + wave * 32 + (lane & 31);
  const bool qvalid = qpos < sq.lq;
  const float* qrow = p.q + sq.qo + prow(sq, qvalid ? qpos : 0) * p.q_ls + head * HD;
  const float* dorow = p.dout + sq.doo + prow(sq, qvalid ? qpos : 0) * p.do_ls + head * HD;
  float qf[NS > 0 ? 1 : KS], dof[NS > 0 ? 1 : KS];
  bf16x8_t qs[KB][NSS], dos[KB][NSS];
  if constexpr (NS > 0) {
    row_frags<HD, NS>(qrow, qvalid, h, qs);
    row_frags<HD, NS>(dorow, qvalid, h, dos);
  } else {
#pragma unroll
    for (int s = 0; s < KS; ++s) { qf[s] = qvalid ? qrow[2 * s + h] : 0.f; dof[s] = qvalid ? dorow[2 * s + h] : 0.f; }
  }
  const float lse2 = qvalid ? p.lse[stat_idx(p, head, sq.stat0, qpos)] * LOG2E : 0.f;
  const float dlt = qvalid ? p.delta[stat_idx(p, head, sq.stat0, qpos)] : 0.f;
  const float sc = p.scale * LOG2E;
  f32x16_t dq[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[b][r] = 0.f;
  const bool causal = NS == 0 && p.causal;
  const int nt = causal ? (min(sq.lk, q0 + 128) + 31) / 32 : (sq.lk + 31) / 32;
  f32x4_t gk[NS > 0 ? HD / 32 : 1], gv[NS > 0 ? HD / 32 : 1];
  if constexpr (NS > 0) {
    tile_load<HD>(p.k + sq.ko, p.k_ls, head, 0, sq.lk, gk, tid);
    tile_load<HD>(p.v + sq.vo, p.v_ls, head, 0, sq.lk, gv, tid);
  }
  for (int t = 0; t < nt; ++t) {
    const int kv0 = t * 32;
    __syncthreads();
    if constexpr (NS > 0) {
      tile_store<HD, NSS, true, true>(gk, pK, pKt, tid);
      tile_store<HD, NSS, true, false>(gv, pV, nullptr, tid);
    } else {
      stage<HD, HDP, 32, PITCH>(p.k + sq.ko, p.k_ls, head, kv0, sq.lk, sK, tid, sq.rop);
      stage<HD, HDP, 32, PITCH>(p.v + sq.vo, p.v_ls, head, kv0, sq.lk, sV, tid, sq.rop);
    }
    __syncthreads();
    if constexpr (NS > 0) {
      if (t + 1 < nt) {
        tile_load<HD>(p.k + sq.ko, p.k_ls, head, kv0 + 32, sq.lk, gk, tid);
        tile_load<HD>(p.v + sq.vo, p.v_ls, head, kv0 + 32, sq.lk, gv, tid);
      }
    }
    f32x16_t sa, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sa[r] = 0.f; dp[r] = 0.f; }
    if constexpr (NS > 0) {
#pragma unroll
      for (int s = 0; s < KB; ++s) {
        bf16x8_t ka[NSS], va[NSS];
        rowm_frag<HD, NSS>(pK, lane & 31, 16 * s + 8 * h, ka);
        rowm_frag<HD, NSS>(pV, lane & 31, 16 * s + 8 * h, va);
        sa = mma_split<NSS>(ka, qs[s], sa);
        dp = mma_split<NSS>(va, dos[s], dp);
      }
    } else {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        sa = __builtin_amdgcn_mfma_f32_32x32x2f32(sK[(lane & 31) * PITCH + 2 * s + h], qf[s], sa, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(sV[(lane & 31) * PITCH + 2 * s + h], dof[s], dp, 0, 0, 0);
      }
    }
    if (kv0 + 32 > sq.lk || !qvalid || (causal && kv0 + 32 > q0 + wave * 32)) {   // the last key tile, a query row past the sequence (its lanes' lse2 is not a statistic), the causal diagonal
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kvpos = kv0 + acc_row(r, h);
        const float pr = (kvpos < sq.lk && qvalid && (!causal || kvpos <= qpos)) ? fexp2(sa[r] * sc - lse2) : 0.f;
        sa[r] = pr * (dp[r] - dlt) * p.scale;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) sa[r] = fexp2(sa[r] * sc - lse2) * (dp[r] - dlt) * p.scale;
    }
    if constexpr (NS > 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bf16x8_t sb[NSS];
        acc_frag<NSS>(sa, j, sb);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          bf16x8_t ka[NSS];
          trans_frag<HD, NSS>(pKt, 32 * b + (lane & 31), j, h, ka);
          dq[b] = mma_split<NSS>(ka, sb, dq[b]);
        }
      }
    } else {
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          dq[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(sK[acc_row(r, h) * PITCH + 32 * b + (lane & 31)], sa[r], dq[b], 0, 0, 0);
    }
  }
  if (!qvalid) return;
  float* drow = p.dq + sq.qo + prow(sq, qpos) * p.q_ls + head * HD;
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = 32 * b + 8 * g + 4 * h;
      if (d < HD) *reinterpret_cast<f32x4_t*>(drow + d) = (f32x4_t){dq[b][4 * g], dq[b][4 * g + 1], dq[b][4 * g + 2], dq[b][4 * g + 3]};
    }
}

// ----------------------------------------------------------------------------- backward dK, dV
template <int HD, int NS = 0>
__global__ __launch_bounds__(256, NS == 2 ? 2 : 1) void attn_f32_dkv_k(const AP p) {
  constexpr int NSS = NS > 0 ? NS : 1, KB = NS > 0 ? HD / 16 : 1;
  constexpr int HDP = (HD + 31) / 32 * 32;
  constexpr int NB = HDP / 32;
  constexpr int KS = HD / 2;
  constexpr int PITCH = HDP + 1;
  // NS > 0: Q and dO as row-major AND transposed bf16 planes
  constexpr int LDS_A = NS > 0 ? (rowm_elems(HD, NSS) + trans_elems(HD, NSS)) * 2 : 32 * PITCH * 4;
  __shared__ __attribute__((aligned(16))) char lds_a[LDS_A];
  __shared__ __attribute__((aligned(16))) char lds_b[LDS_A];
  float* sQ = reinterpret_cast<float*>(lds_a);
  float* sDO = reinterpret_cast<float*>(lds_b);
  unsigned short* pQ = reinterpret_cast<unsigned short*>(lds_a);
  unsigned short* pQt = pQ + (NS > 0 ? rowm_elems(HD, NSS) : 0);
  unsigned short* pDO = reinterpret_cast<unsigned short*>(lds_b);
  unsigned short* pDOt = pDO + (NS > 0 ? rowm_elems(HD, NSS) : 0);
  __shared__ float sLse[32];
  __shared__ float sDlt[32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  const int head = blockIdx.y;
  const Seq sq = seq_of(p, blockIdx.z);
  const int k0 = blockIdx.x * 128;
  if (k0 >= sq.lk) return;
  const int kpos = k0 + wave * 32 + (lane & 31);
  const bool kvalid = kpos < sq.lk;
  const float* krow = p.k + sq.ko + prow(sq, kvalid ? kpos : 0) * p.k_ls + head * HD;
  const float* vrow = p.v + sq.vo + prow(sq, kvalid ? kpos : 0) * p.v_ls + head * HD;
  float kf[NS > 0 ? 1 : KS], vf[NS > 0 ? 1 : KS];
  bf16x8_t ks[KB][NSS], vs[KB][NSS];
  if constexpr (NS > 0) {
    row_frags<HD, NS>(krow, kvalid, h, ks);
    row_frags<HD, NS>(vrow, kvalid, h, vs);
  } else {
#pragma unroll
    for (int s = 0; s < KS; ++s) { kf[s] = kvalid ? krow[2 * s + h] : 0.f; vf[s] = kvalid ? vrow[2 * s + h] : 0.f; }
  }
  const float sc = p.scale * LOG2E;
  f32x16_t dk[NB], dv[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[b][r] = 0.f; dv[b][r] = 0.f; }
  const bool causal = NS == 0 && p.causal;
  const int nt = (sq.lq + 31) / 32;
  const int t_first = causal ? k0 / 32 : 0;       // queries in front of this workgroup's first key see none of its keys
  f32x4_t gq[NS > 0 ? HD / 32 : 1], gd[NS > 0 ? HD / 32 : 1];
  if constexpr (NS > 0) {
    tile_load<HD>(p.q + sq.qo, p.q_ls, head, 0, sq.lq, gq, tid);
    tile_load<HD>(p.dout + sq.doo, p.do_ls, head, 0, sq.lq, gd, tid);
  }
  for (int t = t_first; t < nt; ++t) {
    const int qq0 = t * 32;
    __syncthreads();
    if constexpr (NS > 0) {
      tile_store<HD, NSS, true, true>(gq, pQ, pQt, tid);
      tile_store<HD, NSS, true, true>(gd, pDO, pDOt, tid);
    } else {
      stage<HD, HDP, 32, PITCH>(p.q + sq.qo, p.q_ls, head, qq0, sq.lq, sQ, tid, sq.rop);
      stage<HD, HDP, 32, PITCH>(p.dout + sq.doo, p.do_ls, head, qq0, sq.lq, sDO, tid, sq.rop);
    }
    if (tid < 32) {
      const int qp = qq0 + tid;
      sLse[tid] = qp < sq.lq ? p.lse[stat_idx(p, head, sq.stat0, qp)] * LOG2E : 0.f;
      sDlt[tid] = qp < sq.lq ? p.delta[stat_idx(p, head, sq.stat0, qp)] : 0.f;
    }
    __syncthreads();
    if constexpr (NS > 0) {
      if (t + 1 < nt) {
        tile_load<HD>(p.q + sq.qo, p.q_ls, head, qq0 + 32, sq.lq, gq, tid);
        tile_load<HD>(p.dout + sq.doo, p.do_ls, head, qq0 + 32, sq.lq, gd, tid);
      }
    }
    f32x16_t sa, dp, pa;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sa[r] = 0.f; dp[r] = 0.f; }
    if constexpr (NS > 0) {
#pragma unroll
      for (int s = 0; s < KB; ++s) {
        bf16x8_t qa[NSS], da[NSS];
        rowm_frag<HD, NSS>(pQ, lane & 31, 16 * s + 8 * h, qa);
        rowm_frag<HD, NSS>(pDO, lane & 31, 16 * s + 8 * h, da);
        sa = mma_split<NSS>(qa, ks[s], sa);
        dp = mma_split<NSS>(da, vs[s], dp);
      }
    } else {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        sa = __builtin_amdgcn_mfma_f32_32x32x2f32(sQ[(lane & 31) * PITCH + 2 * s + h], kf[s], sa, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(sDO[(lane & 31) * PITCH + 2 * s + h], vf[s], dp, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi = acc_row(r, h);
      const bool vis = kvalid && (qq0 + qi) < sq.lq && (!causal || kpos <= qq0 + qi);
      const float pr = vis ? fexp2(sa[r] * sc - sLse[qi]) : 0.f;
      pa[r] = pr;
      sa[r] = pr * (dp[r] - sDlt[qi]) * p.scale;
    }
    if constexpr (NS > 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bf16x8_t pb[NSS], sb[NSS];
        acc_frag<NSS>(pa, j, pb);
        acc_frag<NSS>(sa, j, sb);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          bf16x8_t da[NSS], qa[NSS];
          trans_frag<HD, NSS>(pDOt, 32 * b + (lane & 31), j, h, da);
          trans_frag<HD, NSS>(pQt, 32 * b + (lane & 31), j, h, qa);
          dv[b] = mma_split<NSS>(da, pb, dv[b]);
          dk[b] = mma_split<NSS>(qa, sb, dk[b]);
        }
      }
    } else {
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          dv[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(sDO[acc_row(r, h) * PITCH + 32 * b + (lane & 31)], pa[r], dv[b], 0, 0, 0);
          dk[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(sQ[acc_row(r, h) * PITCH + 32 * b + (lane & 31)], sa[r], dk[b], 0, 0, 0);
        }
    }
  }
  if (!kvalid) return;
  float* dkrow = p.dk + sq.ko + prow(sq, kpos) * p.k_ls + head * HD;
  float* dvrow = p.dv + sq.vo + prow(sq, kpos) * p.v_ls + head * HD;
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = 32 * b + 8 * g + 4 * h;
      if (d < HD) {
        *reinterpret_cast<f32x4_t*>(dkrow + d) = (f32x4_t){dk[b][4 * g], dk[b][4 * g + 1], dk[b][4 * g + 2], dk[b][4 * g + 3]};
        *reinterpret_cast<f32x4_t*>(dvrow + d) = (f32x4_t){dv[b][4 * g], dv[b][4 * g + 1], dv[b][4 * g + 2], dv[b][4 * g + 3]};
      }
    }
}

AP to_ap(const vm_attn_f32_args* a) {
  AP p;
  p.q = a->q; p.k = a->k; p.v = a->v; p.out = a->out;
  p.q_bs = a->q_bs; p.q_ls = a->q_ls; p.k_bs = a->k_bs; p.k_ls = a->k_ls; p.v_bs = a->v_bs; p.v_ls = a->v_ls;
  p.o_bs = a->o_bs; p.o_ls = a->o_ls;
  p.lse = a->lse; p.Bn = a->cu_seqlens ? a->n_seq : a->Bn; p.Lq = a->Lq; p.Lk = a->Lk; p.n_heads = a->n_heads;
  p.scale = a->scale; p.cu = a->cu_seqlens;
  p.dout = a->dout; p.do_bs = a->do_bs; p.do_ls = a->do_ls;
  p.dq = a->dq; p.dk = a->dk; p.dv = a->dv; p.delta = a->delta;
  p.causal = a->causal; p.rop = a->row_of_pos;
  return p;
}

bool ok(const vm_attn_f32_args* a) {
  if (!a || !a->q || !a->k || !a->v || !a->out || !a->lse) return false;
  if (a->n_heads <= 0 || a->Lq <= 0 || a->Lk <= 0) return false;
  if (a->cu_seqlens ? a->n_seq <= 0 : a->Bn <= 0) return false;
  if (a->q_ls % 4 || a->k_ls % 4 || a->v_ls % 4 || a->o_ls % 4) return false;
  // causal / row_of_pos: packed self-attention in the exact arithmetic (the towers' fp32 mode)
  if ((a->causal || a->row_of_pos) && (!a->cu_seqlens || (a->f32_split >= 2 && a->head_dim == 64))) return false;
  return true;
}

}  // namespace

#define F32_DISPATCH_HD(hd, ...)                                  \
  switch (hd) {                                                   \
    case 128: { constexpr int HD = 128; __VA_ARGS__; break; }     \
    case 112: { constexpr int HD = 112; __VA_ARGS__; break; }     \
    case 96:  { constexpr int HD = 96;  __VA_ARGS__; break; }     \
    case 64:  { constexpr int HD = 64;  __VA_ARGS__; break; }     \
    case 48:  { constexpr int HD = 48;  __VA_ARGS__; break; }     \
    case 32:  { constexpr int HD = 32;  __VA_ARGS__; break; }     \
    case 16:  { constexpr int HD = 16;  __VA_ARGS__; break; }     \
    case 8:   { constexpr int HD = 8;   __VA_ARGS__; break; }     \
    default: return VM_ERR_UNSUPPORTED;                           \
  }

extern "C" {

int vm_attn_fwd_f32(const vm_attn_f32_args* a, void* stream) {
  if (!ok(a)) return VM_ERR_BAD_ARG;
  AP p = to_ap(a);
  dim3 grid((a->Lq + 127) / 128, a->n_heads, p.Bn);
  void* tok = nullptr;
  // f32_split (head_dim 64, the SAM ViT-B encoders): the products on split-bf16 MFMAs instead of the exact f32 MFMA chain
  if (a->f32_split < 0 || a->f32_split > 3) return VM_ERR_BAD_ARG;          // (argument checks come before the profiling bracket opens)
  vm_prof_begin_(VM_PROF_ATTN, stream, &tok);
  if (a->f32_split >= 2 && a->head_dim == 64) {
    if (a->f32_split == 2) hipLaunchKernelGGL((attn_f32_fwd_k<64, 2>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((attn_f32_fwd_k<64, 3>), grid, dim3(256), 0, (hipStream_t)stream, p);
  } else {
    F32_DISPATCH_HD(a->head_dim, hipLaunchKernelGGL(attn_f32_fwd_k<HD>, grid, dim3(256), 0, (hipStream_t)stream, p));
  }
  vm_prof_end_(VM_PROF_ATTN, stream, tok, 4.0 * a->Lq * (double)a->Lk * a->head_dim * a->n_heads * p.Bn);
  VM_LAUNCH_CHECK();
  return VM_OK;
}

int vm_attn_bwd_f32(const vm_attn_f32_args* a, void* stream) {
  if (!ok(a) || !a->dout || !a->dq || !a->dk || !a->dv || !a->delta) return VM_ERR_BAD_ARG;
  if (a->do_ls % 4) return VM_ERR_BAD_ARG;
  AP p = to_ap(a);
  dim3 gq((a->Lq + 127) / 128, a->n_heads, p.Bn), gk((a->Lk + 127) / 128, a->n_heads, p.Bn);
  dim3 gd((a->Lq + 3) / 4, a->n_heads, p.Bn);
  void* tok = nullptr;
  if (a->f32_split < 0 || a->f32_split > 3) return VM_ERR_BAD_ARG;          // (argument checks come before the profiling bracket opens)
  vm_prof_begin_(VM_PROF_ATTN, stream, &tok);
  if (a->f32_split >= 2 && a->head_dim == 64) {
    hipLaunchKernelGGL(attn_f32_delta_k<64>, gd, dim3(256), 0, (hipStream_t)stream, p);
    if (a->f32_split == 2) {
      hipLaunchKernelGGL((attn_f32_dq_k<64, 2>), gq, dim3(256), 0, (hipStream_t)stream, p);
      hipLaunchKernelGGL((attn_f32_dkv_k<64, 2>), gk, dim3(256), 0, (hipStream_t)stream, p);
    } else {
      hipLaunchKernelGGL((attn_f32_dq_k<64, 3>), gq, dim3(256), 0, (hipStream_t)stream, p);
      hipLaunchKernelGGL((attn_f32_dkv_k<64, 3>), gk, dim3(256), 0, (hipStream_t)stream, p);
    }
  } else {
    F32_DISPATCH_HD(a->head_dim,
                    hipLaunchKernelGGL(attn_f32_delta_k<HD>, gd, dim3(256), 0, (hipStream_t)stream, p);
                    hipLaunchKernelGGL(attn_f32_dq_k<HD>, gq, dim3(256), 0, (hipStream_t)stream, p);
                    hipLaunchKernelGGL(attn_f32_dkv_k<HD>, gk, dim3(256), 0, (hipStream_t)stream, p));
  }
  vm_prof_end_(VM_PROF_ATTN, stream, tok, 10.0 * a->Lq * (double)a->Lk * a->head_dim * a->n_heads * p.Bn);
  VM_LAUNCH_CHECK();
  return VM_OK;
}

}  // extern "C"
