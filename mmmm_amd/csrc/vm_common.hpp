// Shared device helpers for libvividmed_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vividmed_hip.h"

#define VM_WAVE 64

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
typedef __attribute__((ext_vector_type(2))) int i32x2_t;

#define VM_LAUNCH_CHECK()                                   \
  do {                                                      \
    hipError_t e__ = hipGetLastError();                     \
    if (e__ != hipSuccess) return VM_ERR_LAUNCH;            \
  } while (0)

// bf16 <-> f32. A plain cast lowers to v_cvt_pk_bf16_f32 (RNE, NaN-preserving) on gfx950.
__device__ __forceinline__ float bf2f(unsigned short u) {
  return __builtin_bit_cast(float, (unsigned int)u << 16);
}
__device__ __forceinline__ unsigned short f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}

// Element accessor templated on storage type: T = unsigned short (bf16 bits) or float.
template <typename T> struct Elem;
template <> struct Elem<unsigned short> {
  static constexpr int VEC = 8;  // 16 B per lane
  typedef u16x8_t vec_t;
  static __device__ __forceinline__ float ld(unsigned short v) { return bf2f(v); }
  static __device__ __forceinline__ unsigned short st(float f) { return f2bf(f); }
};
template <> struct Elem<float> {
  static constexpr int VEC = 4;
  typedef f32x4_t vec_t;
  static __device__ __forceinline__ float ld(float v) { return v; }
  static __device__ __forceinline__ float st(float f) { return f; }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for blockDim.x a multiple of 64 (<= 1024). `red` has >= 16 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float t = red[0];
  for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
  return t;
}

// Counter-based RNG for dropout masks: one hash serves FOUR consecutive elements (16 bits each), so a lane that owns 4
// or 8 consecutive elements hashes once or twice. keep(idx) is a pure function of (seed, idx, p): the standalone dropout
// kernel, the LoRA down-projection, the GEMM dgrad epilogue and the TN weight-gradient kernel all regenerate the same
// mask (forward, checkpoint recompute and backward agree).
// The per-element part is two 32-bit multiply-xorshift mixers (4 v_mul_lo_u32 per group of four elements). The seed goes
// through splitmix64 on the scalar unit (it is wave-uniform) and its four words are injected before and between the two
// rounds of each mixer, so masks of different seeds are not index permutations of one another
// (tests/test_kernels_gpu.py::test_dropout_mask_independence). Measured: the masked extension costs a 256x256 GEMM tile
// ~40 VALU instructions per four accumulators (index, hash, 4 x extract / compare / select / scale), +25 us on a
// [6280 x 15360] dgrad; the multiplies are not what bounds it (a 64-bit splitmix per group timed the same).
__device__ __forceinline__ uint64_t vm_splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ unsigned vm_mix32(unsigned x, unsigned s_in, unsigned s_mid) {
  x ^= s_in;
  x ^= x >> 16; x *= 0x7FEB352Du;
  x ^= s_mid;
  x ^= x >> 15; x *= 0x846CA68Bu;
  return x ^ (x >> 16);
}
__device__ __forceinline__ uint64_t vm_hash4(uint64_t seed, uint64_t group) {
  const uint64_t sa = vm_splitmix64(seed), sb = vm_splitmix64(seed ^ 0xD1B54A32D192ED03ull);
  const unsigned hi = (unsigned)(group >> 32);
  const unsigned k = (unsigned)group ^ (hi << 17) ^ (hi >> 3) ^ (hi * 0x9E3779B9u);       // hi == 0 below 2^34 elements
  const unsigned a = vm_mix32(k, (unsigned)sa, (unsigned)(sa >> 32));
  const unsigned b = vm_mix32(k, (unsigned)sb, (unsigned)(sb >> 32));
  return (uint64_t)a | ((uint64_t)b << 32);
}
// [r5b] The same function for tensors below 2^34 elements (every tensor of the six workloads), where the group index fits 32 bits and
// its high word contributes nothing: the seed words are split once per kernel (VmSeed), the group index is computed in 32 bits by the
// caller (row * (cols / 4) + col / 4), and the hash is two mix32 — 24 of the ~108 issue slots per 8 elements were the 64-bit index and the
// high-word fold (ISA of lora_down_k). vm_hash4w(vm_seed(seed), (unsigned)group) == vm_hash4(seed, group) for group < 2^32.
struct VmSeed { unsigned a0, a1, b0, b1; };
__device__ __forceinline__ VmSeed vm_seed(uint64_t seed) {
  const uint64_t sa = vm_splitmix64(seed), sb = vm_splitmix64(seed ^ 0xD1B54A32D192ED03ull);
  return {(unsigned)sa, (unsigned)(sa >> 32), (unsigned)sb, (unsigned)(sb >> 32)};
}
struct VmHash4 { unsigned a, b; };      // a: elements 0 (low half) and 1, b: elements 2 and 3 — 16 random bits each
__device__ __forceinline__ VmHash4 vm_hash4w(const VmSeed& s, unsigned group) {
  return {vm_mix32(group, s.a0, s.a1), vm_mix32(group, s.b0, s.b1)};
}
__device__ __forceinline__ bool vm_fits32(int64_t rows, int64_t cols) { return rows * cols < (1ll << 34); }
// The drop threshold is EVEN (p = 0.05: 3276 / 65536, unchanged): "field >= thr" is then "field >> 1 >= thr >> 1" on 15-bit values, which
// the packed 16-bit instructions decide for two elements at once (vm_mask8w below). Consequence of clearing bit 0: for a p whose
// floor(p * 65536) is odd the realised drop rate is 1 / 65536 below p while the fused 1 / (1 - p) scale uses p itself — a relative bias of
// 1.5e-5 / (1 - p) on the LoRA branch, far below the bf16 rounding of the branch's input; torch's own dropout realises p to 2^-24.
// The 32-bit group index of vm_hash4w (element index / 4) requires a row pitch that is a multiple of 4 elements: callers check `cols % 4 == 0`
// next to vm_fits32 (lora_down_k, tn_group_k) and take the 64-bit vm_hash4 form otherwise.
__device__ __forceinline__ unsigned vm_drop_threshold(float p) { return (unsigned)(p * 65536.0f) & ~1u; }
__device__ __forceinline__ bool vm_keep_bits(uint64_t h, int sub, unsigned thr) {
  return ((unsigned)(h >> (16 * sub)) & 0xFFFFu) >= thr;
}
__device__ __forceinline__ bool vm_keep(uint64_t seed, uint64_t idx, float p) {
  return vm_keep_bits(vm_hash4(seed, idx >> 2), (int)(idx & 3), vm_drop_threshold(p));
}

// Mask 8 consecutive bf16 elements (two hash words, 16 random bits per element) WITHOUT touching the kept values: dropped
// elements become +0, kept ones keep their bits. The 1/(1-p) factor of inverted dropout is linear, so the fused consumers
// (LoRA down-projection, factor-gradient kernels) apply it once to their fp32 accumulators instead of once per element —
// one AND per element pair instead of convert / multiply / round / select per element, which made those HBM-streaming
// kernels VALU-bound (+50 % on [6280 x 15360]). The result differs from dropout-then-matmul only by the bf16 rounding of
// x/(1-p) that is no longer applied (it is the more accurate of the two).
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8m_t;
__device__ __forceinline__ void vm_mask8(u16x8m_t& v, uint64_t h0, uint64_t h1, unsigned thr) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4m_t;
  u32x4m_t w = __builtin_bit_cast(u32x4m_t, v);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint64_t h = j < 2 ? h0 : h1;
    const unsigned lo = vm_keep_bits(h, (2 * j) & 3, thr) ? 0x0000FFFFu : 0u;
    const unsigned hi = vm_keep_bits(h, (2 * j + 1) & 3, thr) ? 0xFFFF0000u : 0u;
    w[j] &= lo | hi;
  }
  v = __builtin_bit_cast(u16x8m_t, w);
}
// the same mask from the 32-bit hash words, two elements per instruction: f = field >> 1 (v_pk_lshrrev_b16), d = (thr / 2 - 1) - f
// (v_pk_sub_i16: negative exactly where the element is kept — both sides are below 2^15, no wrap), mask = d >> 15 arithmetic
// (v_pk_ashrrev_i16: 0xFFFF kept, 0 dropped), one AND: 4 instructions per pair instead of 7.
__device__ __forceinline__ unsigned vm_keep_mask2(unsigned hw, unsigned thr) {
  typedef short s16x2m_t __attribute__((ext_vector_type(2)));
  typedef unsigned short u16x2m_t __attribute__((ext_vector_type(2)));
  const u16x2m_t f = __builtin_bit_cast(u16x2m_t, hw) >> (u16x2m_t){1, 1};
  const short c = (short)((int)(thr >> 1) - 1);
  const s16x2m_t d = (s16x2m_t){c, c} - __builtin_bit_cast(s16x2m_t, f);
  return __builtin_bit_cast(unsigned, d >> (s16x2m_t){15, 15});
}
__device__ __forceinline__ void vm_mask8w(u16x8m_t& v, const VmHash4& h0, const VmHash4& h1, unsigned thr) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4m_t;
  u32x4m_t w = __builtin_bit_cast(u32x4m_t, v);
  w[0] &= vm_keep_mask2(h0.a, thr);
  w[1] &= vm_keep_mask2(h0.b, thr);
  w[2] &= vm_keep_mask2(h1.a, thr);
  w[3] &= vm_keep_mask2(h1.b, thr);
  v = __builtin_bit_cast(u16x8m_t, w);
}

// bf16 GELU tables (filled by rowwise.hip gelu_table_init_k, read by gelu_tab_k): all bf16 values
// with 2^-16 <= |x| < 2^4 — 20 binades x 128 mantissas x 2 signs
constexpr int GT_E_LO = 111, GT_E_HI = 130;                       // biased exponents covered
constexpr int GT_HALF = (GT_E_HI - GT_E_LO + 1) * 128;            // entries per sign
constexpr int GT_N = 2 * GT_HALF;
__device__ __forceinline__ unsigned gt_bits_of(int idx) {         // inverse of gt_index
  const unsigned sign = idx >= GT_HALF ? 0x8000u : 0u;
  return sign | (unsigned)((idx % GT_HALF) + (GT_E_LO << 7));
}
__device__ __forceinline__ int gt_index(unsigned bits) {          // < 0: outside the table
  const unsigned rel = (bits & 0x7FFFu) - (unsigned)(GT_E_LO << 7);
  return rel < (unsigned)GT_HALF ? (int)(rel + ((bits & 0x8000u) ? GT_HALF : 0)) : -1;
}
__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
// gelu(h) / dy * gelu'(h) for the 8 bf16 values of one 16-byte chunk through the LDS copies of the tables. All eight look-ups are
// issued before any is used (no per-element control flow). Outside the table the fp32 formulas (gelu_erf / gelu_erf_grad) collapse to closed forms that give the SAME bits:
//   |x| >= 16 (finite): erff = +-1 exactly and exp(-x^2 / 2) underflows to 0 -> gelu = x | -0,  gelu' = 1 | +0;
//   2^-125 <= |x| < 2^-16: 0.5 x (1 + erf) is within 1.2e-5 relative of 0.5 x, itself a bf16 value, and rounds onto it; likewise
//     (0.5 + O(x)) dy onto 0.5 dy when that is a normal bf16 number;
//   x = +-0: +-0 and 0.5.
// What is left (bf16 denormals, inf, NaN, a denormal dy under a tiny x) takes the formula itself behind one branch per chunk.
// (The random-init benchmark model needs the large-|x| form: its residual stream grows with depth, |h| >= 16 for 9 % of the last
// blocks' pre-activations, and with the formula as the only fallback every wave ran erff for all eight elements.)
__device__ __forceinline__ void gelu_tab_fwd8(const unsigned short* tab, const u16x8_t& h, u16x8_t& o) {
  int idx[8];
  unsigned short v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) idx[e] = gt_index(h[e]);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = tab[max(idx[e], 0)];
  int lo = idx[0];
#pragma unroll
  for (int e = 1; e < 8; ++e) lo = min(lo, idx[e]);
  if (__all(lo >= 0)) {                                               // the whole wave inside the table (the normal case of a trained model)
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = v[e];
    return;
  }
  bool rare = false;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const unsigned bits = h[e], mag = bits & 0x7FFFu;
    const bool tiny = mag >= 0x100u && mag < (unsigned)(GT_E_LO << 7);
    const bool big = mag >= (unsigned)((GT_E_HI + 1) << 7) && mag < 0x7F80u;
    unsigned r = bits;                                               // +-0 -> +-0
    if (tiny) r = bits - 0x80u;                                      // x / 2
    if (big) r = (bits & 0x8000u) ? 0x8000u : bits;                  // -0 | x
    o[e] = idx[e] >= 0 ? v[e] : (unsigned short)r;
    rare |= idx[e] < 0 && !tiny && !big && mag != 0;
  }
  if (rare) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned mag = h[e] & 0x7FFFu;
      if ((mag > 0 && mag < 0x100u) || mag >= 0x7F80u) o[e] = f2bf(gelu_erf(bf2f(h[e])));
    }
  }
}
__device__ __forceinline__ void gelu_tab_bwd8(const float* tab, const u16x8_t& h, const u16x8_t& dy, u16x8_t& o) {
  int idx[8];
  float g[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) idx[e] = gt_index(h[e]);
#pragma unroll
  for (int e = 0; e < 8; ++e) g[e] = tab[max(idx[e], 0)];
  int lo = idx[0];
#pragma unroll
  for (int e = 1; e < 8; ++e) lo = min(lo, idx[e]);
  if (__all(lo >= 0)) {
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(g[e] * bf2f(dy[e]));
    return;
  }
  bool rare = false;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const unsigned bits = h[e], mag = bits & 0x7FFFu, dmag = dy[e] & 0x7FFFu;
    const bool tiny = mag >= 0x100u && mag < (unsigned)(GT_E_LO << 7);
    const bool big = mag >= (unsigned)((GT_E_HI + 1) << 7) && mag < 0x7F80u;
    float ge = 0.5f;                                                 // +-0, tiny
    if (big) ge = (bits & 0x8000u) ? 0.0f : 1.0f;
    if (idx[e] >= 0) ge = g[e];
    o[e] = f2bf(ge * bf2f(dy[e]));
    rare |= idx[e] < 0 && ((!tiny && !big && mag != 0) || (tiny && dmag != 0 && dmag < 0x100u));
  }
  if (rare) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned mag = h[e] & 0x7FFFu, dmag = dy[e] & 0x7FFFu;
      const bool tiny = mag >= 0x100u && mag < (unsigned)(GT_E_LO << 7);
      if ((mag > 0 && mag < 0x100u) || mag >= 0x7F80u || (tiny && dmag != 0 && dmag < 0x100u)) o[e] = f2bf(gelu_erf_grad(bf2f(h[e])) * bf2f(dy[e]));
    }
  }
}

