// Skinny-M linear for the decode step of the generation path: out[M<=16, N] = x W^T (+ LoRA) (+ bias)
// (+ residual), i.e. the language-expert nn.Linear calls of VisionExpertAttention / MLP / lm_head with one row per sample
// (/root/reference/mmmm/models/cogvlm/modeling_cogvlm.py:243-245, 277-279, 54-56, 706).
//
// HBM-bound: W (N*K*2 bytes) is read exactly once and nothing else matters. The tiled GEMM kernels are the wrong shape
// for it (a 128-row tile per workgroup gives N/128 = 32..96 workgroups for 256 CUs and a K loop with one tile in flight:
// measured 0.85 TB/s over a decode step). Here a workgroup owns 16 output columns and ALL of K: its 16 waves stride over
// K in 64-wide blocks, every lane streams W straight from global memory into MFMA operand registers (no LDS staging —
// there is no reuse), 32 contiguous bytes per row and lane, 128 contiguous bytes per row and wave instruction pair, with
// UNROLL blocks in flight per wave. x (M rows, L2-resident) is loaded the same way as the other MFMA operand, rows >= M
// as zeros. The 16 partial 16x16 tiles are summed through LDS in a fixed order (deterministic), then wave 0 applies the
// LoRA extension (two more MFMA blocks over [t | B2], K2 = 64), bias and residual with torch's bf16 rounding points.
//
// The k-index permutation inside a 64-block (lane group q covers k = 16q .. 16q+15, split over two MFMAs) is applied to
// both operands, so the contraction is unchanged.
#include "vm_common.hpp"

namespace {

constexpr int GV_WAVES = 16;
constexpr int GV_UNROLL = 4;

__device__ __forceinline__ bf16x8_t ld8(const unsigned short* p) { return *reinterpret_cast<const bf16x8_t*>(p); }
__device__ __forceinline__ bf16x8_t zero8() {
  const f32x4_t z = {0.f, 0.f, 0.f, 0.f};
  return __builtin_bit_cast(bf16x8_t, z);
}

// acc += X[16 x 64] . W[16 x 64]^T for the 64-block starting at k0 (both row pointers already include the lane's row)
__device__ __forceinline__ void block64(f32x4_t& acc, const unsigned short* wrow, const unsigned short* xrow, bool wlive, bool xlive,
                                        int k0, int q) {
  const int k = k0 + q * 16;
  const bf16x8_t w0 = wlive ? ld8(wrow + k) : zero8(), w1 = wlive ? ld8(wrow + k + 8) : zero8();
  const bf16x8_t x0 = xlive ? ld8(xrow + k) : zero8(), x1 = xlive ? ld8(xrow + k + 8) : zero8();
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, x0, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, x1, acc, 0, 0, 0);
}

__global__ __launch_bounds__(GV_WAVES * 64) void gemv_k(const unsigned short* __restrict__ x, int64_t ldx,
                                                       const unsigned short* __restrict__ W, int64_t ldw,
                                                       const unsigned short* __restrict__ x2, int64_t ldx2,
                                                       const unsigned short* __restrict__ W2, int64_t ldw2, float alpha2,
                                                       const unsigned short* __restrict__ bias,
                                                       const unsigned short* __restrict__ residual, int64_t ldr,
                                                       unsigned short* __restrict__ out, int64_t ldo, int M, int N, int K, int K2) {
  __shared__ f32x4_t part[GV_WAVES][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const bool wlive = n0 + r < N, xlive = r < M;
  const unsigned short* wrow = W + (int64_t)(n0 + r) * ldw;
  const unsigned short* xrow = x + (int64_t)r * ldx;
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  const int nblk = K / 64;
  int b = wave;
  for (; b + (GV_UNROLL - 1) * GV_WAVES < nblk; b += GV_UNROLL * GV_WAVES) {
#pragma unroll
    for (int u = 0; u < GV_UNROLL; ++u) block64(acc, wrow, xrow, wlive, xlive, (b + u * GV_WAVES) * 64, q);
  }
  for (; b < nblk; b += GV_WAVES) block64(acc, wrow, xrow, wlive, xlive, b * 64, q);
  part[wave][lane] = acc;
  __syncthreads();
  if (wave != 0) return;
  f32x4_t sum = part[0][lane];
#pragma unroll
  for (int w = 1; w < GV_WAVES; ++w) sum += part[w][lane];
  if (K2 > 0) {                                   // LoRA extension: + alpha2 * x2 . W2^T (K2 % 64 == 0)
    f32x4_t ext = {0.f, 0.f, 0.f, 0.f};
    const unsigned short* w2row = W2 + (int64_t)(n0 + r) * ldw2;
    const unsigned short* x2row = x2 + (int64_t)r * ldx2;
    for (int k0 = 0; k0 < K2; k0 += 64) block64(ext, w2row, x2row, wlive, xlive, k0, q);
    sum += ext * alpha2;
  }
  // lane owns out[m = r][n0 + 4q .. n0 + 4q + 3]
  const int m = r, n = n0 + q * 4;
  if (m >= M || n >= N) return;
  unsigned short o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float v = sum[e];
    if (n + e < N) {
      if (bias) v += bf2f(bias[n + e]);
      if (residual) v = bf2f(f2bf(v)) + bf2f(residual[(int64_t)m * ldr + n + e]);    // torch rounds the linear before the add
    }
    o[e] = f2bf(v);
  }
  unsigned short* cp = out + (int64_t)m * ldo + n;
  if (n + 3 < N && (((uintptr_t)cp) & 7) == 0) *reinterpret_cast<u16x4_t*>(cp) = (u16x4_t){o[0], o[1], o[2], o[3]};
  else for (int e = 0; e < 4 && n + e < N; ++e) cp[e] = o[e];
}

}  // namespace

extern "C" {

int vm_gemv_bf16(const void* x, int64_t ldx, const void* W, int64_t ldw, const void* x2, int64_t ldx2, const void* W2, int64_t ldw2,
                 float alpha2, const void* bias, const void* residual, int64_t ldr, void* out, int64_t ldo, int M, int N, int K,
                 int K2, void* stream) {
  if (!x || !W || !out || M < 0 || N < 0 || K <= 0 || K2 < 0) return VM_ERR_BAD_ARG;
  if (M == 0 || N == 0) return VM_OK;
  if (M > 16 || K % 64 != 0 || K2 % 64 != 0) return VM_ERR_UNSUPPORTED;
  if (K2 > 0 && (!x2 || !W2)) return VM_ERR_BAD_ARG;
  if ((((uintptr_t)x | (uintptr_t)W | (uintptr_t)x2 | (uintptr_t)W2) & 15) || ((ldx | ldw | ldx2 | ldw2) & 7)) return VM_ERR_BAD_ARG;
  hipLaunchKernelGGL(gemv_k, dim3((N + 15) / 16), dim3(GV_WAVES * 64), 0, (hipStream_t)stream, (const unsigned short*)x, ldx,
                     (const unsigned short*)W, ldw, (const unsigned short*)x2, ldx2, (const unsigned short*)W2, ldw2, alpha2,
                     (const unsigned short*)bias, (const unsigned short*)residual, ldr, (unsigned short*)out, ldo, M, N, K, K2);
  return hipGetLastError() == hipSuccess ? VM_OK : VM_ERR_LAUNCH;
}

}  // extern "C"
