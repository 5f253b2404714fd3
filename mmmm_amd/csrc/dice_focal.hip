// Fused Dice + sigmoid-focal loss over full-resolution mask logits:
// /root/reference/mmmm/models/loss.py:13-83 (DiceFocalLoss.dice :32-40, .focal :42-56; focal = luolib.losses.sigmoid_focal_loss,
// the torchvision formula) — ~15 element-wise / reduction launches forward and ~25 backward in eager PyTorch, each a full
// pass over [P, D*H*W] fp32. Here: one streaming pass forward (4 sums per row: sum t*p, sum p, sum t, sum focal) with a
// fixed-order second stage (deterministic, no atomics), one streaming pass backward. HBM-bound: forward reads x (4 B) and
// the boolean target (1 B) per voxel, backward additionally writes dx.
//
//   p = sigmoid(x); dice_r = 1 - 2 * sum(t p) / max(sum t + sum p, 1e-8)
//   ce = max(x, 0) - x t + log1p(exp(-|x|)); p_t = p t + (1 - p)(1 - t); focal = [alpha_t] ce (1 - p_t)^gamma
#include "vm_common.hpp"

namespace {

constexpr int DF_CHUNK = 4096;         // elements per workgroup: [4 x 448 x 448] is 196 workgroups (32768 gave 28: 129 us)
constexpr float DF_EPS = 1e-8f;

struct Elem { float p, ce, pt, omp_g, w; };   // w = alpha_t (1 when alpha < 0)

__device__ __forceinline__ float df_pow(float b, float g) {      // (1 - p_t)^gamma, gamma >= 0, b in [0, 1]
  if (g == 2.f) return b * b;
  if (g == 1.f) return b;
  if (g == 0.f) return 1.f;
  return b > 0.f ? __expf(g * __logf(b)) : 0.f;
}

__device__ __forceinline__ Elem df_elem(float x, float t, float gamma, float alpha) {
  Elem e;
  const float ex = __expf(-fabsf(x));
  e.p = x >= 0.f ? 1.f / (1.f + ex) : ex / (1.f + ex);
  e.ce = fmaxf(x, 0.f) - x * t + log1pf(ex);
  e.pt = e.p * t + (1.f - e.p) * (1.f - t);
  e.omp_g = df_pow(1.f - e.pt, gamma);
  e.w = alpha >= 0.f ? alpha * t + (1.f - alpha) * (1.f - t) : 1.f;
  return e;
}

__device__ __forceinline__ float block_sum(float v, float* red) {      // 256 threads -> every thread gets the total
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// partial[row][chunk][4] = (sum t p, sum p, sum t, sum focal) over the chunk
__global__ __launch_bounds__(256) void dice_focal_partial_k(const float* __restrict__ x, const unsigned char* __restrict__ tgt,
                                                          int64_t n, float gamma, float alpha, float* __restrict__ partial,
                                                          int n_chunks) {
  __shared__ float red[4];
  const int row = blockIdx.y, chunk = blockIdx.x;
  const float* xr = x + (int64_t)row * n;
  const unsigned char* tr = tgt ? tgt + (int64_t)row * n : nullptr;
  const int64_t i0 = (int64_t)chunk * DF_CHUNK, i1 = min(n, i0 + DF_CHUNK);
  float s_tp = 0.f, s_p = 0.f, s_t = 0.f, s_f = 0.f;
  for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
    const float t = tr ? (tr[i] ? 1.f : 0.f) : 0.f;
    const Elem e = df_elem(xr[i], t, gamma, alpha);
    s_tp += t * e.p; s_p += e.p; s_t += t; s_f += e.w * e.ce * e.omp_g;
  }
  s_tp = block_sum(s_tp, red); s_p = block_sum(s_p, red); s_t = block_sum(s_t, red); s_f = block_sum(s_f, red);
  if (threadIdx.x == 0) {
    float* o = partial + ((int64_t)row * n_chunks + chunk) * 4;
    o[0] = s_tp; o[1] = s_p; o[2] = s_t; o[3] = s_f;
  }
}

// sums[row][4] = fixed-order sum of the chunk partials (thread j takes chunks j, j + 256, ...; then a fixed shuffle tree);
// out[row] = (dice, focal sum). One workgroup per row.
__global__ __launch_bounds__(256) void dice_focal_final_k(const float* __restrict__ partial, int n_chunks, int has_target,
                                                        float* __restrict__ sums, float* __restrict__ out) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
  for (int k = threadIdx.x; k < n_chunks; k += 256) {
    const float* p = partial + ((int64_t)row * n_chunks + k) * 4;
    a += p[0]; b += p[1]; c += p[2]; d += p[3];
  }
  a = block_sum(a, red); b = block_sum(b, red); c = block_sum(c, red); d = block_sum(d, red);
  if (threadIdx.x == 0) {
    sums[row * 4 + 0] = a; sums[row * 4 + 1] = b; sums[row * 4 + 2] = c; sums[row * 4 + 3] = d;
    out[row * 2 + 0] = has_target ? 1.f - 2.f * a / fmaxf(c + b, DF_EPS) : 1.f;      // no target: dice == 1 (loss.py:33-34)
    out[row * 2 + 1] = d;
  }
}

// dx = g_dice[row] * d dice/dx + g_focal[row] * d focal/dx
__global__ __launch_bounds__(256) void dice_focal_bwd_k(const float* __restrict__ x, const unsigned char* __restrict__ tgt, int64_t n,
                                                      float gamma, float alpha, const float* __restrict__ sums,
                                                      const float* __restrict__ g_dice, const float* __restrict__ g_focal,
                                                      float* __restrict__ dx) {
  const int row = blockIdx.y;
  const float* xr = x + (int64_t)row * n;
  const unsigned char* tr = tgt ? tgt + (int64_t)row * n : nullptr;
  float* dr = dx + (int64_t)row * n;
  const float inter = sums[row * 4], sp = sums[row * 4 + 1], st = sums[row * 4 + 2];
  const float den = st + sp;
  const bool clipped = !(den > DF_EPS);
  const float D = fmaxf(den, DF_EPS);
  const float gd = (tr && g_dice) ? g_dice[row] : 0.f, gf = g_focal ? g_focal[row] : 0.f;
  // d dice / d p_i = -2 t_i / D + 2 inter / D^2 (second term only while the denominator is not clipped)
  const float c1 = -2.f / D, c2 = clipped ? 0.f : 2.f * inter / (D * D);
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  for (int k = 0; k < 4; ++k) {
    const int64_t i = i0 + k;
    if (i >= n) return;
    const float t = tr ? (tr[i] ? 1.f : 0.f) : 0.f;
    const Elem e = df_elem(xr[i], t, gamma, alpha);
    const float dp = e.p * (1.f - e.p);
    const float ddice = (c1 * t + c2) * dp;
    // focal = w ce (1 - p_t)^g: d/dx = w [ (p - t)(1 - p_t)^g - ce g (1 - p_t)^(g-1) dp_t/dx ],  dp_t/dx = (2t - 1) p (1 - p)
    float dfocal = (e.p - t) * e.omp_g;
    if (gamma > 0.f) {
      const float b = 1.f - e.pt;
      const float pw = gamma == 1.f ? 1.f : (gamma == 2.f ? b : (b > 0.f ? __expf((gamma - 1.f) * __logf(b)) : 0.f));
      dfocal -= e.ce * gamma * pw * (2.f * t - 1.f) * dp;
    }
    dr[i] = gd * ddice + gf * e.w * dfocal;
  }
}

}  // namespace

extern "C" {

int vm_dice_focal_workspace(int rows, int64_t n, int64_t* bytes_host) {
  if (!bytes_host || rows < 0 || n < 0) return VM_ERR_BAD_ARG;
  const int64_t chunks = (n + DF_CHUNK - 1) / DF_CHUNK;
  *bytes_host = (int64_t)rows * (chunks > 0 ? chunks : 1) * 4 * sizeof(float);
  return VM_OK;
}

int vm_dice_focal_fwd(const float* x, const unsigned char* target, int rows, int64_t n, float gamma, float alpha, float* sums,
                      float* out, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!x || !sums || !out || !workspace || rows < 0 || n <= 0 || gamma < 0.f) return VM_ERR_BAD_ARG;
  if (rows == 0) return VM_OK;
  if (rows > 65535) return VM_ERR_UNSUPPORTED;
  int64_t need = 0;
  vm_dice_focal_workspace(rows, n, &need);
  if (workspace_bytes < need) return VM_ERR_BAD_ARG;
  const int chunks = (int)((n + DF_CHUNK - 1) / DF_CHUNK);
  if (chunks > 65535 * 32) return VM_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(dice_focal_partial_k, dim3(chunks, rows), dim3(256), 0, st, x, target, n, gamma, alpha, (float*)workspace, chunks);
  hipLaunchKernelGGL(dice_focal_final_k, dim3(rows), dim3(256), 0, st, (const float*)workspace, chunks, target ? 1 : 0, sums, out);
  return hipGetLastError() == hipSuccess ? VM_OK : VM_ERR_LAUNCH;
}

int vm_dice_focal_bwd(const float* x, const unsigned char* target, int rows, int64_t n, float gamma, float alpha, const float* sums,
                      const float* g_dice, const float* g_focal, float* dx, void* stream) {
  if (!x || !sums || !dx || rows < 0 || n <= 0 || gamma < 0.f) return VM_ERR_BAD_ARG;
  if (rows == 0) return VM_OK;
  if (rows > 65535) return VM_ERR_UNSUPPORTED;
  const int64_t blocks = (n + 1023) / 1024;
  if (blocks > 0x7FFFFFFF) return VM_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(dice_focal_bwd_k, dim3((unsigned)blocks, rows), dim3(256), 0, (hipStream_t)stream, x, target, n, gamma, alpha, sums,
                     g_dice, g_focal, dx);
  return hipGetLastError() == hipSuccess ? VM_OK : VM_ERR_LAUNCH;
}

}  // extern "C"
