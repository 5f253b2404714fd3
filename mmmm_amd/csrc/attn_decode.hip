// Single-query attention against a KV cache (generation path).
// Replaces the manual branch of attention_fn, /root/reference/mmmm/models/cogvlm/modeling_cogvlm.py:129-141.
//
// Layout (MI355X-first; the reference grows [B,H,L,hd] tensors by torch.cat every step): the cache of one layer is two
// preallocated bf16 arrays [B][max_len][H*hd] (token-major rows, the same row format as the packed qkv of the training
// path), valid rows only, and an int32 length per sample. A decode step appends one row per sample (vm_scatter_rows) and
// calls this kernel; nothing is reallocated or copied, there is no padding inside the cache and no mask.
//
// The op is HBM-bound: every K and V row of the sample is read exactly once (2 * len * H * hd * 2 bytes per sample and
// layer). One wave handles one (sample, head, key chunk): a 64-lane load covers 64*16 B = 1 KiB = 1024/(2*hd) cache rows
// of this head (hd/8 lanes per row, 16 bytes each), the q.k dot product is reduced across those lanes with DPP-free
// shuffles, and the chunk is folded with the online-softmax recurrence. Chunks of one (sample, head) are merged by a
// second, tiny kernel (flash-decoding), so a long context of a single sample still fills the 256 CUs.
//
// Rounding points follow the reference: q * hd^-0.5 is rounded to bf16 (`query_layer *= ...` in bf16), each score is
// rounded to bf16 (the einsum output), the softmax runs in fp32. The reference additionally rounds the probabilities to
// bf16 before the PV product; here they stay fp32 (the split form cannot round after normalisation) — inside the bf16
// tolerance of the parity tests and never less accurate.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vm_common.hpp"

namespace {

constexpr int DEC_CHUNK = 32;      // keys per wave: a 500-token context of ONE sample still gives 16 x n_heads waves

typedef unsigned short u16x8_t __attribute__((ext_vector_type(8)));

// LPR = lanes per cache row = hd / 8
template <int LPR>
__global__ __launch_bounds__(64) void attn_decode_k(const unsigned short* __restrict__ q, int64_t ldq,
                                                    const unsigned short* __restrict__ kc, const unsigned short* __restrict__ vc,
                                                    int64_t ld_row, int64_t ld_seq, const int32_t* __restrict__ kv_lens,
                                                    float* __restrict__ part, int n_heads, int n_chunks, float scale) {
  constexpr int HD = LPR * 8;
  constexpr int RPL = 64 / LPR;                 // cache rows per wave load
  const int chunk = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x;
  const int len = kv_lens[b];
  const int k0 = chunk * DEC_CHUNK;
  float* out = part + (((int64_t)b * n_heads + h) * n_chunks + chunk) * (HD + 2);
  if (k0 >= len) {                              // empty chunk: neutral element of the merge
    if (lane == 0) { out[0] = -INFINITY; out[1] = 0.f; }
    return;
  }
  const int k1 = min(len, k0 + DEC_CHUNK);
  const int sub = lane % LPR, rsel = lane / LPR;
  // this lane's 8 elements of q, scaled and rounded as the reference does
  float qf[8];
  {
    const u16x8_t qv = *reinterpret_cast<const u16x8_t*>(q + (int64_t)b * ldq + h * HD + sub * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) qf[e] = bf2f(f2bf(bf2f(qv[e]) * scale));
  }
  const unsigned short* kb = kc + (int64_t)b * ld_seq + h * HD + sub * 8;
  const unsigned short* vb = vc + (int64_t)b * ld_seq + h * HD + sub * 8;
  float m = -INFINITY, l = 0.f, acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  // the whole chunk is requested before anything is consumed: NLOAD K rows and NLOAD V rows per lane in flight
  constexpr int NLOAD = DEC_CHUNK / RPL;
  u16x8_t kv[NLOAD], vv[NLOAD];
#pragma unroll
  for (int i = 0; i < NLOAD; ++i) {
    const int r = k0 + i * RPL + rsel;
    kv[i] = vv[i] = (u16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
    if (r < k1) {
      kv[i] = *reinterpret_cast<const u16x8_t*>(kb + (int64_t)r * ld_row);
      vv[i] = *reinterpret_cast<const u16x8_t*>(vb + (int64_t)r * ld_row);
    }
  }
#pragma unroll
  for (int i = 0; i < NLOAD; ++i) {
    const int r0 = k0 + i * RPL;
    if (r0 >= k1) break;                                                 // wave-uniform
    const bool live = r0 + rsel < k1;
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s = fmaf(qf[e], bf2f(kv[i][e]), s);
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) s += __shfl_xor(s, o, 64);       // every lane of the row group holds the score
    s = live ? bf2f(f2bf(s)) : -INFINITY;
    // running maximum over the RPL rows of this load
    float mx = s;
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    const float m_new = fmaxf(m, mx);                                    // finite: row r0 is live
    const float corr = __expf(m - m_new);                                // m == -inf on the first load -> 0
    const float p = live ? __expf(s - m_new) : 0.f;
    l = l * corr + p;                                                    // per row group; groups are summed at the end
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = fmaf(p, bf2f(vv[i][e]), acc[e] * corr);
    m = m_new;
  }
  // fold the RPL row groups (lanes with equal `sub`)
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1) {
    l += __shfl_xor(l, o, 64);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
  }
  if (lane == 0) { out[0] = m; out[1] = l; }
  if (rsel == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) out[2 + sub * 8 + e] = acc[e];
  }
}

// merge the chunk partials of one (sample, head): one thread per output element. The chunk weights are computed once,
// cooperatively, into LDS (one independent load per thread instead of a dependent chain of 2 * n_chunks loads per thread).
constexpr int MERGE_MAX_CHUNKS = 1024;
__global__ void attn_decode_merge_k(const float* __restrict__ part, unsigned short* __restrict__ out, int64_t ldo, int n_heads,
                                    int n_chunks, int hd) {
  __shared__ float wgt[MERGE_MAX_CHUNKS];
  __shared__ float red[2];
  const int h = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
  const int stride = hd + 2;
  const float* p = part + ((int64_t)b * n_heads + h) * n_chunks * stride;
  // every thread walks the same chunk list for the maximum (n_chunks is small); the loads are independent
  float m = -INFINITY;
  for (int c = d; c < n_chunks; c += blockDim.x) m = fmaxf(m, p[c * stride]);
  // block maximum through LDS atomics-free: wave shuffles then one slot per wave (blockDim <= 128 -> 2 waves)
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((d & 63) == 0) red[d >> 6] = m;
  __syncthreads();
  m = blockDim.x > 64 ? fmaxf(red[0], red[1]) : red[0];
  for (int c = d; c < n_chunks; c += blockDim.x) {
    const float lc = p[c * stride + 1];
    wgt[c] = lc > 0.f ? __expf(p[c * stride] - m) : 0.f;        // empty chunk: weight 0, accumulator slots never written
  }
  __syncthreads();
  if (d >= hd) return;
  float l = 0.f, o = 0.f;
#pragma unroll 4
  for (int c = 0; c < n_chunks; ++c) {
    const float w = wgt[c];
    if (w == 0.f) continue;
    l += w * p[c * stride + 1];
    o += w * p[c * stride + 2 + d];
  }
  out[(int64_t)b * ldo + h * hd + d] = f2bf(l > 0.f ? o / l : 0.f);   // an empty cache row set gives zeros
}

}  // namespace

extern "C" {

int vm_attn_decode_workspace(int batch, int n_heads, int head_dim, int max_len, int64_t* bytes_host) {
  if (!bytes_host || batch < 0 || n_heads <= 0 || head_dim <= 0 || max_len < 0) return VM_ERR_BAD_ARG;
  const int n_chunks = (max_len + DEC_CHUNK - 1) / DEC_CHUNK;
  *bytes_host = (int64_t)batch * n_heads * (n_chunks > 0 ? n_chunks : 1) * (head_dim + 2) * sizeof(float);
  return VM_OK;
}

int vm_attn_decode_bf16(const void* q, int64_t ldq, const void* k_cache, const void* v_cache, int64_t ld_row, int64_t ld_seq,
                        const int32_t* kv_lens_dev, void* out, int64_t ldo, int batch, int n_heads, int head_dim, int max_len,
                        float scale, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!q || !k_cache || !v_cache || !kv_lens_dev || !out || !workspace) return VM_ERR_BAD_ARG;
  if (batch == 0) return VM_OK;
  if (batch < 0 || n_heads <= 0 || max_len <= 0 || batch > 65535 || n_heads > 65535) return VM_ERR_BAD_ARG;
  if (head_dim != 32 && head_dim != 64 && head_dim != 128) return VM_ERR_UNSUPPORTED;
  if (((uintptr_t)q | (uintptr_t)k_cache | (uintptr_t)v_cache) & 15) return VM_ERR_BAD_ARG;
  if ((ldq | ld_row | ld_seq) & 7) return VM_ERR_BAD_ARG;               // 16-byte vectors
  int64_t need = 0;
  vm_attn_decode_workspace(batch, n_heads, head_dim, max_len, &need);
  if (workspace_bytes < need) return VM_ERR_BAD_ARG;
  const int n_chunks = (max_len + DEC_CHUNK - 1) / DEC_CHUNK;
  if (n_chunks > MERGE_MAX_CHUNKS) return VM_ERR_UNSUPPORTED;          // 32k tokens
  const dim3 grid(n_chunks, n_heads, batch);
  hipStream_t st = (hipStream_t)stream;
  const unsigned short *qp = (const unsigned short*)q, *kp = (const unsigned short*)k_cache, *vp = (const unsigned short*)v_cache;
  float* part = (float*)workspace;
  if (head_dim == 128) hipLaunchKernelGGL(attn_decode_k<16>, grid, dim3(64), 0, st, qp, ldq, kp, vp, ld_row, ld_seq, kv_lens_dev, part, n_heads, n_chunks, scale);
  else if (head_dim == 64) hipLaunchKernelGGL(attn_decode_k<8>, grid, dim3(64), 0, st, qp, ldq, kp, vp, ld_row, ld_seq, kv_lens_dev, part, n_heads, n_chunks, scale);
  else hipLaunchKernelGGL(attn_decode_k<4>, grid, dim3(64), 0, st, qp, ldq, kp, vp, ld_row, ld_seq, kv_lens_dev, part, n_heads, n_chunks, scale);
  hipLaunchKernelGGL(attn_decode_merge_k, dim3(n_heads, batch), dim3(head_dim < 64 ? 64 : head_dim), 0, st, part, (unsigned short*)out, ldo, n_heads, n_chunks, head_dim);
  return hipGetLastError() == hipSuccess ? VM_OK : VM_ERR_LAUNCH;
}

}  // extern "C"
