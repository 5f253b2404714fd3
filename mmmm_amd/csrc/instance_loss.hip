// Instance-grounding (iSAM) losses of one sample, and the Hungarian cost matrices of all samples, as single launches.
//
// /root/reference/mmmm/models/segvol/modeling/sam.py:
//   box_loss :148-160      total = l1_w * l1_loss(input, target) + giou_w * (1 - box_pair_giou(cc(input), cc(target)))
//                          (boxes as centre-size in [0, 1]; monai.data.box_utils.box_pair_giou [external], fp32, eps = FLT_EPSILON)
//   disc_loss :162-176     disc_w * sigmoid_focal_loss(logit, label, gamma, alpha)   (luolib [external]: the torchvision formula)
//   _match_instances :178-250   cost[q, j] = box_loss(reg[q], label[j]) + disc cost of a positive, j < n_pos; the disc cost of a
//                          negative on the dummy columns that make the matrix square
//   compute_loss :252-361  certain-entries focal mean, matched-pair box loss, the positive / negative focal means for the log
//
// The eager form of these is ~45 element-wise launches per cost matrix and ~150 forward + ~300 backward launches per sample for
// the losses — a few hundred elements each, ~7 us of step time per launch (measured: memoising the cost matrices alone takes
// 4 ms off a 351 ms step). Here a workgroup handles one cost matrix / one sample's whole loss; every sum is reduced in a fixed
// order (deterministic).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>

#include "vm_common.hpp"
#include "../../include/vividmed_hip.h"

namespace {

struct Focal { float p, ce, pt, omp_g, w; };

__device__ __forceinline__ float fpow(float b, float g) {      // (1 - p_t)^gamma, gamma >= 0, b in [0, 1]
  if (g == 2.f) return b * b;
  if (g == 1.f) return b;
  if (g == 0.f) return 1.f;
  return b > 0.f ? __expf(g * __logf(b)) : 0.f;
}

__device__ __forceinline__ Focal focal_elem(float x, float t, float gamma, float alpha) {
  Focal e;
  const float ex = __expf(-fabsf(x));
  e.p = x >= 0.f ? 1.f / (1.f + ex) : ex / (1.f + ex);
  e.ce = fmaxf(x, 0.f) - x * t + log1pf(ex);
  e.pt = e.p * t + (1.f - e.p) * (1.f - t);
  e.omp_g = fpow(1.f - e.pt, gamma);
  e.w = alpha >= 0.f ? alpha * t + (1.f - alpha) * (1.f - t) : 1.f;
  return e;
}

__device__ __forceinline__ float focal_value(float x, float t, float gamma, float alpha) {
  const Focal e = focal_elem(x, t, gamma, alpha);
  return e.w * e.ce * e.omp_g;
}

// d focal / dx = w [ (p - t)(1 - p_t)^g - ce g (1 - p_t)^(g-1) (2t - 1) p (1 - p) ]
__device__ __forceinline__ float focal_grad(float x, float t, float gamma, float alpha) {
  const Focal e = focal_elem(x, t, gamma, alpha);
  float d = (e.p - t) * e.omp_g;
  if (gamma > 0.f) d -= e.ce * gamma * fpow(1.f - e.pt, gamma - 1.f) * (2.f * t - 1.f) * e.p * (1.f - e.p);
  return e.w * d;
}

struct Box { float lo[3], hi[3]; };

__device__ __forceinline__ Box cs_to_cc(const float* b) {      // box_cs_to_cc: centre-size -> corners
  Box r;
#pragma unroll
  for (int k = 0; k < 3; ++k) { const float h = b[3 + k] / 2.f; r.lo[k] = b[k] - h; r.hi[k] = b[k] + h; }
  return r;
}

__device__ __forceinline__ float vol3(const float* e) { return e[0] * e[1] * e[2]; }

struct Giou { float w[3], v[3], wr[3], vr[3], e1[3], inter, uni, enc, value; };

__device__ __forceinline__ Giou giou_pair(const Box& a, const Box& b) {
  Giou g;
  float e2[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    g.e1[k] = a.hi[k] - a.lo[k];
    e2[k] = b.hi[k] - b.lo[k];
    g.wr[k] = fminf(a.hi[k], b.hi[k]) - fmaxf(a.lo[k], b.lo[k]);
    g.vr[k] = fmaxf(a.hi[k], b.hi[k]) - fminf(a.lo[k], b.lo[k]);
    g.w[k] = fmaxf(g.wr[k], 0.f);
    g.v[k] = fmaxf(g.vr[k], 0.f);
  }
  const float a1 = vol3(g.e1), a2 = vol3(e2);
  g.inter = vol3(g.w);
  g.uni = a1 + a2 - g.inter;
  g.enc = vol3(g.v);
  g.value = g.inter / (g.uni + FLT_EPSILON) - (g.enc - g.uni) / (g.enc + FLT_EPSILON);
  return g;
}

__device__ __forceinline__ float l1_mean6(const float* a, const float* b) {
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 6; ++k) s += fabsf(a[k] - b[k]);
  return s / 6.f;
}

// torch's sub-gradient conventions: min / max split the gradient on ties, clamp(min = 0) passes it where the argument is >= 0
__device__ __forceinline__ float pick_lt(float x, float y) { return x < y ? 1.f : (x == y ? 0.5f : 0.f); }

// gradient of giou(cc(a_cs), b) with respect to the six centre-size entries of a
__device__ __forceinline__ void giou_grad(const Box& a, const Box& b, const Giou& g, float* d_cs) {
  const float ue = g.uni + FLT_EPSILON, ee = g.enc + FLT_EPSILON;
  const float c_inter = 1.f / ue + g.inter / (ue * ue) - 1.f / ee;
  const float c_a1 = -g.inter / (ue * ue) + 1.f / ee;
  const float c_enc = -ue / (ee * ee);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int i = (k + 1) % 3, j = (k + 2) % 3;
    const float pe = g.e1[i] * g.e1[j], pw = g.w[i] * g.w[j], pv = g.v[i] * g.v[j];
    const float wok = g.wr[k] >= 0.f ? 1.f : 0.f, vok = g.vr[k] >= 0.f ? 1.f : 0.f;
    const float dhi = c_a1 * pe + c_inter * pw * wok * pick_lt(a.hi[k], b.hi[k]) + c_enc * pv * vok * pick_lt(b.hi[k], a.hi[k]);
    const float dlo = -c_a1 * pe - c_inter * pw * wok * pick_lt(b.lo[k], a.lo[k]) - c_enc * pv * vok * pick_lt(a.lo[k], b.lo[k]);
    d_cs[k] = dlo + dhi;
    d_cs[3 + k] = (dhi - dlo) / 2.f;
  }
}

__device__ __forceinline__ float block_sum(float v, float* red) {      // 256 threads -> every thread gets the total (fixed order)
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// ---------------------------------------------------------------------------------------------------- Hungarian cost matrices
// desc[problem] = {reg (float* -> [nq, 6] rows, row pitch 6), logit (float* -> [nq]), label (float* -> [n_pos, 6]), n_pos, n_col, nq}
// cost[problem][q][j], [rows x width] per problem, zero outside [nq x n_col]
__global__ __launch_bounds__(256) void match_cost_k(const int64_t* __restrict__ desc, float* __restrict__ cost, int rows, int width,
                                                    float l1_w, float giou_w, float disc_w, int match_ce, float gamma, float alpha) {
  const int64_t* d = desc + (int64_t)blockIdx.x * 6;
  const float* reg = reinterpret_cast<const float*>(d[0]);
  const float* logit = reinterpret_cast<const float*>(d[1]);
  const float* label = reinterpret_cast<const float*>(d[2]);
  const int n_pos = (int)d[3], n_col = (int)d[4], nq = (int)d[5];
  float* out = cost + (int64_t)blockIdx.x * rows * width;
  for (int e = threadIdx.x; e < rows * width; e += 256) {
    const int q = e / width, j = e - q * width;
    float c = 0.f;
    if (q < nq && j < n_col) {
      const float x = logit[q];
      float cp, cn;
      if (match_ce) {
        const float ex = __expf(-fabsf(x));
        const float p = x >= 0.f ? 1.f / (1.f + ex) : ex / (1.f + ex);
        cp = disc_w * (1.f - p); cn = disc_w * p;
      } else {
        cp = disc_w * focal_value(x, 1.f, gamma, alpha); cn = disc_w * focal_value(x, 0.f, gamma, alpha);
      }
      if (j < n_pos) {
        const float* a = reg + (int64_t)q * 6;
        const float* b = label + (int64_t)j * 6;
        const Giou g = giou_pair(cs_to_cc(a), cs_to_cc(b));
        c = (l1_w * l1_mean6(a, b) + giou_w * (1.f - g.value)) + cp;
      } else {
        c = cn;
      }
    }
    out[e] = c;
  }
}

// ---------------------------------------------------------------------------------------------------- one sample's losses
// logit [n] (n = targets x queries, row-major), reg -> [targets][1 + nq][6] (the first row of a target is its semantic box: skipped),
// label [n_boxes, 6], match [n] int64: index of the matched label box, or < 0.
// out[0] focal mean over all n entries (label = matched, alpha as given)      out[1] focal mean of the matched entries against 1 (no alpha)
// out[2] focal mean of the unmatched entries against 0 (no alpha)             out[3] l1 mean over matched pairs x 6
// out[4] 1 - mean giou over matched pairs            (out[1], out[3], out[4] = 0 without matched entries; out[2] = 0 without unmatched)
__global__ __launch_bounds__(256) void instance_loss_fwd_k(const float* __restrict__ logit, const float* __restrict__ reg,
                                                           const float* __restrict__ label, const int64_t* __restrict__ match, int n,
                                                           int nq, float gamma, float alpha, float* __restrict__ out) {
  __shared__ float red[4];
  float s_all = 0.f, s_pos = 0.f, s_neg = 0.f, s_l1 = 0.f, s_g = 0.f, c_pos = 0.f;
  for (int e = threadIdx.x; e < n; e += 256) {
    const int64_t m = match[e];
    const float x = logit[e];
    const bool pos = m >= 0;
    s_all += focal_value(x, pos ? 1.f : 0.f, gamma, alpha);
    if (pos) {
      s_pos += focal_value(x, 1.f, gamma, -1.f);
      const int t = e / nq, q = e - t * nq;
      const float* a = reg + ((int64_t)t * (nq + 1) + 1 + q) * 6;
      const float* b = label + m * 6;
      s_l1 += l1_mean6(a, b);
      s_g += giou_pair(cs_to_cc(a), cs_to_cc(b)).value;
      c_pos += 1.f;
    } else {
      s_neg += focal_value(x, 0.f, gamma, -1.f);
    }
  }
  s_all = block_sum(s_all, red); s_pos = block_sum(s_pos, red); s_neg = block_sum(s_neg, red);
  s_l1 = block_sum(s_l1, red); s_g = block_sum(s_g, red); c_pos = block_sum(c_pos, red);
  if (threadIdx.x == 0) {
    const float c_neg = (float)n - c_pos;
    out[0] = s_all / (float)n;
    out[1] = c_pos > 0.f ? s_pos / c_pos : 0.f;
    out[2] = c_neg > 0.f ? s_neg / c_neg : 0.f;
    out[3] = c_pos > 0.f ? s_l1 / c_pos : 0.f;
    out[4] = c_pos > 0.f ? 1.f - s_g / c_pos : 0.f;
    out[5] = c_pos;
  }
}

// d_logit[e] = g[0] * d focal / n;  d_reg rows of matched queries = g[3] * sign(a - b) / (6 n_pos) - g[4] * d giou / n_pos; zero elsewhere
// (g[1], g[2] belong to the logged means: no gradient). n_pos from the forward's out[5].
__global__ __launch_bounds__(256) void instance_loss_bwd_k(const float* __restrict__ logit, const float* __restrict__ reg,
                                                           const float* __restrict__ label, const int64_t* __restrict__ match, int n,
                                                           int nq, float gamma, float alpha, const float* __restrict__ fwd_out,
                                                           const float* __restrict__ g, float* __restrict__ d_logit,
                                                           float* __restrict__ d_reg) {
  const float g_focal = g[0] / (float)n;
  const float n_pos = fwd_out[5];
  const float g_l1 = n_pos > 0.f ? g[3] / (6.f * n_pos) : 0.f, g_giou = n_pos > 0.f ? -g[4] / n_pos : 0.f;
  const int nt = n / nq;
  // semantic-box rows (the first of each target) get no gradient
  for (int t = threadIdx.x; t < nt * 6; t += 256) d_reg[(int64_t)(t / 6) * (nq + 1) * 6 + t % 6] = 0.f;
  for (int e = threadIdx.x; e < n; e += 256) {
    const int64_t m = match[e];
    const bool pos = m >= 0;
    d_logit[e] = g_focal * focal_grad(logit[e], pos ? 1.f : 0.f, gamma, alpha);
    const int t = e / nq, q = e - t * nq;
    const int64_t row = ((int64_t)t * (nq + 1) + 1 + q) * 6;
    float dr[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (pos) {
      const float* a = reg + row;
      const float* b = label + m * 6;
      const Box ba = cs_to_cc(a), bb = cs_to_cc(b);
      const Giou gi = giou_pair(ba, bb);
      float dg[6];
      giou_grad(ba, bb, gi, dg);
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const float diff = a[k] - b[k];
        dr[k] = g_l1 * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f)) + g_giou * dg[k];
      }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) d_reg[row + k] = dr[k];
  }
}

}  // namespace

extern "C" {

int vm_box_match_cost(const int64_t* desc_dev, int n_problems, float* cost, int rows, int width, float l1_weight, float giou_weight,
                      float disc_weight, int match_ce, float gamma, float alpha, void* stream) {
  if (n_problems <= 0 || rows <= 0 || width <= 0) return VM_OK;
  if (!desc_dev || !cost) return VM_ERR_BAD_ARG;
  hipLaunchKernelGGL(match_cost_k, dim3(n_problems), dim3(256), 0, (hipStream_t)stream, desc_dev, cost, rows, width, l1_weight,
                     giou_weight, disc_weight, match_ce, gamma, alpha);
  VM_LAUNCH_CHECK();
  return VM_OK;
}

int vm_instance_loss_fwd(const float* logit, const float* reg, const float* label, const int64_t* match, int n_targets, int n_queries,
                         float gamma, float alpha, float* out6, void* stream) {
  if (n_targets <= 0 || n_queries <= 0) return VM_OK;
  if (!logit || !reg || !match || !out6) return VM_ERR_BAD_ARG;      // (label may be NULL when the sample has no boxes: nothing is matched)
  hipLaunchKernelGGL(instance_loss_fwd_k, dim3(1), dim3(256), 0, (hipStream_t)stream, logit, reg, label, match, n_targets * n_queries,
                     n_queries, gamma, alpha, out6);
  VM_LAUNCH_CHECK();
  return VM_OK;
}

int vm_instance_loss_bwd(const float* logit, const float* reg, const float* label, const int64_t* match, int n_targets, int n_queries,
                         float gamma, float alpha, const float* out6, const float* grad_out, float* d_logit, float* d_reg, void* stream) {
  if (n_targets <= 0 || n_queries <= 0) return VM_OK;
  if (!logit || !reg || !match || !out6 || !grad_out || !d_logit || !d_reg) return VM_ERR_BAD_ARG;
  hipLaunchKernelGGL(instance_loss_bwd_k, dim3(1), dim3(256), 0, (hipStream_t)stream, logit, reg, label, match, n_targets * n_queries,
                     n_queries, gamma, alpha, out6, grad_out, d_logit, d_reg);
  VM_LAUNCH_CHECK();
  return VM_OK;
}

}  // extern "C"
