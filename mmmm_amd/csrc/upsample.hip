// Trilinear up-sampling of the mask logits to image resolution: F.interpolate(low, size, mode='trilinear',
// align_corners=False) in Sam._predict_masks, /root/reference/mmmm/models/segvol/modeling/sam.py:57-87.
// ATen's backward scatters with fp32 atomics (order-dependent, the remaining run-to-run noise of the heads) and costs
// 340 us per call; here the backward GATHERS: one thread per low-resolution voxel sums the <= (2s+1)^3 output gradients that
// reference it, in a fixed order (deterministic). Index rule = ATen's area_pixel_compute_source_index with align_corners
// false: src = max(scale * (dst + 0.5) - 0.5, 0), i0 = floor(src), i1 = min(i0 + 1, n - 1), weight of i1 = src - i0.
#include "vm_common.hpp"

namespace {

struct Axis { int n_in, n_out; float scale; };      // scale = n_in / n_out

__device__ __forceinline__ void src_index(const Axis& a, int o, int& i0, int& i1, float& w1) {
  float s = a.scale * (o + 0.5f) - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = min((int)s, a.n_in - 1);
  i1 = min(i0 + 1, a.n_in - 1);
  w1 = s - (float)i0;
}

__global__ __launch_bounds__(256) void upsample3d_fwd_k(const float* __restrict__ x, float* __restrict__ y, Axis az, Axis ay, Axis ax) {
  const int64_t plane_in = (int64_t)az.n_in * ay.n_in * ax.n_in, plane_out = (int64_t)az.n_out * ay.n_out * ax.n_out;
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= plane_out) return;
  const float* xn = x + blockIdx.y * plane_in;
  const int ox = (int)(o % ax.n_out), oy = (int)((o / ax.n_out) % ay.n_out), oz = (int)(o / ((int64_t)ax.n_out * ay.n_out));
  int z0, z1, y0, y1, x0, x1;
  float wz, wy, wx;
  src_index(az, oz, z0, z1, wz); src_index(ay, oy, y0, y1, wy); src_index(ax, ox, x0, x1, wx);
  auto at = [&](int z, int yy, int xx) { return xn[((int64_t)z * ay.n_in + yy) * ax.n_in + xx]; };
  // same association as ATen's kernel: ((x-lerp over y-lerp) over z-lerp) written out as weighted sums
  const float v = (1.f - wz) * ((1.f - wy) * ((1.f - wx) * at(z0, y0, x0) + wx * at(z0, y0, x1)) + wy * ((1.f - wx) * at(z0, y1, x0) + wx * at(z0, y1, x1)))
                + wz * ((1.f - wy) * ((1.f - wx) * at(z1, y0, x0) + wx * at(z1, y0, x1)) + wy * ((1.f - wx) * at(z1, y1, x0) + wx * at(z1, y1, x1)));
  y[blockIdx.y * plane_out + o] = v;
}

// per input index i of one axis: the range of output indices whose i0 or i1 can be i, and the weight each gives to i
__device__ __forceinline__ void out_range(const Axis& a, int i, int& lo, int& hi) {
  const float inv = 1.f / a.scale;
  lo = max(0, (int)floorf((i - 1 + 0.5f) * inv - 0.5f) - 1);
  hi = min(a.n_out - 1, (int)ceilf((i + 1 + 0.5f) * inv - 0.5f) + 1);
}
__device__ __forceinline__ float weight_to(const Axis& a, int o, int i) {
  int i0, i1; float w1;
  src_index(a, o, i0, i1, w1);
  float w = 0.f;
  if (i0 == i) w += 1.f - w1;
  if (i1 == i) w += w1;          // (i0 == i1 at the upper edge: both halves land on the same voxel)
  return w;
}

__global__ __launch_bounds__(256) void upsample3d_bwd_k(const float* __restrict__ gy, float* __restrict__ gx, Axis az, Axis ay, Axis ax) {
  const int64_t plane_in = (int64_t)az.n_in * ay.n_in * ax.n_in, plane_out = (int64_t)az.n_out * ay.n_out * ax.n_out;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= plane_in) return;
  const float* gn = gy + blockIdx.y * plane_out;
  const int ix = (int)(i % ax.n_in), iy = (int)((i / ax.n_in) % ay.n_in), iz = (int)(i / ((int64_t)ax.n_in * ay.n_in));
  int zl, zh, yl, yh, xl, xh;
  out_range(az, iz, zl, zh); out_range(ay, iy, yl, yh); out_range(ax, ix, xl, xh);
  float acc = 0.f;
  for (int oz = zl; oz <= zh; ++oz) {
    const float wz = weight_to(az, oz, iz);
    if (wz == 0.f) continue;
    for (int oy = yl; oy <= yh; ++oy) {
      const float wzy = wz * weight_to(ay, oy, iy);
      if (wzy == 0.f) continue;
      const float* row = gn + ((int64_t)oz * ay.n_out + oy) * ax.n_out;
      float r = 0.f;
      for (int ox = xl; ox <= xh; ++ox) r += weight_to(ax, ox, ix) * row[ox];
      acc += wzy * r;
    }
  }
  gx[blockIdx.y * plane_in + i] = acc;
}

}  // namespace

extern "C" {

static int upsample_check(int n, int di, int hi, int wi, int dout, int ho, int wo) {
  if (n < 0 || di <= 0 || hi <= 0 || wi <= 0 || dout <= 0 || ho <= 0 || wo <= 0) return VM_ERR_BAD_ARG;
  if (n > 65535) return VM_ERR_UNSUPPORTED;
  return VM_OK;
}

int vm_upsample_trilinear3d_fwd(const float* x, float* y, int n, int di, int hi, int wi, int dout, int ho, int wo, void* stream) {
  if (!x || !y) return VM_ERR_BAD_ARG;
  const int rc = upsample_check(n, di, hi, wi, dout, ho, wo);
  if (rc != VM_OK || n == 0) return rc;
  const Axis az{di, dout, (float)di / dout}, ay{hi, ho, (float)hi / ho}, ax{wi, wo, (float)wi / wo};
  const int64_t blocks = ((int64_t)dout * ho * wo + 255) / 256;
  if (blocks > 0x7FFFFFFF) return VM_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(upsample3d_fwd_k, dim3((unsigned)blocks, n), dim3(256), 0, (hipStream_t)stream, x, y, az, ay, ax);
  return hipGetLastError() == hipSuccess ? VM_OK : VM_ERR_LAUNCH;
}

int vm_upsample_trilinear3d_bwd(const float* gy, float* gx, int n, int di, int hi, int wi, int dout, int ho, int wo, void* stream) {
  if (!gy || !gx) return VM_ERR_BAD_ARG;
  const int rc = upsample_check(n, di, hi, wi, dout, ho, wo);
  if (rc != VM_OK || n == 0) return rc;
  const Axis az{di, dout, (float)di / dout}, ay{hi, ho, (float)hi / ho}, ax{wi, wo, (float)wi / wo};
  const int64_t blocks = ((int64_t)di * hi * wi + 255) / 256;
  hipLaunchKernelGGL(upsample3d_bwd_k, dim3((unsigned)blocks, n), dim3(256), 0, (hipStream_t)stream, gy, gx, az, ay, ax);
  return hipGetLastError() == hipSuccess ? VM_OK : VM_ERR_LAUNCH;
}

}  // extern "C"
