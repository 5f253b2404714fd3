// Sustained bf16 MFMA rate of THIS device, measured in-process: the denominator bench.py reports next to the nominal 2.5 PFLOP/s
// ("re-measure with an MFMA micro-bench on the box and use the measured peak alongside"). Not on the training path.
// Bare v_mfma_f32_16x16x32_bf16 (the dominant GEMM's instruction), operands in registers, two waves per SIMD, 8 independent
// accumulators per wave, uniform random operands in [-1, 1) — the chip lowers its clock under matrix load and holds a LOWER one on
// random data than on zeros (guide MI355X_MICROARCH.md "DVFS give-back": 1 247 vs 1 483 TFLOP/s for one binary), so zeros would
// overstate what any real GEMM can reach.
#include "vm_common.hpp"

namespace {

__global__ __launch_bounds__(512, 2) void mfma_rate_k(float* sink, int iters, unsigned seed) {
  // per-lane random operands: hash -> float in [-1, 1) -> bf16
  bf16x8_t a, b;
  u16x8_t ua, ub;
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const unsigned ha = vm_mix32(t * 16 + i, seed, 0x9E3779B9u), hb = vm_mix32(t * 16 + 8 + i, seed ^ 0x85EBCA6Bu, 0xC2B2AE35u);
    ua[i] = f2bf((float)(int)ha * (1.0f / 2147483648.0f));
    ub[i] = f2bf((float)(int)hb * (1.0f / 2147483648.0f));
  }
  a = __builtin_bit_cast(bf16x8_t, ua);
  b = __builtin_bit_cast(bf16x8_t, ub);
  f32x4_t acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) sink[t] = s;          // keeps the chain alive; practically never stores
}

}  // namespace

extern "C" int vm_ubench_mfma_bf16(float seconds, float* tflops_host, void* stream) {
  if (!tflops_host || !(seconds > 0.f)) return VM_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess) return VM_ERR_LAUNCH;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
  float* sink = nullptr;
  if (hipMalloc(&sink, (size_t)cus * 512 * sizeof(float)) != hipSuccess) return VM_ERR_LAUNCH;
  const int iters = 4000;                                            // ~4-5 ms per launch
  const double flops_per_launch = (double)cus * 8 /*waves*/ * iters * 8 /*accumulators*/ * 2.0 * 16 * 16 * 32;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { (void)hipFree(sink); return VM_ERR_LAUNCH; }
  // first half: load only (the clock settles); second half: timed
  double done_s = 0.0, timed_s = 0.0, timed_flops = 0.0;
  int rc = VM_OK;
  while (done_s < seconds) {
    const bool timed = done_s >= 0.5 * seconds;
    const int reps = 20;
    (void)hipEventRecord(e0, st);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(mfma_rate_k, dim3(cus), dim3(512), 0, st, sink, iters, 1234u + r);
    (void)hipEventRecord(e1, st);
    if (hipEventSynchronize(e1) != hipSuccess) { rc = VM_ERR_LAUNCH; break; }
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { rc = VM_ERR_LAUNCH; break; }
    done_s += ms * 1e-3;
    if (timed) { timed_s += ms * 1e-3; timed_flops += flops_per_launch * reps; }
    if (ms <= 0.f) { rc = VM_ERR_LAUNCH; break; }
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(sink);
  if (rc != VM_OK) return rc;
  *tflops_host = timed_s > 0 ? (float)(timed_flops / timed_s / 1e12) : 0.f;
  return VM_OK;
}
