// bare MFMA issue-rate micro-benchmark: what does this MI355X sustain per MFMA shape / waves per SIMD / data?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int NACC>
__global__ __launch_bounds__(512, 2) void k16(const unsigned* in, float* out, int iters) {
  bf16x8_t a, b;
  unsigned v[4];
  for (int i = 0; i < 4; ++i) v[i] = in[(threadIdx.x * 4 + i) & 1023];
  a = __builtin_bit_cast(bf16x8_t, *(uint4*)v);
  for (int i = 0; i < 4; ++i) v[i] = in[(threadIdx.x * 4 + i + 512) & 1023];
  b = __builtin_bit_cast(bf16x8_t, *(uint4*)v);
  f32x4_t acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4_t){0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(512, 2) void k32(const unsigned* in, float* out, int iters) {
  bf16x8_t a, b;
  unsigned v[4];
  for (int i = 0; i < 4; ++i) v[i] = in[(threadIdx.x * 4 + i) & 1023];
  a = __builtin_bit_cast(bf16x8_t, *(uint4*)v);
  for (int i = 0; i < 4; ++i) v[i] = in[(threadIdx.x * 4 + i + 512) & 1023];
  b = __builtin_bit_cast(bf16x8_t, *(uint4*)v);
  f32x16_t acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
double run(F launch, double flops_per_launch) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) launch();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 10; ++i) launch();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return flops_per_launch * 10 / (ms * 1e-3) / 1e12;
}

int main() {
  unsigned* in; float* out;
  hipMalloc(&in, 4096); hipMalloc(&out, 1 << 24);
  for (int data = 0; data < 2; ++data) {
    std::vector<unsigned> h(1024);
    for (auto& x : h) x = data ? (((rand() & 0x7fff) | 0x3c00) << 16 | ((rand() & 0x7fff) | 0x3c00)) ^ ((rand() & 1) << 31) : 0;
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    const int iters = 20000;
    for (int threads : {256, 512}) {
      const int blocks = 256;
      double f16 = (double)blocks * (threads / 64) * iters * 2.0 * 16 * 16 * 32;
      printf("data=%s threads/block=%d (waves/SIMD=%d):\n", data ? "random" : "zero", threads, threads / 256);
      printf("  16x16x32 x8acc  : %7.0f TF\n", run([&] { hipLaunchKernelGGL(k16<8>, dim3(blocks), dim3(threads), 0, 0, in, out, iters); }, f16 * 8));
      printf("  16x16x32 x32acc : %7.0f TF\n", run([&] { hipLaunchKernelGGL(k16<32>, dim3(blocks), dim3(threads), 0, 0, in, out, iters); }, f16 * 32));
      double f32 = (double)blocks * (threads / 64) * iters * 2.0 * 32 * 32 * 16;
      printf("  32x32x16 x4acc  : %7.0f TF\n", run([&] { hipLaunchKernelGGL(k32<4>, dim3(blocks), dim3(threads), 0, 0, in, out, iters); }, f32 * 4));
      printf("  32x32x16 x8acc  : %7.0f TF\n", run([&] { hipLaunchKernelGGL(k32<8>, dim3(blocks), dim3(threads), 0, 0, in, out, iters); }, f32 * 8));
    }
  }
  return 0;
}
