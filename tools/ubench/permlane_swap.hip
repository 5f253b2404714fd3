// v_permlane16_swap / v_permlane32_swap: raw semantics and the row-reduction built from them against shfl_xor
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float row4_sum(float x) {
  // (inline assembly: with both operands holding the same value the builtin's two results come back as ONE register — hipcc 7.2 adds
  // r[0] to itself; the s_nop covers the VALU-write -> permlane-read wait states the assembler does not insert for inline code)
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  a = a + b; b = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__global__ void k(unsigned* out, float* f) {
  unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[threadIdx.x] = r[0]; out[64 + threadIdx.x] = r[1];
  auto q = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[128 + threadIdx.x] = q[0]; out[192 + threadIdx.x] = q[1];
  float x = (float)(threadIdx.x * threadIdx.x % 37);
  float ref = x + __shfl_xor(x, 16, 64); ref += __shfl_xor(ref, 32, 64);
  f[threadIdx.x] = row4_sum(x); f[64 + threadIdx.x] = ref;
}
int main() {
  unsigned* d; float* f; (void)hipMalloc(&d, 256 * 4); (void)hipMalloc(&f, 128 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, f);
  unsigned h[256]; float hf[128]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); (void)hipMemcpy(hf, f, sizeof(hf), hipMemcpyDeviceToHost);
  const char* names[4] = {"swap16 r[0]", "swap16 r[1]", "swap32 r[0]", "swap32 r[1]"};
  for (int v = 0; v < 4; ++v) { printf("%s:", names[v]); for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[v * 64 + i]); printf("\n"); }
  int bad = 0; for (int i = 0; i < 64; ++i) bad += hf[i] != hf[64 + i];
  printf("row4_sum vs shfl_xor: %d mismatches; lane 5: %g vs %g, lane 21: %g vs %g\n", bad, hf[5], hf[69], hf[21], hf[85]);
  return 0;
}
