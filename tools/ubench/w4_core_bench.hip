// Stand-alone bench of the FOUR-WAVE GEMM body (csrc/gemm256w.hpp, INCLUDED here: what is timed is the code that ships) on plain products,
// against the shipped library's eight-wave kernel (vm_gemm_bf16 with vm_gemm_w4_mode_(0)) as reference and yardstick. One binary per build
// variant (-DVM_W4_EXPERIMENT=<bits>: knock-outs of csrc/gemm256w.hpp, results wrong by construction and not checked then), seconds to compile:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DVM_KEEP_DENORMS -Immmm_amd/csrc -Iinclude [-DVM_W4_EXPERIMENT=n] tools/ubench/w4_core_bench.hip \
//         -Lmmmm_amd/lib -lvividmed_hip -Wl,-rpath,'$ORIGIN/../../mmmm_amd/lib' -o tools/ubench/w4_core_bench
//   tools/ubench/w4_core_bench [rounds]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../include/vividmed_hip.h"
#include "gemm256w.hpp"

extern "C" int vm_gemm_w4_mode_(int mode);
extern "C" int vm_gemm_force_tile_(int tile);
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define VK(x) do { int r_ = (x); if (r_ != 0) { printf("vm error %d at %s:%d\n", r_, __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_k(unsigned short* p, size_t n, unsigned seed, float scale) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned x = (unsigned)i * 2654435761u ^ seed;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    const float u = ((x >> 8) * (1.0f / 16777216.0f)) * 2.f - 1.f;
    __bf16 b = (__bf16)(u * scale);
    p[i] = __builtin_bit_cast(unsigned short, b);
  }
}
static unsigned short* dev_bf16(size_t n, unsigned seed, float scale) {
  unsigned short* p;
  CK(hipMalloc(&p, n * 2));
  hipLaunchKernelGGL(fill_k, dim3(2048), dim3(256), 0, 0, p, n, seed, scale);
  return p;
}
struct Shape { const char* name; int M, N, K; };

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 8;
  const int ta = argc > 2 ? atoi(argv[2]) : 8;          // 8: 256-row tiles, 6: 192-row tiles
  const int k2 = argc > 3 ? atoi(argv[3]) : 0;          // LoRA extension columns (0 | 64)
  const int zero = argc > 4 ? atoi(argv[4]) : 0;        // debug: 1 = A2 all zero (the extension contributes nothing), 2 = A all zero (only the extension contributes)
  std::vector<Shape> shapes = {{"vit fc1   6280x15360 K1792", 6280, 15360, 1792}, {"vit qkv   6280x5376  K1792", 6280, 5376, 1792},
                               {"vit fc2   6280x1792  K15360", 6280, 1792, 15360}, {"dec qkv   3648x12288 K4096", 3648, 12288, 4096},
                               {"dec down  3648x4096  K11008", 3648, 4096, 11008}, {"sq        8192x8192  K8192", 8192, 8192, 8192},
                               {"odd tiles 6280x5376  K1856", 6280, 5376, 1856}, {"        6280x5376  K1728", 6280, 5376, 1728}, {"small 512x512 K256", 512, 512, 256}, {"small 512x512 K192", 512, 512, 192}};
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute((const void*)gemm256w_k<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
  CK(hipFuncSetAttribute((const void*)gemm256w_k<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
  int cus = 256;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  VK(vm_gemm_w4_mode_(0));
  VK(vm_gemm_force_tile_(ta == 6 ? 192 : 256));
  printf("VM_W4_EXPERIMENT = %d, tile rows %d, K2 %d\n", (int)VM_W4_EXPERIMENT, ta == 6 ? 192 : 256, k2);
  int fails = 0;
  for (const Shape& s : shapes) {
    const int NCOPY = 3;
    unsigned short *A[NCOPY], *B[NCOPY], *A2[NCOPY], *B2[NCOPY];
    for (int c = 0; c < NCOPY; ++c) {
      A[c] = dev_bf16((size_t)s.M * s.K, 11 + c, 1.0f); B[c] = dev_bf16((size_t)s.N * s.K, 23 + c, 1.0f / sqrtf((float)s.K));
      A2[c] = k2 ? dev_bf16((size_t)s.M * k2, 41 + c, 1.0f) : nullptr; B2[c] = k2 ? dev_bf16((size_t)s.N * k2, 53 + c, 0.05f) : nullptr;
      if (zero == 1 && A2[c]) CK(hipMemset(A2[c], 0, (size_t)s.M * k2 * 2));
      if (zero == 2) CK(hipMemset(A[c], 0, (size_t)s.M * s.K * 2));
    }
    const size_t cbytes = (size_t)s.M * s.N * 2;
    void *C8, *C4;
    CK(hipMalloc(&C8, cbytes)); CK(hipMalloc(&C4, cbytes));
    auto lib_args = [&](int c) {
      vm_gemm_args g; std::memset(&g, 0, sizeof g);
      g.A = A[c]; g.lda = s.K; g.B = B[c]; g.ldb = s.K; g.alpha2 = 1.f; g.C = C8;
      g.A2 = A2[c]; g.lda2 = k2; g.B2 = B2[c]; g.ldb2 = k2; g.K2 = k2; g.ldc = s.N; g.M = s.M; g.N = s.N; g.K = s.K; g.split = -1;
      g.out_dtype = VM_BF16; g.alpha = 1.f;
      return g;
    };
    auto w4_params = [&](int c) {
      GemmParams p; std::memset(&p, 0, sizeof p);
      p.A = (const char*)A[c]; p.lda = s.K; p.B0 = p.B1 = (const char*)B[c]; p.ldb = s.K; p.alpha2 = 1.f; p.C = C4; p.ldc = s.N;
      p.A2 = (const char*)A2[c]; p.lda2 = k2; p.B2_0 = p.B2_1 = (const char*)B2[c]; p.ldb2 = k2; p.K2 = k2;
      const int bmt = ta == 6 ? 192 : 256;
      p.M = s.M; p.N = s.N; p.K = s.K; p.split = -1; p.tiles_m = (s.M + bmt - 1) / bmt; p.tiles_n = (s.N + 255) / 256; p.ksplit = 1; p.kchunk = s.K; p.tail_base = 256;
      return p;
    };
    auto launch4 = [&](int c) { const GemmParams p = w4_params(c); const int tiles = p.tiles_m * p.tiles_n;
      if (ta == 6) hipLaunchKernelGGL((gemm256w_k<6>), dim3(tiles < cus ? tiles : cus), dim3(256), 163840, st, p);
      else hipLaunchKernelGGL((gemm256w_k<8>), dim3(tiles < cus ? tiles : cus), dim3(256), 163840, st, p); };
    CK(hipMemset(C8, 0xFF, cbytes)); CK(hipMemset(C4, 0xFF, cbytes));
    { vm_gemm_args g = lib_args(0); VK(vm_gemm_bf16(&g, st)); }
    launch4(0);
    CK(hipStreamSynchronize(st));
    size_t ndiff = 0;
    if (VM_W4_EXPERIMENT == 0 || (VM_W4_EXPERIMENT & ~W4X_SCHED_MASK) == 0) {
      std::vector<unsigned short> h8((size_t)s.M * s.N), h4((size_t)s.M * s.N);
      CK(hipMemcpy(h8.data(), C8, cbytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(h4.data(), C4, cbytes, hipMemcpyDeviceToHost));
      size_t colh[16] = {0}, rowh[16] = {0};
      for (size_t i = 0; i < h8.size(); ++i) if (h8[i] != h4[i]) { ++ndiff; colh[(i % s.N) / 16 % 16]++; rowh[(i / s.N) / 16 % 16]++; }
      if (ndiff) {
        ++fails;
        printf("  differing elements by (n / 16) %% 16:"); for (int k = 0; k < 16; ++k) printf(" %zu", colh[k]);
        printf("\n  by (m / 16) %% 16:"); for (int k = 0; k < 16; ++k) printf(" %zu", rowh[k]); printf("\n");
      }
    }
    std::vector<float> t[2];
    for (int r = 0; r < rounds + 2; ++r)
      for (int v = 0; v < 2; ++v) {
        const int reps = 4;
        CK(hipEventRecord(e0, st));
        for (int k = 0; k < reps; ++k) { const int c = (r * reps + k) % NCOPY; if (v == 0) { vm_gemm_args g = lib_args(c); VK(vm_gemm_bf16(&g, st)); } else launch4(c); }
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) t[v].push_back(ms * 1e3f / reps);
      }
    const double flop = 2.0 * s.M * s.N * (double)(s.K + k2);
    printf("%-28s %s  ", s.name, ndiff ? "DIFFERS" : "bit-identical");
    for (int v = 0; v < 2; ++v) { std::sort(t[v].begin(), t[v].end()); const float med = t[v][t[v].size() / 2]; printf("%s %7.1f us (min %7.1f) %5.0f TF   ", v ? "W4" : "W8", med, t[v][0], flop / med / 1e6); }
    printf("\n"); fflush(stdout);
    for (int c = 0; c < NCOPY; ++c) { hipFree(A[c]); hipFree(B[c]); if (A2[c]) hipFree(A2[c]); if (B2[c]) hipFree(B2[c]); }
    hipFree(C8); hipFree(C4);
  }
  printf(fails ? "FAILED: %d\n" : "all checks passed\n", fails);
  return fails ? 1 : 0;
}
