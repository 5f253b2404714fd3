// Stand-alone A/B bench + bitwise check of the FOUR-WAVE form of the bf16 GEMM (gemm256w_k, round 6) against the eight-wave form (gemm256_k) on the
// C ABI of the shipped library: vm_gemm_w4_mode_(0 | 1) switches the form, everything else (scheduler, tile rows, epilogue) is shared. Both forms
// add the K-tiles of an output element in the same order, so the results must be BIT-IDENTICAL. Random operands rotated over three copies,
// variants interleaved in one process (guide rules 24 / 25).
//   tools/ubench/build_gemm_bench.sh w4;  tools/ubench/gemm_w4_bench [rounds] [only_shape] [force_tile: 0 | 192 | 256]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/vividmed_hip.h"

extern "C" int vm_gemm_sched_mode_(int mode);
extern "C" int vm_gemm_w4_mode_(int mode);
extern "C" int vm_gemm_force_tile_(int tile);
// -DGB_DEBUG: built together with csrc/gemm.hip + csrc/gemm256.hip under -DVM_GEMM_DEBUG_BUILD (tools/ubench/build_gemm_bench.sh): extra timing
// variants that knock out parts of the kernels (results wrong by construction; they are never checked)
#ifdef GB_DEBUG
extern "C" int vm_gemm_debug_set_(int bits);
#else
static int vm_gemm_debug_set_(int) { return 0; }
#endif

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define VK(x) do { int r_ = (x); if (r_ != 0) { printf("vm error %d at %s:%d\n", r_, __FILE__, __LINE__); exit(1); } } while (0)

struct Shape { const char* name; int M, N, K, K2; int seg_split; /* <0: none; else rows of segment 0 (device counts) */ int bias, residual, f32out; };

static unsigned short f2bf_host(float f) { unsigned u; std::memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static float bf2f_host(unsigned short b) { unsigned u = (unsigned)b << 16; float f; std::memcpy(&f, &u, 4); return f; }

__global__ void fill_k(unsigned short* p, size_t n, unsigned seed, float scale) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned x = (unsigned)i * 2654435761u ^ seed;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    const float u = ((x >> 8) * (1.0f / 16777216.0f)) * 2.f - 1.f;        // uniform [-1, 1)
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

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 10;
  const int only = argc > 2 ? atoi(argv[2]) : -1;
  const int stress = 0;
  VK(vm_gemm_force_tile_(argc > 3 ? atoi(argv[3]) : 0));
  std::vector<Shape> shapes = {
    // headline workload (phase-vg-448, batch 8): ViT-E M = 6280, decoder M = 3648 (two experts from device counts)
    {"vit qkv      6280x5376  K1792+64", 6280, 5376, 1792, 64, -1, 1, 0, 0},
    {"vit dense    6280x1792  K1792+64", 6280, 1792, 1792, 64, -1, 1, 1, 0},
    {"vit fc1      6280x15360 K1792+64", 6280, 15360, 1792, 64, -1, 1, 0, 0},
    {"vit fc2      6280x1792  K15360+64", 6280, 1792, 15360, 64, -1, 1, 1, 0},
    {"vit qkv dg   6280x1792  K5376+64", 6280, 1792, 5376, 64, -1, 0, 0, 0},
    {"dec gate/up  3648x11008 K4096+64 seg", 3648, 11008, 4096, 64, 2048, 0, 0, 0},
    {"dec qkv      3648x12288 K4096+64 seg", 3648, 12288, 4096, 64, 2048, 1, 0, 0},
    {"dec dense    3648x4096  K4096+64 seg", 3648, 4096, 4096, 64, 2048, 0, 1, 0},
    {"dec down     3648x4096  K11008+64 seg", 3648, 4096, 11008, 64, 2048, 0, 1, 0},
    // configs[3] / [4]: decoder rows 4128 (phase-grg-3d) and 4176 (model-hr-2d): 288 tiles of 256 rows
    {"grg dense    4128x4096  K4096+64 seg", 4128, 4096, 4096, 64, 2064, 0, 1, 0},
    {"grg down     4128x4096  K11008+64 seg", 4128, 4096, 11008, 64, 2064, 0, 1, 0},
    {"hr  dense    4176x4096  K4096+64 seg", 4176, 4096, 4096, 64, 3000, 0, 1, 0},
    {"grg gate/up  4128x11008 K4096+64 seg", 4128, 11008, 4096, 64, 2064, 0, 0, 0},
    // odd ones: ragged N, fp32 output, no extension, tiny
    {"ragged       1000x1000  K512", 1000, 1000, 512, 0, -1, 1, 1, 0},
    {"f32 out      2049x3000  K1024+64", 2049, 3000, 1024, 64, 700, 1, 1, 1},
    {"one tile row 200x8192   K2048", 200, 8192, 2048, 0, -1, 0, 0, 0},
    // plain products (the yardstick shapes of tools/bench_gemm_library.py)
    {"plain fc1    6280x15360 K1792", 6280, 15360, 1792, 0, -1, 0, 0, 0},
    {"plain dec qkv 3648x12288 K4096", 3648, 12288, 4096, 0, -1, 0, 0, 0},
    {"plain sq     8192x8192  K8192", 8192, 8192, 8192, 0, -1, 0, 0, 0},
  };
  int64_t ws_bytes = 0;
  VK(vm_gemm_workspace_bytes(&ws_bytes));
  void* ws;
  CK(hipMalloc(&ws, ws_bytes));
  CK(hipMemset(ws, 0, ws_bytes));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int fails = 0;
  for (size_t si = 0; si < shapes.size(); ++si) {
    if (only >= 0 && (int)si != only) continue;
    const Shape& s = shapes[si];
    const int NCOPY = 3;
    const int64_t ldc = (s.N + 7) / 8 * 8;
    unsigned short *A[NCOPY], *B0[NCOPY], *B1[NCOPY], *A2[NCOPY], *B2[NCOPY], *B21[NCOPY], *R[NCOPY];
    for (int c = 0; c < NCOPY; ++c) {
      A[c] = dev_bf16((size_t)s.M * s.K, 11 + c, 1.0f);
      B0[c] = dev_bf16((size_t)s.N * s.K, 23 + c, 1.0f / sqrtf((float)s.K));
      B1[c] = s.seg_split >= 0 ? dev_bf16((size_t)s.N * s.K, 37 + c, 1.0f / sqrtf((float)s.K)) : nullptr;
      A2[c] = s.K2 ? dev_bf16((size_t)s.M * s.K2, 41 + c, 1.0f) : nullptr;
      B2[c] = s.K2 ? dev_bf16((size_t)s.N * s.K2, 53 + c, 0.05f) : nullptr;
      B21[c] = (s.K2 && s.seg_split >= 0) ? dev_bf16((size_t)s.N * s.K2, 59 + c, 0.05f) : nullptr;
      R[c] = s.residual ? dev_bf16((size_t)s.M * ldc * (s.f32out ? 2 : 1), 61 + c, 1.0f) : nullptr;
    }
    unsigned short* bias = s.bias ? dev_bf16((size_t)s.N * (s.f32out ? 2 : 1), 71, 0.5f) : nullptr;
    const size_t cbytes = (size_t)s.M * ldc * (s.f32out ? 4 : 2);
    void *Cdp, *Csk, *Cst;
    CK(hipMalloc(&Cdp, cbytes)); CK(hipMalloc(&Csk, cbytes)); CK(hipMalloc(&Cst, cbytes));
    int* counts = nullptr;
    if (s.seg_split >= 0) {
      CK(hipMalloc(&counts, 8));
      const int h[2] = {s.seg_split, s.M};
      CK(hipMemcpy(counts, h, 8, hipMemcpyHostToDevice));
    }
    if (s.f32out && s.residual) { for (int c = 0; c < NCOPY; ++c) CK(hipMemset(R[c], 0, (size_t)s.M * ldc * 4)); }   // (bf16 random bits are not sane floats)
    if (s.f32out && bias) CK(hipMemset(bias, 0, (size_t)s.N * 4));
    CK(hipDeviceSynchronize());
    auto args_for = [&](int c, void* Cout, bool with_ws) {
      vm_gemm_args g;
      std::memset(&g, 0, sizeof g);
      g.A = A[c]; g.lda = s.K;
      g.B = B0[c]; g.B_1 = B1[c]; g.ldb = s.K;
      g.A2 = A2[c]; g.lda2 = s.K2; g.B2 = B2[c]; g.B2_1 = B21[c]; g.ldb2 = s.K2; g.K2 = s.K2; g.alpha2 = 1.0f;
      g.bias = bias; g.bias_1 = bias;
      g.residual = R[c]; g.ldr = ldc;
      g.C = Cout; g.ldc = ldc;
      g.M = s.M; g.N = s.N; g.K = s.K;
      g.counts_dev = counts; g.split = -1;
      g.act = 0; g.out_dtype = s.f32out ? VM_F32 : VM_BF16;
      g.alpha = 1.0f;
      g.workspace = with_ws ? ws : nullptr; g.workspace_bytes = with_ws ? ws_bytes : 0;
      return g;
    };
    // ---- check: W4 vs W8 on copy 0, bit for bit
    CK(hipMemset(Cdp, 0xFF, cbytes)); CK(hipMemset(Csk, 0xFF, cbytes));
    VK(vm_gemm_w4_mode_(0));
    { vm_gemm_args g = args_for(0, Cdp, false); VK(vm_gemm_bf16(&g, st)); }
    VK(vm_gemm_w4_mode_(1));
    { vm_gemm_args g = args_for(0, Csk, false); VK(vm_gemm_bf16(&g, st)); }
    CK(hipStreamSynchronize(st));
    std::vector<unsigned char> hd(cbytes), hs(cbytes);
    CK(hipMemcpy(hd.data(), Cdp, cbytes, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hs.data(), Csk, cbytes, hipMemcpyDeviceToHost));
    size_t ndiff = 0;
    const size_t esz = s.f32out ? 4 : 2;
    size_t rowhist[16] = {0}, colhist[16] = {0};
    for (int m = 0; m < s.M; ++m)
      for (int n = 0; n < s.N; ++n) {
        const size_t i = ((size_t)m * ldc + n) * esz;
        if (std::memcmp(&hd[i], &hs[i], esz) != 0) {
          if (ndiff < 6) {
            if (s.f32out) { float a, b; std::memcpy(&a, &hd[i], 4); std::memcpy(&b, &hs[i], 4); printf("     first differences: m %d n %d: W8 %g W4 %g\n", m, n, a, b); }
            else { unsigned short a, b; std::memcpy(&a, &hd[i], 2); std::memcpy(&b, &hs[i], 2); printf("     first differences: m %d n %d: W8 %g (%04x) W4 %g (%04x)\n", m, n, bf2f_host(a), a, bf2f_host(b), b); }
          }
          ++ndiff;
          if (m < 4096 && n < 4096) { rowhist[m / 16 % 16]++; colhist[n / 16 % 16]++; }
        }
      }
    const bool ok = ndiff == 0;
    if (!ok) { printf("     by (m / 16) %% 16:"); for (int k = 0; k < 16; ++k) printf(" %zu", rowhist[k]); printf("\n     by (n / 16) %% 16:"); for (int k = 0; k < 16; ++k) printf(" %zu", colhist[k]); printf("\n"); }
    if (!ok) ++fails;
    printf("[%2zu] %-40s check %s: %zu of %zu elements differ\n", si, s.name, ok ? "ok (bit-identical)" : "FAIL", ndiff, (size_t)s.M * s.N);
    (void)stress;
    // ---- timing: DP / SK / AUTO interleaved, operands rotated
    constexpr int NV = 2;
    const char* vn[NV] = {"W8", "W4"};
    std::vector<float> t[NV];
    for (int r = 0; r < rounds + 2; ++r)
      for (int v = 0; v < NV; ++v) {
        VK(vm_gemm_w4_mode_(v));
        const int reps = 4;
        CK(hipEventRecord(e0, st));
        for (int k = 0; k < reps; ++k) { vm_gemm_args g = args_for((r * reps + k) % NCOPY, v == 0 ? Cdp : Csk, false); VK(vm_gemm_bf16(&g, st)); }
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) t[v].push_back(ms * 1e3f / reps);
      }
    const double flop = 2.0 * s.M * s.N * (double)(s.K + s.K2);
    printf("     ");
    for (int v = 0; v < NV; ++v) {
      std::sort(t[v].begin(), t[v].end());
      const float med = t[v][t[v].size() / 2], mn = t[v][0];
      printf("%s %7.1f us (min %7.1f) %5.0f TF   ", vn[v], med, mn, flop / med / 1e6);
    }
    printf("\n");
    fflush(stdout);
    for (int c = 0; c < NCOPY; ++c) { hipFree(A[c]); hipFree(B0[c]); if (B1[c]) hipFree(B1[c]); if (A2[c]) hipFree(A2[c]); if (B2[c]) hipFree(B2[c]); if (B21[c]) hipFree(B21[c]); if (R[c]) hipFree(R[c]); }
    if (bias) hipFree(bias);
    hipFree(Cdp); hipFree(Csk); hipFree(Cst);
    if (counts) hipFree(counts);
  }
  printf(fails ? "FAILED: %d\n" : "all checks passed\n", fails);
  return fails ? 1 : 0;
}
