// Host-side AddressSanitizer run of the C ABI ("ASan host build of the extension"; CPU only — GPU ASan / xnack+ code objects
// are not available on this pool). Built by tools/asan_host.sh: the translation units are compiled with -fsanitize=address for the HOST
// (device code is not instrumented) and linked with this driver, which walks the host-only parts of the ABI — argument validation of every
// entry point that validates before it launches, the workspace-size helpers, the event-profiling bookkeeping — without a GPU: every call
// below must return before any kernel launch.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../include/vividmed_hip.h"

static int fails = 0;
#define EXPECT(call, want)                                                                  \
  do {                                                                                      \
    const int got__ = (call);                                                               \
    if (got__ != (want)) { std::printf("FAIL %s -> %d (want %d)\n", #call, got__, (want)); ++fails; } \
  } while (0)

int main() {
  std::vector<char> buf(1 << 16, 0);
  void* p = buf.data();                       // 16-byte aligned host memory: never dereferenced by the entry points on these paths
  float* f = reinterpret_cast<float*>(buf.data());
  EXPECT(vm_version() >= 100 ? 0 : 1, 0);

  vm_gemm_args g;
  std::memset(&g, 0, sizeof g);
  EXPECT(vm_gemm_bf16(nullptr, nullptr), VM_ERR_BAD_ARG);
  EXPECT(vm_gemm_bf16(&g, nullptr), VM_ERR_BAD_ARG);                     // null operands
  g.A = p; g.B = p; g.C = p; g.M = 0; g.N = 8;
  EXPECT(vm_gemm_bf16(&g, nullptr), VM_OK);                              // empty problem
  g.M = 8; g.N = 8; g.K = 60; g.lda = g.ldb = 64; g.ldc = 8; g.split = -1;
  EXPECT(vm_gemm_bf16(&g, nullptr), VM_ERR_BAD_ARG);                     // K % 64
  g.K = 64; g.lda = 60;
  EXPECT(vm_gemm_bf16(&g, nullptr), VM_ERR_BAD_ARG);                     // lda % 8
  g.lda = 64; g.out_dtype = 7;
  EXPECT(vm_gemm_bf16(&g, nullptr), VM_ERR_BAD_ARG);                     // unknown output dtype
  g.out_dtype = VM_BF16; g.K2 = 64;
  EXPECT(vm_gemm_bf16(&g, nullptr), VM_ERR_BAD_ARG);                     // extension without operands
  g.K2 = 0; g.out_dtype = VM_BF16;
  EXPECT(vm_gemm_f32(&g, nullptr), VM_ERR_UNSUPPORTED);                  // fp32 operands need fp32 output
  g.ksplit = 4;
  EXPECT(vm_gemm_bf16(&g, nullptr), VM_ERR_BAD_ARG);                     // split-K needs fp32 output
  g.ksplit = 0;
  EXPECT(vm_gemm_f32_mode(1), VM_ERR_BAD_ARG);
  EXPECT(vm_gemm_f32_mode(3), VM_OK);
  EXPECT(vm_gemm_fp8(&g, nullptr, nullptr, nullptr, nullptr), VM_ERR_BAD_ARG);   // no scales
  g.K = 64;
  EXPECT(vm_gemm_fp8(&g, f, f, nullptr, nullptr), VM_ERR_BAD_ARG);       // K % 128
  g.K = 128; g.counts_dev = reinterpret_cast<const int32_t*>(p);
  EXPECT(vm_gemm_fp8(&g, f, f, nullptr, nullptr), VM_ERR_BAD_ARG);       // two segments without the second weight / scales

  EXPECT(vm_quant_rows_fp8(p, 64, p, 64, f, f, 4, 60, VM_BF16, nullptr, nullptr), VM_ERR_BAD_ARG);   // cols % 16
  EXPECT(vm_quant_rows_fp8(p, 64, p, 64, f, f, 0, 64, VM_BF16, nullptr, nullptr), VM_OK);
  EXPECT(vm_scale_rows_bf16(p, 64, f, p, 64, 4, 60, nullptr), VM_ERR_BAD_ARG);
  EXPECT(vm_adamw(p, p, p, p, 12, 1e-3f, 0.9f, 0.999f, 1e-8f, 0.f, 1, nullptr, VM_BF16, nullptr), VM_ERR_BAD_ARG);   // n % 8
  EXPECT(vm_adamw(p, p, p, p, 16, 1e-3f, 0.9f, 0.999f, 1e-8f, 0.f, 0, nullptr, VM_BF16, nullptr), VM_ERR_BAD_ARG);   // step < 1
  EXPECT(vm_adamw(p, p, p, p, 0, 1e-3f, 0.9f, 0.999f, 1e-8f, 0.f, 1, nullptr, VM_BF16, nullptr), VM_OK);
  EXPECT(vm_transpose_batched(nullptr, 3, 1, VM_BF16, nullptr), VM_ERR_BAD_ARG);
  EXPECT(vm_transpose_batched(reinterpret_cast<const int64_t*>(p), 70000, 1, VM_BF16, nullptr), VM_ERR_BAD_ARG);
  EXPECT(vm_accum_f32_table(nullptr, 2, 4, nullptr), VM_ERR_BAD_ARG);
  EXPECT(vm_accum_f32_table(reinterpret_cast<const int64_t*>(p), 0, 4, nullptr), VM_OK);
  EXPECT(vm_transpose_colsum(p, 8, p, 8, 4, 4, VM_F32, nullptr, nullptr), VM_ERR_BAD_ARG);
  EXPECT(vm_transpose_segment(p, 8, p, 8, 4, 4, VM_BF16, nullptr, 0, nullptr), VM_ERR_BAD_ARG);

  int64_t bytes = -1;
  EXPECT(vm_lora_down_workspace(6280, 15360, 0, &bytes), VM_OK);
  EXPECT(bytes >= 0 ? 0 : 1, 0);
  EXPECT(vm_tn_skinny_workspace(6280, 15360, &bytes), VM_OK);
  EXPECT(bytes > 0 ? 0 : 1, 0);
  EXPECT(vm_tn_skinny_workspace(0, 64, &bytes), VM_ERR_BAD_ARG);
  EXPECT(vm_lora_down(nullptr, 0, nullptr, nullptr, 0, nullptr, 0, 8, 64, 64, nullptr, -1, 0.f, 0, nullptr, 0, nullptr), VM_ERR_BAD_ARG);
  EXPECT(vm_tn_skinny_bf16(nullptr, 0, 64, nullptr, 0, nullptr, nullptr, 0, VM_BF16, 0, 0, 8, nullptr, -1, nullptr, 1.f, 0.f, 0, nullptr, 0, nullptr),
         VM_ERR_BAD_ARG);

  // instance losses / matching costs / forked norms [r2]
  EXPECT(vm_box_match_cost(nullptr, 3, f, 6, 8, 5.f, 2.f, 2.f, 1, 2.f, 0.85f, nullptr), VM_ERR_BAD_ARG);
  EXPECT(vm_box_match_cost(nullptr, 0, f, 6, 8, 5.f, 2.f, 2.f, 1, 2.f, 0.85f, nullptr), VM_OK);               // no problems: nothing to do
  EXPECT(vm_instance_loss_fwd(f, f, f, nullptr, 2, 3, 2.f, 0.85f, f, nullptr), VM_ERR_BAD_ARG);                // no assignment
  EXPECT(vm_instance_loss_fwd(f, f, f, reinterpret_cast<const int64_t*>(p), 0, 3, 2.f, 0.85f, f, nullptr), VM_OK);
  EXPECT(vm_instance_loss_bwd(f, f, f, reinterpret_cast<const int64_t*>(p), 2, 3, 2.f, 0.85f, f, nullptr, f, f, nullptr), VM_ERR_BAD_ARG);
  EXPECT(vm_rmsnorm_bwd_res(p, p, p, f, p, nullptr, nullptr, 4, 64, VM_BF16, nullptr, nullptr), VM_ERR_BAD_ARG);   // residual gradient without dx
  EXPECT(vm_layernorm_bwd_res(p, p, p, f, f, p, nullptr, nullptr, nullptr, 4, 64, VM_F32, nullptr), VM_ERR_BAD_ARG);
  EXPECT(vm_rmsnorm_bwd_res(p, p, p, f, nullptr, p, nullptr, 4, 60, VM_BF16, nullptr, nullptr), VM_ERR_BAD_ARG);    // cols % 8

  // event-profiling bookkeeping (no events are created while the mask is 0)
  EXPECT(vm_prof_enable(0), VM_OK);
  EXPECT(vm_prof_stride(0), VM_ERR_BAD_ARG);
  EXPECT(vm_prof_stride(4), VM_OK);
  EXPECT(vm_prof_reset(), VM_OK);
  double ms = -1, fl = -1; int64_t n = -1;
  EXPECT(vm_prof_collect(9, &ms, &fl, &n), VM_ERR_BAD_ARG);
  EXPECT(vm_prof_collect(VM_PROF_GEMM_BF16, &ms, &fl, &n), VM_OK);
  EXPECT(n == 0 ? 0 : 1, 0);
  double by = -1;
  EXPECT(vm_prof_last_bytes(nullptr), VM_ERR_BAD_ARG);
  EXPECT(vm_prof_last_bytes(&by), VM_OK);
  std::printf("%s: %d failed checks\n", fails ? "FAILED" : "OK", fails);
  return fails ? 1 : 0;
}
