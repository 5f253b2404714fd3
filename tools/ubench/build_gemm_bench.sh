#!/bin/bash
# gemm_bench: the shipped library's scheduler A/B (DP / stream-K / auto) + check.  gemm_bench_dbg: the same bench with the two GEMM
# translation units compiled IN under -DVM_GEMM_DEBUG_BUILD, which adds the knock-out timing variants (no fix-up, no operand traffic).
set -e
cd "$(dirname "$0")/../.."
F="-O3 -std=c++17 --offload-arch=gfx950 -DVM_KEEP_DENORMS -Wno-unused-value"
/opt/rocm/bin/hipcc $F tools/ubench/gemm_bench.hip -Lmmmm_amd/lib -lvividmed_hip -Wl,-rpath,'$ORIGIN/../../mmmm_amd/lib' -o tools/ubench/gemm_bench
if [ "$1" = "dbg" ]; then
  /opt/rocm/bin/hipcc $F -DGB_DEBUG -DVM_GEMM_DEBUG_BUILD tools/ubench/gemm_bench.hip mmmm_amd/csrc/gemm.hip mmmm_amd/csrc/gemm256.hip -o tools/ubench/gemm_bench_dbg
fi
if [ "$1" = "w4" ]; then
  /opt/rocm/bin/hipcc $F tools/ubench/gemm_w4_bench.hip -Lmmmm_amd/lib -lvividmed_hip -Wl,-rpath,'$ORIGIN/../../mmmm_amd/lib' -o tools/ubench/gemm_w4_bench
fi
