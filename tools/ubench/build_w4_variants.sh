#!/bin/bash
# tools/ubench/build_w4_variants.sh "0 1 2 4 8 256 ..." : one w4_core_bench_<bits> binary per VM_W4_EXPERIMENT value (built in parallel)
set -e
cd "$(dirname "$0")/../.."
F="-O3 -std=c++17 --offload-arch=gfx950 -DVM_KEEP_DENORMS -Wno-unused-value -Immmm_amd/csrc -Iinclude"
for v in ${1:-0}; do
  /opt/rocm/bin/hipcc $F -DVM_W4_EXPERIMENT=$v $W4_EXTRA tools/ubench/w4_core_bench.hip -Lmmmm_amd/lib -lvividmed_hip -Wl,-rpath,'$ORIGIN/../../mmmm_amd/lib' -o tools/ubench/w4_core_bench_$v &
done
wait
ls -la tools/ubench/w4_core_bench_*
