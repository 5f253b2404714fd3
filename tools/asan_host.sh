#!/bin/bash
# Host-side AddressSanitizer build + run of the C ABI's host code (no GPU needed): see tests/asan/abi_host_asan.cpp
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/vm_asan}
mkdir -p $OUT
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="-O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -DVM_KEEP_DENORMS -fsanitize=address -shared-libsan -Wno-option-ignored"
pids=()
for f in gemm gemm256 rowwise lora instance_loss; do
  $HIPCC $FLAGS -c $R/mmmm_amd/csrc/$f.hip -o $OUT/$f.o > $OUT/$f.log 2>&1 &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC $FLAGS -c $R/tests/asan/abi_host_asan.cpp -o $OUT/driver.o > $OUT/driver.log 2>&1 || { cat $OUT/driver.log; exit 1; }
$HIPCC --offload-arch=gfx950 -fsanitize=address -shared-libsan -Wno-option-ignored $OUT/driver.o $OUT/gemm.o $OUT/gemm256.o $OUT/rowwise.o $OUT/lora.o $OUT/instance_loss.o -o $OUT/abi_host_asan > $OUT/link.log 2>&1 || { cat $OUT/link.log; exit 1; }
RT=$(dirname $($HIPCC --print-file-name=libclang_rt.asan-x86_64.so 2>/dev/null || true))
ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 LD_LIBRARY_PATH=$RT:/opt/rocm/lib:/opt/rocm/lib/llvm/lib/clang/$(ls /opt/rocm/lib/llvm/lib/clang | head -1)/lib/linux:$LD_LIBRARY_PATH $OUT/abi_host_asan
