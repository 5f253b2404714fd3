#!/bin/bash
# Build the kernels of another git revision into mmmm_amd/lib/libvividmed_hip_<name>.so for A/B runs inside one gpurun call
# (VM_LIB_PATH=mmmm_amd/lib/libvividmed_hip_<name>.so python bench.py ...). The C ABI must be compatible with the current Python side.
# usage: bash tools/build_ref_lib.sh <git-rev> <name>
set -e
REV=$1; NAME=$2
T=$(mktemp -d)
git -C /root/repo archive $REV mmmm_amd/csrc include mmmm_amd/build.py | tar -x -C $T
cd $T
for f in mmmm_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DVM_KEEP_DENORMS -c $f -o ${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o /root/repo/mmmm_amd/lib/libvividmed_hip_$NAME.so mmmm_amd/csrc/*.o
rm -rf $T
ls -la /root/repo/mmmm_amd/lib/libvividmed_hip_$NAME.so
