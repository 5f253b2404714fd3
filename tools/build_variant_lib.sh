#!/bin/bash
# Build the CURRENT sources with extra -D flags into mmmm_amd/lib/libvividmed_hip_<name>.so for A/B runs inside one gpurun call
# (VM_LIB_PATH=mmmm_amd/lib/libvividmed_hip_<name>.so python ...). usage: bash tools/build_variant_lib.sh <name> -DFLAG1 -DFLAG2 ...
set -e
NAME=$1; shift
T=$(mktemp -d)
R=$(cd "$(dirname "$0")/.." && pwd)
for f in $R/mmmm_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DVM_KEEP_DENORMS "$@" -I$R/mmmm_amd/csrc -c $f -o $T/$(basename ${f%.hip}).o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/mmmm_amd/lib/libvividmed_hip_$NAME.so $T/*.o
rm -rf $T
ls -la $R/mmmm_amd/lib/libvividmed_hip_$NAME.so
