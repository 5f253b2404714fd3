#!/bin/bash
# A/B of library builds inside the training step, variants alternated in ONE call (boxes differ by +-3 %):
#   bash tools/ab_lib.sh "<name> <name> ..." [bench.py arguments]      ("default" = the shipped library; others: mmmm_amd/lib/libvividmed_hip_<name>.so,
#   built by tools/build_variant_lib.sh or tools/build_ref_lib.sh)
NAMES=$1; shift
for i in 1 2; do
  for v in $NAMES; do
    if [ "$v" = default ]; then unset VM_LIB_PATH; else export VM_LIB_PATH=mmmm_amd/lib/libvividmed_hip_$v.so; fi
    python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-events --no-peak-probe --also '' "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lib=$v', round(d['ms_per_step'],2), 'ms/step, host enqueue', round(d['host_enqueue_ms'],1))"
  done
done
