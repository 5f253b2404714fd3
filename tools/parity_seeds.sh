#!/bin/bash
# Is a model-level parity number that moved between two kernel builds a property of the kernels or of the rounding pattern?
# Runs tests/test_config0_gpu.py (iSAM case, report only) for several weight seeds with the current library and with a reference build
# (tools/build_ref_lib.sh <rev> r3 -> mmmm_amd/lib/libvividmed_hip_r3.so) and prints e_hip / e_ref per watched tensor.
# usage (on the GPU box): bash tools/parity_seeds.sh "0 1 2"
SEEDS=${1:-"0 1 2"}
for S in $SEEDS; do
  for LIB in new r3; do
    if [ $LIB = r3 ]; then export VM_LIB_PATH=$PWD/mmmm_amd/lib/libvividmed_hip_r3.so; else unset VM_LIB_PATH; fi
    VM_PARITY_NOASSERT=1 VM_PARITY_SEED=$S VM_PARITY_REPORT_NAME=parity_${LIB}_$S.json python -m pytest tests/test_config0_gpu.py -q -k "iSAM or zz_report" > /dev/null 2>&1
  done
done
python3 - <<PY
import json, glob
rows = {}
for f in sorted(glob.glob('gpurun_out/parity_*_*.json')):
    _, lib, seed = f[:-5].rsplit('_', 2)
    for k, v in json.load(open(f)).items():
        if isinstance(v, dict) and 'e_ref' in v and 'e_hip' in v and v['e_ref'] > 0 and ('grad' in k or 'prompts' in k or 'logits' in k):
            rows.setdefault(k, {})[(lib.split('/')[-1], seed)] = v['e_hip'] / v['e_ref']
keys = sorted({c for r in rows.values() for c in r})
print('e_hip / e_ref per tensor; columns:', ' '.join(f'{l}:{s}' for l, s in keys))
for k, r in rows.items():
    print(f'{k[:90]:90s}', ' '.join(f'{r.get(c, float("nan")):6.2f}' for c in keys))
PY
