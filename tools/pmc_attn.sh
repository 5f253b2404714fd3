#!/bin/bash
# usage (on the GPU box): tools/pmc_attn.sh "<counters>"  -> per-kernel sums for the attention kernels
R=$PWD; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_attn
rocprofv3 --pmc $1 --output-format csv -d $R/gpurun_out/pmc_attn -o a -- python3 $R/tools/bench_attn.py 3 > /dev/null 2>&1
cd $R
python3 - <<PY
import csv, collections, re
rows=list(csv.DictReader(open("gpurun_out/pmc_attn/a_counter_collection.csv")))
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    m=re.search(r'(attn\w+<\d+>)', r["Kernel_Name"])
    if not m: continue
    agg[m.group(1)][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in agg.items():
    print(k, ' '.join(f'{c}={x:.3e}' for c,x in sorted(v.items())))
PY
