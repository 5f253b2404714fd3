#!/bin/bash
# counters of ONE attention-bench shape / variant (on the GPU box): tools/pmc_attn_bench.sh <shape idx> <variant> "<counters>" [more counter sets ...]
# each counter set is its own rocprofv3 pass (PMC slots: 8 SQ per pass); prints per-kernel sums divided by the number of dispatches
R=$PWD; SH=$1; VAR=$2; shift 2
cd /tmp && export TMPDIR=/tmp
# the host must not run ahead of a PMC pass (every dispatch is serialised and slow): with thousands of packets queued the profiler's
# intercept queue overflowed (SIGSEGV inside a launch / "AQL packet is malformed", then a hang) — one launch at a time, and a time limit
export AMD_SERIALIZE_KERNEL=3
i=0
for CS in "$@"; do
  rm -rf /tmp/pmc_ab_$i
  timeout 900 rocprofv3 --pmc $CS --output-format csv -d /tmp/pmc_ab_$i -o a -- $R/tools/ubench/attn_bench 2 $SH $VAR > /dev/null 2>&1
  python3 - /tmp/pmc_ab_$i <<'PY'
import csv, collections, re, sys, glob
f = glob.glob(sys.argv[1] + '/**/a_counter_collection.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"]
    if 'ref_fwd' in k: continue
    k = re.sub(r'\(.*', '', k)[:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k, v in agg.items():
    print(k, 'dispatches', len(n[k]), ' '.join(f'{c}={x / len(n[k]):.4e}' for c, x in sorted(v.items())))
PY
  i=$((i+1))
done
