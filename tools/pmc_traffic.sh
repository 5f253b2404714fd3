#!/bin/bash
# HBM traffic of the dominant GEMM kernel per launch (MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE in separate
# passes over a depth-scaled model (same GEMM shapes, fewer layers: PMC collection serialises every dispatch); FETCH_SIZE x2 on gfx950 for wide coalesced reads). Run on the GPU box; writes gpurun_out/gemm_traffic.json
R=$PWD; cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -o t -- python3 $R/bench.py --steps 1 --warmup 0 --depth-scale 0.1 --checkpointing reference --no-cpu-baseline --no-kernel-events > /dev/null 2>&1
done
cd $R
python3 - <<PY
import csv, json, collections, re
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open(f"gpurun_out/pmc_{c}/t_counter_collection.csv")))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        if r["Counter_Name"] != c: continue
        m = re.search(r"(gemm256_k<[\w, ]+>|gemm_nt_k<[\w, ]+>)", r["Kernel_Name"])
        if not m: continue
        a = agg[m.group(1)]; a[0] += 1; a[1] += float(r["Counter_Value"])
    out[c] = {k: {"launches": n, "sum": v} for k, (n, v) in agg.items()}
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/gemm_traffic_raw.json", "w"), indent=1)
PY
