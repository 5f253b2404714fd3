#!/bin/bash
# HBM traffic of the dominant GEMM kernel per launch (MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE in separate
# passes; FETCH_SIZE x2 on gfx950 for wide coalesced reads), in the BENCHMARKED configuration: bench.py defaults (phase-vg-448, batch 8,
# full depth, HBM-budget checkpointing), 2 timed steps — PMC collection serialises every dispatch, the planning / calibration steps run
# the same GEMM shapes. Run on the GPU box; writes gpurun_out/gemm_traffic.json
R=$PWD; cd /tmp && export TMPDIR=/tmp
# the host must not run ahead of a PMC pass (every dispatch is serialised and slow): with thousands of packets queued the profiler's
# intercept queue overflowed (SIGSEGV inside a launch / "AQL packet is malformed", then a hang) — one launch at a time, and a time limit
export AMD_SERIALIZE_KERNEL=3
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c
  timeout 900 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -o t -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-kernel-events --no-peak-probe --also "" > $R/gpurun_out/pmc_$c.log 2>&1
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
        m = re.search(r"(gemm256w?_k<[\w, ]+>|gemm_nt_k<[\w, ]+>)", r["Kernel_Name"])
        if not m: continue
        a = agg[m.group(1)]; a[0] += 1; a[1] += float(r["Counter_Value"])
    out[c] = {k: {"launches": n, "sum": v} for k, (n, v) in agg.items()}
json.dump(out, open("gpurun_out/gemm_traffic_raw.json", "w"), indent=1)
forms = sorted(k for k in out["FETCH_SIZE"] if k.startswith("gemm256w_k<") or (k.startswith("gemm256_k<false") and "true" not in k.split(",")[-1]))
n = sum(out["FETCH_SIZE"][k]["launches"] for k in forms)
fetch_kb = sum(out["FETCH_SIZE"][k]["sum"] for k in forms) / n
write_kb = sum(out["WRITE_SIZE"][k]["sum"] for k in forms) / max(1, sum(out["WRITE_SIZE"][k]["launches"] for k in forms))
res = {
 "kernel": " + ".join(forms) + " (vm_gemm_bf16, 256x256 and 192x256 tile forms)",
 "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/pmc_traffic.sh) over the BENCHMARKED configuration: python bench.py (phase-vg-448, batch 8, full depth, HBM-budget checkpointing), planning + calibration + 2 timed steps",
 "launches": n, "fetch_size_kb_per_launch_raw": fetch_kb, "write_size_kb_per_launch": write_kb, "gfx950_fetch_correction": 2.0,
 "bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024,
 "per_form": {k: {"launches": out["FETCH_SIZE"][k]["launches"],
                  "bytes_per_launch": (2.0 * out["FETCH_SIZE"][k]["sum"] / out["FETCH_SIZE"][k]["launches"] + out["WRITE_SIZE"][k]["sum"] / out["WRITE_SIZE"][k]["launches"]) * 1024} for k in forms},
 "note": "memory-side (fabric) bytes of the L2s: Infinity Cache hits are counted (MI355X_MICROARCH.md, HBM section), so this is an upper bound of HBM traffic",
}
import sys
sys.path.insert(0, ".")
import bench
res["source_digest"] = bench.gemm_source_digest()          # bench.py only trusts a measurement taken on the sources it runs
json.dump(res, open("gpurun_out/gemm_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
