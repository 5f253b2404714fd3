#!/bin/bash
# SQ / TCC counters of the dominant GEMM (gemm256_k, both tile forms) in situ: one rocprofv3 --pmc pass per counter group over
# bench.py --depth-scale 0.1 --checkpointing reference (the step's GEMM shapes, fewer layers: counter collection serialises
# dispatches). Writes gpurun_out/r2_gemm_pmc.json (per-launch averages per kernel form).
R=$PWD; cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT" \
           "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pmc_g$i
  rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_g$i -o t -- python3 $R/bench.py --steps 1 --warmup 0 --depth-scale 0.1 --checkpointing reference --no-cpu-baseline --no-kernel-events > /tmp/pmc_g$i.log 2>&1
done
cd $R
python3 - <<PY
import csv, json, collections, re, glob
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in sorted(glob.glob("/tmp/pmc_g*/t_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(gemm256_k<[\w, ]+>)", r["Kernel_Name"])
        if not m: continue
        a = agg[m.group(1)][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
out = {k: {c: {"launches": n, "per_launch": v / n} for c, (n, v) in cs.items()} for k, cs in agg.items()}
for k, cs in out.items():
    g = lambda c: cs.get(c, {}).get("per_launch", float("nan"))
    cs["derived"] = {
        "mfma_busy_frac_of_sq_busy": g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_BUSY_CYCLES"),
        "wait_inst_any_frac_of_wave_cycles": g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"),
        "wait_inst_lds_frac_of_wave_cycles": g("SQ_WAIT_INST_LDS") / g("SQ_WAVE_CYCLES"),
        "l2_hit_rate": g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")),
        "lds_bank_conflict_frac_of_lds_active": g("SQ_LDS_BANK_CONFLICT") / g("SQ_ACTIVE_INST_LDS"),
    }
json.dump({"source": "rocprofv3 --pmc, one pass per counter group (tools/pmc_gemm.sh) over bench.py --depth-scale 0.1 --checkpointing reference: the benchmarked step's GEMM shapes in situ", "kernels": out}, open("gpurun_out/r2_gemm_pmc.json", "w"), indent=1)
for k, cs in out.items(): print(k, json.dumps(cs["derived"]))
PY
