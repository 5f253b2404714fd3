#!/bin/bash
# SQ / TCC counters of the dominant GEMM (gemm256_k, every grid it is launched with) and of the bf16 attention kernels IN SITU, in the
# BENCHMARKED configuration: bench.py defaults (phase-vg-448, batch 8, full depth, HBM-budget checkpointing), 1 timed step after the
# planning step (counter collection serialises the dispatches it covers: only the kernels named in --kernel-include-regex are collected,
# and the calibration steps are skipped: they run the same shapes). One rocprofv3 --pmc pass
# per counter group (MI355X_MICROARCH.md 'rocprofv3 PMC slots'); FETCH_SIZE and WRITE_SIZE in passes of their own.
# Writes gpurun_out/<tag>_pmc.json (per kernel and grid: per-launch averages + derived ratios) and <tag>_gemm_traffic.json.
# usage (via gpurun): bash tools/pmc_step.sh <tag>
TAG=${1:-r4}
R=$PWD; cd /tmp && export TMPDIR=/tmp
# the host must not run ahead of a PMC pass (every dispatch is serialised and slow): with thousands of packets queued the profiler's
# intercept queue overflowed (SIGSEGV inside a launch / "AQL packet is malformed", then a hang) — one launch at a time, and a time limit
export AMD_SERIALIZE_KERNEL=3
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC" \
           "TCC_HIT_sum TCC_MISS_sum" \
           "FETCH_SIZE" \
           "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pmc_g$i
  timeout 600 rocprofv3 --pmc $grp --kernel-include-regex 'gemm256_k|attn16_|fwd_k' --output-format csv -d /tmp/pmc_g$i -o t -- python3 $R/bench.py --steps 1 --warmup 0 --no-calibrate --no-cpu-baseline --no-kernel-events --no-peak-probe --also '' > /tmp/pmc_g$i.log 2>&1
  tail -2 /tmp/pmc_g$i.log | cut -c1-200
done
cd $R
python3 - "$TAG" <<'PY'
import csv, json, collections, re, glob, sys, hashlib, pathlib
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in sorted(glob.glob("/tmp/pmc_g*/t_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(gemm256_k<[\w, ]+>|attn16_\w+<\d+>|fwd_k<\d+, \w+>)", r["Kernel_Name"])
        if not m: continue
        grid = int(r.get("Grid_Size", 0) or 0) // max(int(r.get("Workgroup_Size", 1) or 1), 1)
        a = agg[f"{m.group(1)} x{grid}"][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
out = {k: {c: {"launches": n, "per_launch": v / n} for c, (n, v) in cs.items()} for k, cs in agg.items()}
for k, cs in out.items():
    g = lambda c: cs.get(c, {}).get("per_launch", float("nan"))
    act = g("GRBM_GUI_ACTIVE") / 8            # cycles the chip was active per launch (sum over 8 XCDs)
    cs["derived"] = {
        "mfma_busy_frac_of_active_simd_cycles": g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024 * act),
        "wait_any_frac_of_wave_cycles (s_waitcnt / barrier)": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"),
        "wait_inst_any_frac_of_wave_cycles (issue stall)": g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"),
        "active_inst_any_frac_of_wave_cycles": g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"),
        "wait_inst_lds_frac_of_wave_cycles": g("SQ_WAIT_INST_LDS") / g("SQ_WAVE_CYCLES"),
        "valu_inst_frac_of_wave_cycles": g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES"),
        "l2_hit_rate": g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")),
        "lds_bank_conflict_frac_of_lds_idx_active": g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"),
        "bytes_per_launch (FETCH_SIZE KB x2 gfx950 + WRITE_SIZE KB)": (2.0 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024,
    }
h = hashlib.sha256()
for f in ("gemm256.hip", "gemm.hip", "gemm_common.hpp", "vm_tile.hpp"):
    h.update((pathlib.Path("mmmm_amd/csrc") / f).read_bytes())
digest = h.hexdigest()[:16]
src = ("rocprofv3 --pmc, one pass per counter group (tools/pmc_step.sh) over the BENCHMARKED configuration: python bench.py (phase-vg-448, batch 8, "
       "full depth, HBM-budget checkpointing; --no-calibrate), planning step + 1 timed step, counters collected for gemm256_k / attn16_* only")
json.dump({"source": src, "source_digest": digest, "kernels": out}, open(f"gpurun_out/{tag}_pmc.json", "w"), indent=1)
gem = {k: v for k, v in out.items() if k.startswith("gemm256_k<false") and "FETCH_SIZE" in v and "WRITE_SIZE" in v}
n = sum(v["FETCH_SIZE"]["launches"] for v in gem.values())
by = sum(v["derived"]["bytes_per_launch (FETCH_SIZE KB x2 gfx950 + WRITE_SIZE KB)"] * v["FETCH_SIZE"]["launches"] for v in gem.values()) / max(n, 1)
json.dump({"kernel": "gemm256_k<false, 3|4, ...> (vm_gemm_bf16, 256x256 and 192x256 tile forms), all grids", "source": src, "source_digest": digest,
           "launches": n, "gfx950_fetch_correction": 2.0, "bytes_per_launch": by,
           "per_grid": {k: {"launches": v["FETCH_SIZE"]["launches"], "bytes_per_launch": v["derived"]["bytes_per_launch (FETCH_SIZE KB x2 gfx950 + WRITE_SIZE KB)"]} for k, v in gem.items()},
           "note": "memory-side (fabric) bytes of the L2s: Infinity Cache hits are counted (MI355X_MICROARCH.md, HBM section), so this is an upper bound of HBM traffic"},
          open(f"gpurun_out/{tag}_gemm_traffic.json", "w"), indent=1)
for k, cs in sorted(out.items()): print(k, json.dumps({a: round(b, 4) if isinstance(b, float) else b for a, b in cs["derived"].items()}))
PY
