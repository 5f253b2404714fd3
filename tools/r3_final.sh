#!/bin/bash
# end-of-round evidence on ONE box: (optional: counters of the dominant GEMM / attention in situ), GPU suite, smoke, default bench line, kernel trace
# usage (via gpurun): bash tools/r3_final.sh [pmc]
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
if [ "$1" = "pmc" ]; then
  bash tools/pmc_step.sh r3 > gpurun_out/r3_pmc.log 2>&1; tail -3 gpurun_out/r3_pmc.log
  cp gpurun_out/r3_gemm_traffic.json profiles/r3_gemm_traffic.json 2>/dev/null   # (so that the bench line below finds the fresh measurement)
fi
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r3_suite.log 2>&1; tail -3 gpurun_out/r3_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err; tail -c 600 gpurun_out/r3_bench_default.json
bash tools/profile_step.sh r3_v5 --also '' > /dev/null 2>&1; head -12 gpurun_out/r3_v5_timeline.md
