#!/bin/bash
# end-of-round evidence on ONE box: counters of the dominant GEMM / attention in situ, GPU suite, default bench line, kernel trace
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
bash tools/pmc_step.sh r3 > gpurun_out/r3_pmc.log 2>&1; tail -3 gpurun_out/r3_pmc.log
cp gpurun_out/r3_gemm_traffic.json profiles/r3_gemm_traffic.json 2>/dev/null   # (so that the bench line below finds the fresh measurement)
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r3_suite.log 2>&1; tail -3 gpurun_out/r3_suite.log
timeout 900 python bench.py > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err; tail -c 1500 gpurun_out/r3_bench_default.json
bash tools/profile_step.sh r3_v4 --also '' > /dev/null 2>&1; head -12 gpurun_out/r3_v4_timeline.md
