#!/bin/bash
# end-of-round evidence on ONE box: GPU suite, smoke, default bench line (with its children), attention A/B bench
# usage (via gpurun): bash tools/r4_final.sh
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r4_suite.log 2>&1; tail -3 gpurun_out/r4_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1200 python bench.py > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err; tail -c 400 gpurun_out/r4_bench_default.json; echo
tools/ubench/attn_bench 8 > gpurun_out/r4_attn_bench.txt 2>&1; grep -c OK gpurun_out/r4_attn_bench.txt
