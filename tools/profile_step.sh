#!/bin/bash
# kernel trace + stats of a few steps of bench.py on the GPU box; writes gpurun_out/<tag>_timeline.md and <tag>_kernel_stats.md
# usage (via gpurun): bash tools/profile_step.sh <tag> [bench.py arguments]   (--also '' always: child benchmarks would inherit the profiler)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o vm -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-events --no-peak-probe --also '' "$@" > /tmp/prof_$TAG.log 2>&1
grep '^{' /tmp/prof_$TAG.log | cut -c1-260
cd $R
TRACE=$(find /tmp/prof_$TAG -name '*kernel_trace.csv' | head -1)
STATS=$(find /tmp/prof_$TAG -name '*kernel_stats.csv' | head -1)
python3 tools/step_timeline.py $TRACE gpurun_out/${TAG}_timeline.md > /dev/null
python3 tools/summarize_prof.py $STATS gpurun_out/${TAG}_kernel_stats.md "$TAG: bench.py --steps 4 --warmup 2 $* (planning + calibration steps included)" > /dev/null
python3 tools/trace_by_grid.py $TRACE gpurun_out/${TAG}_by_grid.md > /dev/null
head -1 $TRACE > gpurun_out/${TAG}_trace_header.csv
head -40 gpurun_out/${TAG}_timeline.md
