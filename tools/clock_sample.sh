#!/bin/bash
# shader clock / power of the GPU while a command runs (rocm-smi sampled twice a second): is the dominant GEMM's rate a clock story?
# usage (GPU box): bash tools/clock_sample.sh <out.txt> <command ...>
OUT=$1; shift
"$@" > /dev/null 2>&1 &
PID=$!
: > $OUT
while kill -0 $PID 2>/dev/null; do
  echo "t=$(date +%s.%N)" >> $OUT
  /opt/rocm/bin/rocm-smi --showclocks --showpower --showuse --showtemp 2>/dev/null | grep -i "sclk\|mclk\|power\|GPU use\|junction\|fclk" >> $OUT
  sleep 0.5
done
wait $PID
