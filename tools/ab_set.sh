#!/bin/bash
# A/B of one module-level constant inside the training step, variants alternated in ONE call (boxes differ by +-3 %):
#   bash tools/ab_set.sh functional.WGRAD_SIDE_STREAM False True [bench.py arguments]
NAME=$1; A=$2; B=$3; shift 3
for i in 1 2; do
  for v in $A $B; do
    python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-events --no-peak-probe --also '' --set $NAME=$v "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$NAME=$v', round(d['ms_per_step'],2), 'ms/step, host enqueue min', round(d['host_enqueue_ms'],1), 'median', round(d.get('host_enqueue_median_ms', 0),1))"
  done
done
