#!/bin/bash
# functional.FORK_LINEAR off / on, alternated in one call: bash tools/ab_fork_linear.sh [bench.py arguments, e.g. --workload phase-grg-3d]
run() { python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-events --no-peak-probe --also "" "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2))"; }
for i in 1 2 3; do
  echo "FORK_LINEAR=False $(run --set functional.FORK_LINEAR=False "$@")"; echo "FORK_LINEAR=True  $(run --set functional.FORK_LINEAR=True "$@")"
done
