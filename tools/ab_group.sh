for i in 1 2; do
  for v in old new; do
    if [ $v = old ]; then EXTRA="--set hip.TN_GROUP_MAX=24 --set functional.WGRAD_GROUP_SLOTS=100000"; else EXTRA=""; fi
    python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-events --no-peak-probe --also '' $EXTRA "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('group=$v', round(d['ms_per_step'],2), 'ms/step, host enqueue', round(d['host_enqueue_ms'],1))"
  done
done
