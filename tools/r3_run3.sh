set -x
python -m pytest tests/test_kernels_gpu.py -x -q -k "tn_skinny" > gpurun_out/r3_t3.log 2>&1; tail -3 gpurun_out/r3_t3.log
python -m pytest tests/test_model_gpu.py tests/test_ddp_gpu.py tests/test_truewidth_gpu.py -x -q > gpurun_out/r3_t3b.log 2>&1; tail -3 gpurun_out/r3_t3b.log
B="python bench.py --no-cpu-baseline --steps 12 --warmup 3 --also ''"
for cfg in "0 1" "0 0" "1 0" "1 1" "1 0" "0 0"; do
  set -- $cfg
  VM_WGRAD_GROUP=$1 VM_WGRAD_STREAM=$2 python bench.py --no-cpu-baseline --steps 12 --warmup 3 --also '' 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('group=$1 side=$2', round(j['ms_per_step'],1), 'ms', round(j['roofline']['achieved']), 'TF', 'host', round(j['host_enqueue_ms'],1))" >> gpurun_out/r3_ab3.log
done
cat gpurun_out/r3_ab3.log
