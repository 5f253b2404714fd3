python -m pytest tests/test_kernels_gpu.py -x -q -k "lora" > gpurun_out/r3_t4.log 2>&1; tail -3 gpurun_out/r3_t4.log
for cfg in 0 1 0 1; do
  VM_LORA_DOWN_DEEP=$cfg VM_WGRAD_STREAM=0 python bench.py --no-cpu-baseline --steps 12 --warmup 3 --also '' 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('deep=$cfg', round(j['ms_per_step'],1), 'ms', round(j['roofline']['achieved']), 'TF')" >> gpurun_out/r3_ab4.log
done
cat gpurun_out/r3_ab4.log
