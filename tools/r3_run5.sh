python -m pytest tests/test_kernels_gpu.py -x -q -k "lora or tn_skinny" > gpurun_out/r3_t5.log 2>&1; tail -3 gpurun_out/r3_t5.log
python -m pytest tests/test_model_gpu.py tests/test_truewidth_gpu.py tests/test_config0_gpu.py -x -q > gpurun_out/r3_t5b.log 2>&1; tail -3 gpurun_out/r3_t5b.log
for cfg in 0 1 0 1; do
  VM_FUSE_EW_LORA=$cfg VM_WGRAD_STREAM=0 python bench.py --no-cpu-baseline --steps 12 --warmup 3 --also '' 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('fuse=$cfg', round(j['ms_per_step'],1), 'ms', round(j['roofline']['achieved']), 'TF', 'loss', j['loss'])" >> gpurun_out/r3_ab5.log
done
cat gpurun_out/r3_ab5.log
