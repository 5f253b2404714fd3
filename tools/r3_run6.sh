for cfg in 0 1 2 4 0; do
  VM_FUSE_EW_LORA=$cfg VM_WGRAD_STREAM=0 python bench.py --no-cpu-baseline --steps 12 --warmup 3 --also '' 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('fuse=$cfg', round(j['ms_per_step'],1), 'ms', round(j['roofline']['achieved']), 'TF', 'loss', j['loss'])" >> gpurun_out/r3_ab6.log
done
cat gpurun_out/r3_ab6.log
