python -m pytest tests/test_kernels_gpu.py -x -q -k "lora" > gpurun_out/r3_t12.log 2>&1; tail -3 gpurun_out/r3_t12.log
python -m pytest tests/test_model_gpu.py tests/test_truewidth_gpu.py -x -q > gpurun_out/r3_t12b.log 2>&1; tail -3 gpurun_out/r3_t12b.log
run() { python bench.py --no-cpu-baseline --steps 12 --warmup 3 --also '' 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$1', round(j['ms_per_step'],1), 'ms', round(j['roofline']['achieved']), 'TF', 'loss', j['loss'])" >> gpurun_out/r3_ab12.log; }
VM_LORA_D16_MAXK=0 run "d16 off"
VM_LORA_D16_MAXK=2048 run "d16 K<=2048"
VM_LORA_D16_MAXK=4096 run "d16 K<=4096"
VM_LORA_D16_MAXK=5376 run "d16 K<=5376"
VM_LORA_D16_MAXK=0 run "d16 off"
VM_LORA_D16_MAXK=4096 run "d16 K<=4096"
cat gpurun_out/r3_ab12.log
