run() { python bench.py --no-cpu-baseline --steps 12 --warmup 3 --also '' 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$1', round(j['ms_per_step'],1), 'ms', round(j['roofline']['achieved']), 'TF', 'loss', j['loss'])" >> gpurun_out/r3_ab10.log; }
run "base"
VM_LORA_KT1=0 run "kt1=0 (always split)"
VM_LORA_KT1=0 VM_LORA_WANT=768 run "kt1=0 want=768"
VM_LORA_WANT=768 run "want=768"
run "base"
VM_LORA_KT1=0 run "kt1=0 (always split)"
cat gpurun_out/r3_ab10.log
