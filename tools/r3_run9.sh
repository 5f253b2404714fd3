run() { python bench.py --no-cpu-baseline --steps 12 --warmup 3 --also '' 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$1', round(j['ms_per_step'],1), 'ms', round(j['roofline']['achieved']), 'TF', 'loss', j['loss'])" >> gpurun_out/r3_ab9.log; }
VM_F32_TN_WGRAD=0 run "tn=0"
VM_TN_F32_TARGET=800 run "tn=1 target 800"
VM_TN_F32_TARGET=1400 run "tn=1 target 1400"
VM_F32_TN_WGRAD=0 run "tn=0"
VM_TN_F32_TARGET=800 run "tn=1 target 800"
VM_TN_F32_TARGET=1400 run "tn=1 target 1400"
cat gpurun_out/r3_ab9.log
