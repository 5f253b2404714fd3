python -m pytest tests/test_kernels_gpu.py -x -q -k "tn_skinny or lora" > gpurun_out/r3_t11.log 2>&1; tail -3 gpurun_out/r3_t11.log
python -m pytest tests/test_model_gpu.py tests/test_ddp_gpu.py -x -q > gpurun_out/r3_t11b.log 2>&1; tail -3 gpurun_out/r3_t11b.log
run() { python bench.py --no-cpu-baseline --steps 12 --warmup 3 --also '' 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$1', round(j['ms_per_step'],1), 'ms', round(j['roofline']['achieved']), 'TF', 'loss', j['loss'])" >> gpurun_out/r3_ab11.log; }
VM_F32_TN_WGRAD=0 VM_LIB_PATH=mmmm_amd/lib/libvividmed_hip_prev.so run "prev lib (HEAD: no dwdb-vec, no TN, group64)"
VM_TN_F32_TARGET=800 VM_LORA_KT1=0 run "new: group256 + TN800 + kt1=0"
VM_F32_TN_WGRAD=0 VM_LIB_PATH=mmmm_amd/lib/libvividmed_hip_prev.so run "prev lib"
VM_TN_F32_TARGET=800 VM_LORA_KT1=0 run "new: group256 + TN800 + kt1=0"
cat gpurun_out/r3_ab11.log
