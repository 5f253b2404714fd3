python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm_tn_f32" > gpurun_out/r3_t8.log 2>&1; tail -3 gpurun_out/r3_t8.log
python -m pytest tests/test_sam_gpu.py tests/test_config0_gpu.py -x -q > gpurun_out/r3_t8b.log 2>&1; tail -3 gpurun_out/r3_t8b.log
run() { python bench.py --no-cpu-baseline --steps 12 --warmup 3 --also '' 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$1', round(j['ms_per_step'],1), 'ms', round(j['roofline']['achieved']), 'TF', 'loss', j['loss'])" >> gpurun_out/r3_ab8.log; }
VM_F32_TN_WGRAD=0 VM_LIB_PATH=mmmm_amd/lib/libvividmed_hip_prev.so run "prev-lib(dwdb old) tn=0"
VM_F32_TN_WGRAD=0 run "new-lib tn=0"
VM_F32_TN_WGRAD=1 run "new-lib tn=1"
VM_F32_TN_WGRAD=0 VM_LIB_PATH=mmmm_amd/lib/libvividmed_hip_prev.so run "prev-lib(dwdb old) tn=0"
VM_F32_TN_WGRAD=0 run "new-lib tn=0"
VM_F32_TN_WGRAD=1 run "new-lib tn=1"
cat gpurun_out/r3_ab8.log
