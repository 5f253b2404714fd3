python -m pytest tests/test_fp8_gpu.py -x -q -s -k "trains_like" 2>&1 | grep -E "loss over|largest|passed|failed"
VM_F32_TN_WGRAD=0 VM_LIB_PATH=mmmm_amd/lib/libvividmed_hip_prev.so python -m pytest tests/test_fp8_gpu.py -x -q -s -k "trains_like" 2>&1 | grep -E "loss over|largest|passed|failed"
VM_WGRAD_GROUP=0 python -m pytest tests/test_fp8_gpu.py -x -q -s -k "trains_like" 2>&1 | grep -E "loss over|largest|passed|failed"
VM_WGRAD_GROUP=0 VM_WGRAD_STREAM=1 VM_F32_TN_WGRAD=0 VM_LIB_PATH=mmmm_amd/lib/libvividmed_hip_prev.so python -m pytest tests/test_fp8_gpu.py -x -q -s -k "trains_like" 2>&1 | grep -E "loss over|largest|passed|failed"
