set -x
python -m pytest tests/test_config0_gpu.py -x -q -s > gpurun_out/r3_config0_test.log 2>&1; echo "rc=$?" >> gpurun_out/r3_config0_test.log
python -m pytest tests/test_sam_gpu.py tests/test_truewidth_gpu.py -x -q -s -k "parameter_gradients or f13_training_step" > gpurun_out/r3_sam_grad.log 2>&1
VM_ENC_F32_SPLIT=3 python -m pytest tests/test_sam_gpu.py -x -q -s -k "parameter_gradients" > gpurun_out/r3_sam_grad_split3.log 2>&1
VM_WGRAD_STREAM=0 python tools/gemm_shapes.py > gpurun_out/r3_gemm_shapes_v0.md 2>&1
tail -5 gpurun_out/r3_config0_test.log
