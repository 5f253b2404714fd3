python -m pytest tests/test_kernels_gpu.py -x -q -k "attention or attn or flash" > gpurun_out/r3_t15.log 2>&1; tail -3 gpurun_out/r3_t15.log
for st in 0 1; do echo "stagger=$st"; VM_ATTN_STAGGER=$st python tools/bench_attn.py 2>&1 | grep -v amdgpu; done
