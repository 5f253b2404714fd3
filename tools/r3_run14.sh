python -m pytest tests/test_kernels_gpu.py -x -q -k "attn" > gpurun_out/r3_t14.log 2>&1; tail -3 gpurun_out/r3_t14.log
VM_ATTN_DKV_NW=16 python -m pytest tests/test_kernels_gpu.py -x -q -k "attn" > gpurun_out/r3_t14b.log 2>&1; tail -3 gpurun_out/r3_t14b.log
python tools/bench_attn.py > gpurun_out/r3_bench_attn_nw16.log 2>&1; VM_ATTN_NW=8 python tools/bench_attn.py > gpurun_out/r3_bench_attn_nw8.log 2>&1; VM_ATTN_DKV_NW=16 python tools/bench_attn.py > gpurun_out/r3_bench_attn_dkv16.log 2>&1
tail -8 gpurun_out/r3_bench_attn_nw8.log; tail -8 gpurun_out/r3_bench_attn_nw16.log; tail -8 gpurun_out/r3_bench_attn_dkv16.log
run() { python bench.py --no-cpu-baseline --steps 12 --warmup 3 --also '' 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$1', round(j['ms_per_step'],1), 'ms', round(j['roofline']['achieved']), 'TF', 'attn', round(j['attention']['achieved_tflops']), 'loss', j['loss'])" >> gpurun_out/r3_ab14.log; }
VM_ATTN_NW=8 run "nw=8"
run "nw=16 (dkv 8)"
VM_ATTN_DKV_NW=16 run "nw=16 dkv 16"
VM_ATTN_NW=8 run "nw=8"
run "nw=16 (dkv 8)"
cat gpurun_out/r3_ab14.log
