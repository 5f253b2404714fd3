python -m pytest tests/test_kernels_gpu.py -x -q > gpurun_out/r3_t7.log 2>&1; tail -3 gpurun_out/r3_t7.log
python -m pytest tests/test_model_gpu.py tests/test_sam_gpu.py -x -q > gpurun_out/r3_t7b.log 2>&1; tail -3 gpurun_out/r3_t7b.log
for i in 1 2; do
python bench.py --no-cpu-baseline --steps 12 --warmup 3 --also '' 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('dwdb vec', round(j['ms_per_step'],1), 'ms', round(j['roofline']['achieved']), 'TF', 'loss', j['loss'])" >> gpurun_out/r3_ab7.log
done
cat gpurun_out/r3_ab7.log
