python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r3_gpu_suite.log 2>&1; tail -16 gpurun_out/r3_gpu_suite.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench_v1.json 2> gpurun_out/r3_bench_v1.err; cut -c1-1500 gpurun_out/r3_bench_v1.json
