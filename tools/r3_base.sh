set -x
python bench.py --no-cpu-baseline --steps 12 --warmup 3 > gpurun_out/r3_base_on.json 2> gpurun_out/r3_base_on.err
VM_WGRAD_STREAM=0 python bench.py --no-cpu-baseline --steps 12 --warmup 3 > gpurun_out/r3_base_off.json 2> gpurun_out/r3_base_off.err
VM_WGRAD_STREAM=0 bash tools/profile_step.sh r3_v0_noside > gpurun_out/r3_v0_noside.log 2>&1
bash tools/profile_step.sh r3_v0_side > gpurun_out/r3_v0_side.log 2>&1
python tools/host_bound.py > gpurun_out/r3_host_bound.log 2>&1
cat gpurun_out/r3_base_on.json gpurun_out/r3_base_off.json | cut -c1-400
