"""A/B of the K-split rank-64 projection: one launch with the last-arriver reduction (round 5) vs partials + lora_reduce_k, interleaved in one
process at the shapes of the training step (and the decode step's M = 1 .. 8)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from mmmm_amd import hip, kernels as K
from tools.bench_kernels import timeit
dev = torch.device('cuda:0')
lib = hip.lib()
for M, Kd in [(6280, 1792), (6280, 5376), (6280, 15360), (3648, 4096), (3648, 11008), (3648, 12288), (16392, 1792), (8, 4096), (1, 4096)]:
    x = torch.randn(M, Kd, device=dev).bfloat16()
    A = (torch.randn(64, Kd, device=dev) / 64).bfloat16()
    for dp in (0.0, 0.05):
        res = {}
        for rnd in range(3):
            for two in (1, 2):
                lib.vm_lora_down_two_kernels_(two)
                ms = timeit(lambda: K.lora_down(x, A, drop_p=dp, drop_seed=3), iters=40)
                res.setdefault(two, []).append(ms * 1e3)
        lib.vm_lora_down_two_kernels_(0)
        print(f'lora_down M={M} K={Kd} drop={dp}: two kernels {min(res[1]):.1f} us   one launch {min(res[2]):.1f} us', flush=True)
