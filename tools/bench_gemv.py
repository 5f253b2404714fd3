"""skinny-M streaming linear (decode step): effective HBM bandwidth on the 7B shapes. usage: python tools/bench_gemv.py [M]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from mmmm_amd import kernels as K
from tools.bench_kernels import timeit

dev = torch.device('cuda:0')
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1
tot_b = tot_t = 0.0
for name, N, Kd in [('qkv', 12288, 4096), ('dense', 4096, 4096), ('gate', 11008, 4096), ('down', 4096, 11008), ('lm_head', 32008, 4096), ('lora_A', 64, 4096)]:
    # several distinct weight copies so that consecutive launches do not hit in the 256 MB Infinity Cache
    ws = [(torch.randn(N, Kd, device=dev) / 64).bfloat16() for _ in range(max(2, int(600e6 // (N * Kd * 2))))]
    x = torch.randn(M, Kd, device=dev).bfloat16()
    t = torch.randn(M, 64, device=dev).bfloat16()
    b2 = (torch.randn(N, 64, device=dev) / 8).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    i = [0]

    def run():
        i[0] = (i[0] + 1) % len(ws)
        K.gemv(x, ws[i[0]], a2=t, b2=b2, alpha2=0.5, out=out)
    ms = timeit(run, iters=40)
    nbytes = N * Kd * 2
    if name != 'lora_A':
        tot_b += nbytes * (32 if name != 'lm_head' else 1) * (2 if name == 'gate' else 1)
        tot_t += ms * (32 if name != 'lm_head' else 1) * (2 if name == 'gate' else 1)
    print(f'M={M} {name:8s} N={N:6d} K={Kd:6d}: {ms * 1e3:7.1f} us  {nbytes / ms / 1e9:6.2f} TB/s', flush=True)
print(f'weighted over one decode step: {tot_t:.2f} ms for {tot_b / 1e9:.1f} GB = {tot_b / tot_t / 1e9:.2f} TB/s')
