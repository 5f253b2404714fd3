"""lora_down / gemm_tn streaming kernels at the shapes of the training step: time and effective HBM bandwidth."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from mmmm_amd import kernels as K
from tools.bench_kernels import timeit
dev = torch.device('cuda:0')
for M, Kd in [(6280, 15360), (6280, 1792), (6280, 5376), (3648, 4096), (3648, 11008), (3648, 12288)]:
    x = torch.randn(M, Kd, device=dev).bfloat16()
    A = (torch.randn(64, Kd, device=dev) / 64).bfloat16()
    for dp in (0.0, 0.05):
        ms = timeit(lambda: K.lora_down(x, A, drop_p=dp, drop_seed=3), iters=30)
        print(f'lora_down M={M} K={Kd} drop={dp}: {ms*1e3:.1f} us  {M*Kd*2/ms/1e9:.2f} TB/s', flush=True)
    u = torch.randn(M, 64, device=dev).bfloat16()
    for dp in (0.0, 0.05):
        ms = timeit(lambda: K.gemm_tn(u, x, alpha=0.5, drop_p=dp, drop_seed=3), iters=30)
        print(f'gemm_tn [64 x {Kd}] over {M} rows drop={dp}: {ms*1e3:.1f} us  {M*Kd*2/ms/1e9:.2f} TB/s', flush=True)
    ms = timeit(lambda: K.gemm_tn(x, u, alpha=0.5), iters=30)
    print(f'gemm_tn [{Kd} x 64] over {M} rows: {ms*1e3:.1f} us  {M*Kd*2/ms/1e9:.2f} TB/s', flush=True)
    for dp in (0.0, 0.05):
        ms = timeit(lambda: K.tn_skinny(x, u, transpose_out=True, alpha=0.5, drop_p=dp, drop_seed=3), iters=30)
        print(f'tn_skinny dA [64 x {Kd}] over {M} rows drop={dp}: {ms*1e3:.1f} us  {M*Kd*2/ms/1e9:.2f} TB/s', flush=True)
    ms = timeit(lambda: K.tn_skinny(x, u, transpose_out=False, alpha=0.5), iters=30)
    print(f'tn_skinny dB [{Kd} x 64] over {M} rows: {ms*1e3:.1f} us  {M*Kd*2/ms/1e9:.2f} TB/s', flush=True)
