"""cost of the LoRA K-extension and of its dropout mask on the big GEMM shapes (isolated, 256x256 tile auto-selected)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from mmmm_amd import kernels as K
from tools.bench_kernels import timeit
dev = torch.device('cuda:0')
shapes = [(6280, 15360, 1792), (6280, 1792, 15360), (6280, 5376, 1792), (6280, 1792, 1792), (3648, 11008, 4096), (3648, 4096, 11008), (3648, 4096, 4096)]
for M, N, Kd in shapes:
    a = torch.randn(M, Kd, device=dev).bfloat16()
    w = (torch.randn(N, Kd, device=dev) / 64).bfloat16()
    a2 = torch.randn(M, 64, device=dev).bfloat16()
    b2 = (torch.randn(N, 64, device=dev) / 64).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = []
    for name, kw in [('plain', {}), ('ext', dict(a2=a2, b2=b2, alpha2=0.5)), ('ext+drop', dict(a2=a2, b2=b2, alpha2=0.5, drop_p=0.05, drop_seed=5))]:
        ms = timeit(lambda: K.gemm(a, w, out=out, **kw), iters=30)
        res.append(f'{name} {ms*1e3:.0f} us')
    print(f'M={M} N={N} K={Kd}: ' + ' | '.join(res), flush=True)
