import sys, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from mmmm_amd import kernels as K
from tools.bench_kernels import timeit
dev = torch.device('cuda:0')
tag = f"tile={os.environ.get('VM_GEMM_TILE','auto')} dbg={os.environ.get('VM_GEMM_DEBUG','0')}"
for M, N in [(3648, 12288), (6280, 15360)]:
    for Kd in (64, 1792):
        a = torch.randn(M, Kd, device=dev).bfloat16()
        w = (torch.randn(N, Kd, device=dev) / 64).bfloat16()
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ms = timeit(lambda: K.gemm(a, w, out=out), iters=30)
        msz = timeit(lambda: out.zero_(), iters=30)
        print(tag, f'M={M} N={N} K={Kd}', f'{ms*1e3:.0f} us', f'(zero_ {msz*1e3:.0f} us)', flush=True)
