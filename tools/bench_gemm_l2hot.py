"""Does the 256x256 GEMM main loop wait on operand arrival? Same shapes, three operand placements: real matrices, every row
aliased to row 0 (stride 0: the whole operand stream is L2 / L1 hot, data still random), zero-record descriptors
(VM_GEMM_DEBUG=1: no traffic at all, zeros -> lower MFMA power)."""
import sys, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from mmmm_amd import kernels as K
from tools.bench_kernels import timeit
dev = torch.device('cuda:0')
for M, N, Kd in [(6280, 15360, 1792), (3648, 4096, 11008), (6280, 1792, 15360)]:
    a = torch.randn(M, Kd, device=dev).bfloat16()
    w = (torch.randn(N, Kd, device=dev) / 64).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ms = timeit(lambda: K.gemm(a, w, out=out), iters=30)
    a0 = a[:1].expand(M, Kd)
    w0 = w[:1].expand(N, Kd)
    ms0 = timeit(lambda: K.gemm(a0, w0, out=out), iters=30)
    fl = 2.0 * M * N * Kd
    print(f'M={M} N={N} K={Kd}: real {ms*1e3:.0f} us {fl/ms/1e9:.0f} TF | rows aliased (cache-hot) {ms0*1e3:.0f} us {fl/ms0/1e9:.0f} TF', flush=True)
