"""fp32 products with a few dozen rows (the mask decoders' token-side linears, box / discriminator heads): us per launch.
Run once per library build (VM_LIB_PATH) — the 32-column tile of gemm.hip's `M <= 64` branch against 128 x 128 tiles."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from mmmm_amd import kernels as K
from tools.bench_kernels import timeit
dev = torch.device('cuda:0')
for M, N, Kd in [(40, 768, 768), (40, 2048, 768), (40, 768, 2048), (40, 384, 768), (8, 96, 768), (64, 768, 768), (7, 6, 768), (40, 768, 384)]:
    a = torch.randn(M, Kd, device=dev)
    w = torch.randn(N, Kd, device=dev) / 32
    b = torch.randn(N, device=dev)
    for mode in (3, 2):
        ms = timeit(lambda: K.gemm(a, w, bias=b, f32_split=mode), iters=50)
        print(f'[{M} x {N}] x {Kd} split-bf16 mode {mode}: {ms*1e3:6.1f} us', flush=True)
