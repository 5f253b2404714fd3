import sys, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from mmmm_amd import kernels as K
from tools.bench_kernels import timeit
dev = torch.device('cuda:0')
_tile = int(os.environ.get('GEMM_TILE', '0'))          # tool-side knob: GEMM_TILE=128|192|256|-192 -> vm_gemm_force_tile_
if _tile:
    from mmmm_amd import hip as _hip
    assert _hip.lib().vm_gemm_force_tile_(_tile) == 0
tag = f"tile={_tile or 'auto'} dbg={os.environ.get('VM_GEMM_DEBUG','0')}"
for name, M, N, Kd in [('lm.qkv', 3648, 12288, 4096), ('lm.down', 3648, 4096, 11008), ('vit.fc1', 6280, 15360, 1792), ('sq.8192', 8192, 8192, 8192)]:
    a = torch.randn(M, Kd, device=dev).bfloat16()
    w = (torch.randn(N, Kd, device=dev) / 64).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ms = timeit(lambda: K.gemm(a, w, out=out), iters=30)
    print(tag, name, f'{ms*1e3:.0f} us', round(2 * M * N * Kd / ms / 1e9), 'TF', flush=True)
