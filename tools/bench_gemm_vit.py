"""A/B timing of the ViT-shaped GEMMs (short K, wide N): GEMM_TILE (this script -> vm_gemm_force_tile_) / VM_GEMM_DEBUG (debug builds of the library) are read once per process."""
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
shapes = [(6280, 15360, 1792, 64, True), (6280, 15360, 1792, 64, False), (6280, 15360, 1792, 0, False), (6280, 1792, 15360, 64, True),
          (6280, 5376, 1792, 64, True), (6280, 1792, 1792, 64, True), (3648, 11008, 4096, 64, False), (3648, 4096, 11008, 64, False)]
for M, N, Kd, K2, has_bias in shapes:
    a = torch.randn(M, Kd, device=dev).bfloat16()
    w = (torch.randn(N, Kd, device=dev) / 64).bfloat16()
    a2 = torch.randn(M, K2, device=dev).bfloat16() if K2 else None
    b2 = (torch.randn(N, K2, device=dev) / 64).bfloat16() if K2 else None
    bias = torch.randn(N, device=dev).bfloat16() if has_bias else None
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ms = timeit(lambda: K.gemm(a, w, a2=a2, b2=b2, alpha2=0.5, bias=bias, out=out), iters=30)
    tf = 2.0 * M * N * (Kd + K2) / ms / 1e9
    print(tag, f'M={M} N={N} K={Kd}+{K2} bias={int(has_bias)}', f'{ms*1e3:.0f} us {tf:.0f} TF', flush=True)
