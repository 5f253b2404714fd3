"""GELU forward / backward on one ViT-E block's activations [6280 x 15360] bf16: table kernels vs erf arithmetic
(argument 2: 0 = the erf arithmetic kernels through vm_gelu_table_)."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from mmmm_amd import kernels as K
dev = torch.device('cuda:0')
sig = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
table = int(sys.argv[2]) if len(sys.argv) > 2 else 1
from mmmm_amd import hip as _hip
_hip.lib().vm_gelu_table_(table)
h = [(torch.randn(6280, 15360, device=dev) * sig).bfloat16() for _ in range(3)]
dy = [torch.randn(6280, 15360, device=dev).bfloat16() for _ in range(3)]
for name, fn in (('fwd', lambda i: K.gelu(h[i % 3])), ('bwd', lambda i: K.gelu_bwd(h[i % 3], dy[i % 3]))):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(30):
        fn(i)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 30 * 1e3
    byts = h[0].numel() * 2 * (2 if name == 'fwd' else 3)
    print(f"table={table} sigma {sig:g} gelu {name}: {us:.1f} us, {byts / us / 1e6:.2f} TB/s", flush=True)
