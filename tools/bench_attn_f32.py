"""fp32 attention of the SAM ViT-B encoders (head_dim 64, 12 heads, packed var-len): exact f32 MFMA vs split-bf16 products.
usage: python tools/bench_attn_f32.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmmm_amd import kernels as K  # noqa: E402
from tools.bench_kernels import timeit  # noqa: E402

dev = torch.device('cuda', 0)
H, hd = 12, 64
C = H * hd
for lens in ([784] * 4, [2048] * 4, [4608] * 2):
    T = sum(lens)
    qkv = torch.randn(T, 3 * C, device=dev)
    dout = torch.randn(T, C, device=dev)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=dev)
    dqkv = torch.empty_like(qkv)
    fl = 4.0 * sum(n * n for n in lens) * hd * H
    for mode in (0, 2, 3):
        out, lse = K.attn_f32_fwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], H, hd, hd ** -0.5, cu, max(lens), f32_split=mode)
        f = timeit(lambda: K.attn_f32_fwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], H, hd, hd ** -0.5, cu, max(lens), f32_split=mode), iters=20)
        b = timeit(lambda: K.attn_f32_bwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, lse, dout, H, hd, hd ** -0.5, cu, max(lens),
                                          grads=(dqkv[:, :C], dqkv[:, C:2 * C], dqkv[:, 2 * C:]), f32_split=mode), iters=20)
        print(f'lens {lens[0]} x {len(lens)}  f32_split={mode}: fwd {f * 1e3:.0f} us ({fl / f / 1e9:.0f} TF)  bwd {b * 1e3:.0f} us ({2.5 * fl / b / 1e9:.0f} TF)', flush=True)
