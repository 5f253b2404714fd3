"""Attention yardstick: the hand-written bf16 kernels (vm_attn_fwd / bwd) against torch's scaled_dot_product_attention (whatever fused backend
ROCm's ATen picks for the shape: flash / memory-efficient; hd 112 is padded to 128 for it) on the step's shapes. Not a product path."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import torch.nn.functional as F
from mmmm_amd import kernels as K
from tools.bench_kernels import timeit
dev = torch.device('cuda:0')
for name, B, L, H, hd, causal in [('vit-e 8x785', 8, 785, 16, 112, False), ('decoder 8x456', 8, 456, 32, 128, True), ('3d 8x2049', 8, 2049, 16, 112, False),
                                  ('hr 4x3137', 4, 3137, 16, 112, False), ('hr3d 4x4609', 4, 4609, 16, 112, False)]:
    lens = [L] * B
    rows = B * L
    qkv = torch.randn(rows, 3 * H * hd, device=dev).bfloat16()
    q, k, v = qkv[:, :H * hd], qkv[:, H * hd:2 * H * hd], qkv[:, 2 * H * hd:]
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=dev)
    out, lse = K.attn_fwd(q, k, v, cu, L, H, hd, hd ** -0.5, causal)
    dout = torch.randn_like(out)
    ms = timeit(lambda: K.attn_fwd(q, k, v, cu, L, H, hd, hd ** -0.5, causal), iters=10, warm=2)
    msb = timeit(lambda: K.attn_bwd(q, k, v, out, lse, dout, cu, L, H, hd, hd ** -0.5, causal), iters=10, warm=2)
    fl = B * 4 * L * L * hd * H * (0.5 if causal else 1)
    line = f'{name:14s} hand-written fwd {ms*1e3:7.0f} us {fl/ms/1e9:5.0f} TF  bwd {msb*1e3:7.0f} us {2.5*fl/msb/1e9:5.0f} TF'
    for pad in ((0, 16) if hd == 112 else (0,)):
        try:
            hp = hd + pad
            qq = torch.randn(B, H, L, hp, device=dev, dtype=torch.bfloat16, requires_grad=True)
            kk = torch.randn(B, H, L, hp, device=dev, dtype=torch.bfloat16, requires_grad=True)
            vv = torch.randn(B, H, L, hp, device=dev, dtype=torch.bfloat16, requires_grad=True)
            o = F.scaled_dot_product_attention(qq, kk, vv, is_causal=causal, scale=hd ** -0.5)
            g = torch.randn_like(o)
            mf = timeit(lambda: F.scaled_dot_product_attention(qq, kk, vv, is_causal=causal, scale=hd ** -0.5), iters=10, warm=2)
            def fb():
                o2 = F.scaled_dot_product_attention(qq, kk, vv, is_causal=causal, scale=hd ** -0.5)
                o2.backward(g)
            mfb = timeit(fb, iters=10, warm=2)
            line += f' | SDPA hd{hp} fwd {mf*1e3:7.0f} us  fwd+bwd {mfb*1e3:7.0f} us (ours {1e3*(ms+msb):7.0f})'
        except Exception as e:      # noqa
            line += f' | SDPA hd{hd+pad}: {type(e).__name__}'
    print(line, flush=True)
