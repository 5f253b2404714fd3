import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from mmmm_amd import kernels as K
from tools.bench_kernels import timeit
dev = torch.device('cuda:0')
for name, lens, H, hd, causal in [('vit', [785] * 8, 16, 112, False), ('lm', [456] * 8, 32, 128, True)]:
    rows = sum(lens)
    qkv = torch.randn(rows, 3 * H * hd, device=dev).bfloat16()
    q, k, v = qkv[:, :H * hd], qkv[:, H * hd:2 * H * hd], qkv[:, 2 * H * hd:]
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=dev)
    out, lse = K.attn_fwd(q, k, v, cu, max(lens), H, hd, hd ** -0.5, causal)
    dout = torch.randn_like(out)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    ms = timeit(lambda: K.attn_fwd(q, k, v, cu, max(lens), H, hd, hd ** -0.5, causal), iters=n, warm=1)
    msb = timeit(lambda: K.attn_bwd(q, k, v, out, lse, dout, cu, max(lens), H, hd, hd ** -0.5, causal), iters=n, warm=1)
    fl = sum(4 * l * l * hd * H for l in lens) * (0.5 if causal else 1)
    print(name, f'fwd {ms*1e3:.0f} us {fl/ms/1e9:.0f} TF | bwd {msb*1e3:.0f} us {2.5*fl/msb/1e9:.0f} TF', flush=True)
