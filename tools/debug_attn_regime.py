import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from mmmm_amd import kernels as K
from tests.test_kernels_gpu import _attn_ref, rel_err
dev = torch.device('cuda:0')
torch.manual_seed(0)
H, hd, lens = 16, 112, [401, 401]
rows = sum(lens); cu = [0, 401, 802]
for qs in (1.0, 0.35, 0.1):
    qkv = torch.randn(rows, 3, H, hd, device=dev)
    qkv[:, :2] *= qs
    qkv = qkv.bfloat16()
    q, k, v = (qkv[:, i].reshape(rows, H * hd) for i in range(3))
    qkv2 = qkv.view(rows, 3 * H * hd)
    qv, kv, vv = qkv2[:, :H * hd], qkv2[:, H * hd:2 * H * hd], qkv2[:, 2 * H * hd:]
    cu_t = torch.tensor(cu, dtype=torch.int32, device=dev)
    out, lse = K.attn_fwd(qv, kv, vv, cu_t, 401, H, hd, hd ** -0.5, False)
    qf, kf, vf = (t.float().view(rows, H, hd).clone().requires_grad_() for t in (q, k, v))
    ref = _attn_ref(qf, kf, vf, cu, hd ** -0.5, False)
    dout = torch.randn(rows, H * hd, device=dev).bfloat16()
    ref.backward(dout.float().view(rows, H, hd))
    # reference evaluated with the bf16-ROUNDED output (what any flash backward sees: delta = rowsum(dO * O_bf16))
    dqkv = K.attn_bwd(qv, kv, vv, out, lse, dout, cu_t, 401, H, hd, hd ** -0.5, False)
    print(f'q/k scale {qs}: out {rel_err(out.view(rows, H, hd), ref):.4f} dq {rel_err(dqkv[:, 0].view(rows, H, hd), qf.grad):.4f} '
          f'dk {rel_err(dqkv[:, 1].view(rows, H, hd), kf.grad):.4f} dv {rel_err(dqkv[:, 2].view(rows, H, hd), vf.grad):.4f} '
          f'|dq| {qf.grad.norm().item():.3e}', flush=True)
