import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from mmmm_amd import kernels as K
from tests.test_kernels_gpu import _attn_ref, rel_err
dev = torch.device('cuda:0')
for hd, H, causal, lens in [(64, 2, False, [9, 17, 17]), (64, 2, True, [9, 17, 17]), (64, 2, False, [64]), (64, 2, False, [65]), (128, 2, False, [9, 17, 17]), (64, 2, False, [17]), (64, 2, False, [33, 40])]:
    cu = [0]
    for l in lens:
        cu.append(cu[-1] + l)
    rows = cu[-1]
    torch.manual_seed(0)
    qkv = torch.randn(rows, 3 * H * hd, device=dev).bfloat16()
    q, k, v = qkv[:, :H * hd], qkv[:, H * hd:2 * H * hd], qkv[:, 2 * H * hd:]
    cu_t = torch.tensor(cu, dtype=torch.int32, device=dev)
    scale = hd ** -0.5
    out, lse = K.attn_fwd(q, k, v, cu_t, max(lens), H, hd, scale, causal)
    qf, kf, vf = (t.float().reshape(rows, H, hd).clone().requires_grad_() for t in (q, k, v))
    ref = _attn_ref(qf, kf, vf, cu, scale, causal)
    dout = torch.randn(rows, H * hd, device=dev).bfloat16()
    ref.backward(dout.float().view(rows, H, hd))
    dqkv = K.attn_bwd(q, k, v, out, lse, dout, cu_t, max(lens), H, hd, scale, causal)
    print(hd, H, causal, lens, 'fwd %.4f dq %.4f dk %.4f dv %.4f' % (
        rel_err(out.view(rows, H, hd), ref), rel_err(dqkv[:, 0].view(rows, H, hd), qf.grad),
        rel_err(dqkv[:, 1].view(rows, H, hd), kf.grad), rel_err(dqkv[:, 2].view(rows, H, hd), vf.grad)))
    # per-sequence dk error
    for i in range(len(lens)):
        s, e = cu[i], cu[i + 1]
        print('   seq', i, 'dq %.4f dk %.4f dv %.4f' % (rel_err(dqkv[s:e, 0].view(-1, H, hd), qf.grad[s:e]),
              rel_err(dqkv[s:e, 1].view(-1, H, hd), kf.grad[s:e]), rel_err(dqkv[s:e, 2].view(-1, H, hd), vf.grad[s:e])))
