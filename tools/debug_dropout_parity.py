import os, sys
sys.path.insert(0, '.')
import torch
from tests.test_model_gpu import tiny_config, make_inputs, run_oracle
from tests._gpu_common import lora_dropout_on, randomize_, rel
from mmmm_amd.models.mmmm import MMMMForCausalLM, VisionArgs
from mmmm_amd.models.lora import LoraConfig, StepState
from mmmm_amd.utils import apply_lora
dev = torch.device('cuda:0')
# DG=1: every GEMM whose LoRA extension carries a dropout mask (the input-gradient GEMMs) is checked against main product + masked extension
# built from vm_dropout's mask of the same seed
if os.environ.get('DG') == '1':
    from mmmm_amd import kernels as _Kg
    _gemm = _Kg.gemm

    def _gemm_checked(a, w, **kw):
        out = _gemm(a, w, **kw)
        if kw.get('drop_p', 0.0) > 0 and kw.get('a2') is not None and a.dtype == torch.bfloat16:
            kw2 = {k: v for k, v in kw.items() if k not in ('a2', 'b2', 'b2_1', 'alpha2', 'drop_p', 'drop_seed', 'residual', 'out')}
            main = _gemm(a, w, out_dtype=torch.float32, **{k: v for k, v in kw2.items() if k != 'out_dtype'}).float()
            M, N = out.shape
            mask = _Kg.dropout(torch.ones(M, N, device=a.device), kw['drop_p'], kw['drop_seed'])
            cnt = kw.get('counts')
            ext = kw['a2'].float() @ kw['b2'].float().T
            if kw.get('b2_1') is not None and cnt is not None:
                nv = int(cnt[0]); ext[nv:] = kw['a2'][nv:].float() @ kw['b2_1'].float().T
            ref = main + kw.get('alpha2', 1.0) * ext * mask
            if kw.get('residual') is not None:
                ref = ref.bfloat16().float() + kw['residual'].float()
            n = int(cnt[1]) if cnt is not None else M
            print(f'   dgrad GEMM [{M} x {N}] K {a.shape[1]} seed {kw["drop_seed"]}: rel err vs main + masked extension {rel(out[:n].float(), ref[:n]):.5f}')
        return out
    _Kg.gemm = _gemm_checked
# LD=1: every forward rank-64 projection t = drop(x) A^T of the HIP path is kept by seed, and when the oracle asks for that site's mask the
# oracle-side projection (x * mask / (1 - p)) A^T is compared with it (ViT-E / adapter sites: same row order on both sides)
_LD = {}
if os.environ.get('LD') == '1':
    from mmmm_amd import kernels as _Kl
    from tests import _gpu_common as _G
    _ld = _Kl.lora_down

    def _ld_rec(x, A0, A1=None, **kw):
        t = _ld(x, A0, A1, **kw)
        if kw.get('drop_p', 0.0) > 0 and A1 is None:
            _LD[kw['drop_seed']] = (x.detach().float().cpu(), A0.detach().float().cpu(), t.detach().float().cpu())
        return t
    _Kl.lora_down = _ld_rec
    _call = _G.LoraMasks.__call__

    def _call_chk(self, name, x):
        m = _call(self, name, x)
        if m is not None and not name.startswith('model.layers.'):
            from mmmm_amd.models.lora import StepState as _S
            rec = _LD.get(_S.seed_for(self.sites[name]._site))
            if rec is not None and x.dtype == torch.float32 and rec[0].shape == x.reshape(-1, x.shape[-1]).shape:
                xo = x.reshape(-1, x.shape[-1]).detach()
                to = (xo * m.reshape(xo.shape).float() / (1 - self.p)) @ rec[1].T
                th_same_x = (rec[0] * m.reshape(xo.shape).float() / (1 - self.p)) @ rec[1].T
                print(f'   {name}: oracle t vs HIP t {rel(rec[2], to):.4f}; HIP x through the ORACLE mask vs HIP t {rel(rec[2], th_same_x):.4f}; x itself {rel(rec[0], xo):.4f}')
        return m
    _G.LoraMasks.__call__ = _call_chk
# GR=1: the ViT-E attention's output gradient and q / k / v gradients of the HIP path next to the fp32 oracle's, layer by layer
_GR_H, _GR_O = [], []
if os.environ.get('GR') == '1':
    from mmmm_amd import kernels as _Kr
    from oracle import vividmed as _O
    _ab = _Kr.attn_bwd

    def _ab_rec(q, k, v, out, lse, dout, cu, max_seqlen, H, hd, scale, causal, row_of_pos=None, total_pos_max=None):
        res = _ab(q, k, v, out, lse, dout, cu, max_seqlen, H, hd, scale, causal, row_of_pos, total_pos_max)
        if row_of_pos is None:
            _GR_H.append((dout.float().cpu().reshape(-1, H, hd), res.float().cpu()))
        return res
    _Kr.attn_bwd = _ab_rec
    _va = _O.vit_attention

    def _va_rec(q, k, v, lens, scale):
        q, k, v = (t.clone() for t in (q, k, v))
        for t in (q, k, v):
            t.retain_grad()
        o = _va(q, k, v, lens, scale)
        o.retain_grad()
        _GR_O.append((q, k, v, o))
        return o
    _O.vit_attention = _va_rec
# ATTN=1: every attention backward is checked on the spot against fp32 torch autograd on the SAME bf16 inputs (is a large gradient error
# downstream of an attention layer the kernel's, or the conditioning of that layer's softmax on this input?)
if os.environ.get('ATTN') == '1':
    from mmmm_amd import kernels as _K
    _orig = _K.attn_bwd

    def _checked(q, k, v, out, lse, dout, cu, max_seqlen, H, hd, scale, causal, row_of_pos=None, total_pos_max=None):
        res = _orig(q, k, v, out, lse, dout, cu, max_seqlen, H, hd, scale, causal, row_of_pos, total_pos_max)
        if row_of_pos is None:
            cuh = cu.tolist()
            with torch.enable_grad():
                qf, kf, vf = (t.float().reshape(-1, H, hd).detach().requires_grad_(True) for t in (q, k, v))
                outs, pmax = [], []
                for i in range(len(cuh) - 1):
                    s_, e_ = cuh[i], cuh[i + 1]
                    sc = (qf[s_:e_].transpose(0, 1) @ kf[s_:e_].transpose(0, 1).transpose(1, 2)) * scale
                    pr = sc.softmax(-1)
                    pmax.append(pr.max(-1).values.mean().item())
                    outs.append((pr @ vf[s_:e_].transpose(0, 1)).transpose(0, 1))
                o = torch.cat(outs, 0)
                o.backward(dout.float().reshape(-1, H, hd))
            n = H * hd
            e = [rel(res[:, i].reshape(-1, H, hd), g) for i, g in enumerate((qf.grad, kf.grad, vf.grad))]
            # the same gradient with O rounded to bf16 in delta = rowsum(dO * O), as every flash-style backward computes it
            print(f'   attention backward rows {q.shape[0]} H {H} hd {hd}: dq {e[0]:.4f} dk {e[1]:.4f} dv {e[2]:.4f} vs fp32 autograd; mean max-probability {sum(pmax)/len(pmax):.3f}')
        return res
    _K.attn_bwd = _checked
m = MMMMForCausalLM(tiny_config(), vision_override=VisionArgs(pos_embed_shape=(2, 2, 4), patch_size=(4, 8, 8)))
apply_lora(m, LoraConfig(r=64, lora_alpha=8, lora_dropout=0.0, use_rslora=True))
randomize_(m, 123)
m.to(dev).to(torch.bfloat16)
qks = float(os.environ.get('QKS', '1'))
if qks != 1:
    # softer attention: scale the q and k rows of every fused qkv weight (the v rows stay)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith('query_key_value.weight'):
                p[: 2 * p.shape[0] // 3] *= qks
m.train()
ck = int(os.environ.get('CK', '1'))
if ck: m.gradient_checkpointing_enable()
batch, _ = make_inputs(dev, seed=11)
for seed in (int(s) for s in sys.argv[1:]):
    StepState.seed = seed
    p_ = float(os.environ.get('P', '0.05'))
    import contextlib
    with (lora_dropout_on(m, batch['vlm_inputs'], p_) if p_ > 0 else contextlib.nullcontext()) as masks:
        for p in m.parameters(): p.grad = None
        out = m(**batch['vlm_inputs'], image=batch['image'], patch_size=batch['patch_size'], pool_size=batch['pool_size'])
        out.loss.backward()
        if masks: masks.begin()
        ref, sd = run_oracle(m, batch, need_grad=True); ref.loss.backward()
        if masks: masks.begin()
        ref16, sd16 = run_oracle(m, batch, need_grad=True, dtype=torch.bfloat16); ref16.loss.backward()
    if os.environ.get('SENS') == '1':
        # conditioning of the gradient itself: the fp32 oracle once more on an image perturbed by one bf16 ulp (x * (1 + 2^-7)), same masks
        b2 = dict(batch, image=[(x.float() * (1 + 2.0 ** -7)).to(x.dtype) for x in batch['image']])
        with (lora_dropout_on(m, batch['vlm_inputs'], p_) if p_ > 0 else contextlib.nullcontext()) as masks2:
            if masks2: masks2.begin()
            refp, sdp = run_oracle(m, b2, need_grad=True); refp.loss.backward()
        sens = sorted(((rel(sdp[n].grad, sd[n].grad), n) for n, _ in m.named_parameters() if n in sd and sd[n].grad is not None and sd[n].grad.norm() > 0), reverse=True)
        print('   fp32 oracle, image perturbed by 2^-7 relative (one bf16 ulp): gradient change per tensor (top 5):', [(round(a, 4), b.split('model.')[-1]) for a, b in sens[:5]])
    if _GR_H and _GR_O:
        nl = len(_GR_H)
        fp32_calls = _GR_O[:nl]                      # the fp32 oracle ran first (layer 0 .. L-1); the HIP backward recorded L-1 .. 0
        for li in range(nl):
            dout_h, dqkv_h = _GR_H[nl - 1 - li]
            q, k, v, o = fp32_calls[li]
            print(f'   ViT layer {li}: d(attention output) HIP vs fp32 oracle {rel(dout_h, o.grad):.4f}; dq {rel(dqkv_h[:, 0].reshape(q.shape), q.grad):.4f} dk {rel(dqkv_h[:, 1].reshape(q.shape), k.grad):.4f} dv {rel(dqkv_h[:, 2].reshape(q.shape), v.grad):.4f}')
        _GR_H.clear(); _GR_O.clear()
    rows = []
    for name, p in m.named_parameters():
        if not p.requires_grad or sd[name].grad is None or sd[name].grad.norm() == 0: continue
        e_ref = rel(sd16[name].grad.float(), sd[name].grad); e_hip = rel(p.grad.float(), sd[name].grad)
        rows.append((e_hip / max(e_ref, 1e-9), name, e_ref, e_hip))
    rows.sort(reverse=True)
    print('seed', seed, 'p', p_, 'ck', ck, 'loss', out.loss.item(), ref.loss.item(), ref16.loss.item())
    for r in rows[:8]: print('   %.2f  %s  e_ref %.4f e_hip %.4f' % r)
    import statistics
    print('   median ratio %.2f over %d tensors' % (statistics.median(r[0] for r in rows), len(rows)))
