import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from tests._gpu_common import randomize_, rel
from tests.test_model_gpu import run_oracle
from mmmm_amd.models.cogvlm.configuration_cogvlm import CogVLMConfig
from mmmm_amd.models.mmmm import MMMMForCausalLM, VisionArgs
from mmmm_amd.models.lora import LoraConfig
from mmmm_amd.utils import apply_lora
from mmmm_amd.data.synthetic import SpecialTokens, make_batch
import os
dev = torch.device('cuda:0')
cfg = CogVLMConfig(vocab_size=2056, hidden_size=4096, intermediate_size=11008, num_hidden_layers=1, num_attention_heads=32,
                   vision_config=dict(in_channels=3, hidden_size=1792, num_heads=16, num_hidden_layers=1, intermediate_size=15360,
                                      layer_norm_eps=1e-6, patch_size=(16, 16, 16), pos_embed_shape=(2, 20, 20)))
m = MMMMForCausalLM(cfg, vision_override=VisionArgs(pos_embed_shape=(2, 20, 20), patch_size=16))
apply_lora(m, LoraConfig(r=64, lora_alpha=8, lora_dropout=0.0, use_rslora=True))
randomize_(m, 321)
with torch.no_grad():
    for n_, p_ in m.named_parameters():
        if 'vision' in n_ and 'query_key_value.weight' in n_ and 'lora' not in n_:
            p_.mul_(float(os.environ.get('QKV_MUL', '1')))
m.to(dev).to(torch.bfloat16).train()
tok = SpecialTokens(base_vocab=2048)
batch = make_batch([(3, 1, 320, 320), (3, 1, 320, 320)], [(1, 16, 16)] * 2, [(1, 2, 2)] * 2, [96, 61], tok=tok, seed=3, device=dev)
from mmmm_amd import kernels as K
from tests.test_kernels_gpu import _attn_ref
_orig = K.attn_bwd
def _chk(q, k, v, out, lse, dout, cu, max_seqlen, H, hd, scale, causal, row_of_pos=None, total_pos_max=None):
    r = _orig(q, k, v, out, lse, dout, cu, max_seqlen, H, hd, scale, causal, row_of_pos, total_pos_max)
    if hd == 112:
        rows = q.shape[0]
        with torch.enable_grad():
            qf, kf, vf = (t.detach().float().reshape(rows, H, hd).clone().requires_grad_() for t in (q, k, v))
            ref = _attn_ref(qf, kf, vf, cu.tolist(), scale, causal)
            ref.backward(dout.float().view(rows, H, hd))
        e = lambda a, b: ((a.float().reshape(b.shape) - b).norm() / b.norm()).item()
        print('IN-MODEL attn bwd check: out', e(out, ref), 'dq', e(r[:, 0], qf.grad), 'dk', e(r[:, 1], kf.grad), 'dv', e(r[:, 2], vf.grad),
              '| dout contiguous', dout.is_contiguous(), 'q stride', q.stride(), 'max', max_seqlen, 'cu', cu.tolist(), flush=True)
    return r
K.attn_bwd = _chk
cap = {}
lyr = m.model.vision.transformer.layers[0]
def _fh(mod, inp, outp):
    cap['x'] = inp[0].detach().clone(); cap['cu'] = inp[1]; cap['max'] = inp[2]
    outp.register_hook(lambda g_: cap.__setitem__('dy', g_.detach().clone()))
lyr.register_forward_hook(_fh)
out = m(**batch['vlm_inputs'], image=batch['image'], patch_size=batch['patch_size'], pool_size=batch['pool_size'], output_hidden_states=True)
out.loss.backward()
ref, sd = run_oracle(m, batch, need_grad=True)
ref.loss.backward()
print('loss', out.loss.item(), ref.loss.item())
errs = []
for name, p in m.named_parameters():
    if not p.requires_grad or p.grad is None: continue
    g = sd[name].grad
    if g is None or g.norm() == 0: continue
    errs.append((rel(p.grad.float(), g), name, g.norm().item()))
for e in sorted(errs, reverse=True)[:25]: print(f'{e[0]:.4f} {e[1]} refnorm {e[2]:.3e}')
print('n', len(errs), 'median', sorted(e[0] for e in errs)[len(errs)//2])
n = 'model.vision.transformer.layers.0.attention.query_key_value.lora_B.default.weight'
g, r = dict(m.named_parameters())[n].grad.float().cpu(), sd[n].grad
for nm, sl in (('q', slice(0, 1792)), ('k', slice(1792, 3584)), ('v', slice(3584, 5376))):
    print(nm, 'rel', ((g[sl] - r[sl]).norm() / r[sl].norm()).item(), 'refnorm', r[sl].norm().item(), 'gotnorm', g[sl].norm().item())

# replay the captured layer input / output gradient through the oracle layer alone
from oracle import vividmed as O
lsd = {('L.' + k): v.detach().float().cpu().requires_grad_() for k, v in lyr.state_dict().items()}
ocfg = O.Cfg(vocab_size=8, hidden_size=8, intermediate_size=8, num_hidden_layers=1, num_attention_heads=1,
             vision=O.VisionCfg(hidden_size=1792, num_heads=16, num_hidden_layers=1, intermediate_size=15360, layer_norm_eps=1e-6))
xr = cap['x'].float().cpu().requires_grad_()
lens = (cap['cu'][1:] - cap['cu'][:-1]).tolist()
yr = O.vit_layer(lsd, ocfg, 'L', xr, lens)
yr.backward(cap['dy'].float().cpu())
print('REPLAY with the product-side x, dy: |x| stats', cap['x'].float().std().item(), cap['x'].float().abs().max().item(), '|dy|', cap['dy'].float().norm().item())
for n_, p_ in lyr.named_parameters():
    if p_.grad is not None and lsd['L.' + n_].grad is not None:
        print(f'   {rel(p_.grad.float(), lsd["L." + n_].grad):.4f} {n_}')
