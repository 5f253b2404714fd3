import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from tests.test_model_gpu import tiny_config, make_inputs
from tests._gpu_common import cpu, oracle_cfg, oracle_state, randomize_, rel
from mmmm_amd.models.mmmm import MMMMForCausalLM, VisionArgs
from mmmm_amd.models.lora import LoraConfig
from mmmm_amd.utils import apply_lora
from mmmm_amd import functional as Fh
from oracle import vividmed as O

dev = torch.device('cuda:0')
m = MMMMForCausalLM(tiny_config(), vision_override=VisionArgs(pos_embed_shape=(2, 2, 4), patch_size=(4, 8, 8)))
apply_lora(m, LoraConfig(r=64, lora_alpha=8, lora_dropout=0.0, use_rslora=True))
randomize_(m, 123)
m.to(dev).to(torch.bfloat16)
m.train()
batch, _ = make_inputs(dev, seed=5)

rec = []
orig = Fh.attention
def patched(qkv, *a, **k):
    out = orig(qkv, *a, **k)
    entry = {'qkv': qkv.detach().clone()}
    qkv.register_hook(lambda g, e=entry: e.__setitem__('dqkv', g.detach().clone()))
    out.register_hook(lambda g, e=entry: e.__setitem__('dout', g.detach().clone()))
    entry['out'] = out.detach().clone()
    rec.append(entry)
    return out
Fh.attention = patched
import mmmm_amd.models.cogvlm.visual as V
V.Fh.attention = patched

feats = m.model.vision(batch['image'], batch['patch_size'], batch['pool_size'])
g = torch.Generator().manual_seed(1)
ws = [torch.randn(f.shape, generator=g) for f in feats]
sum((f.float() * w.to(dev)).sum() for f, w in zip(feats, ws)).backward()

orec = []
oorig = O.vit_attention
def opatched(q, k, v, lens, scale):
    q.retain_grad(); k.retain_grad(); v.retain_grad()
    out = oorig(q, k, v, lens, scale)
    out.retain_grad()
    orec.append(dict(q=q, k=k, v=v, out=out))
    return out
O.vit_attention = opatched
sd = {k: v.requires_grad_(v.is_floating_point()) for k, v in oracle_state(m).items()}
ref = O.vision_forward(sd, oracle_cfg(m.config), [x.float() for x in cpu(batch['image'])], batch['patch_size'], batch['pool_size'])
sum((f * w).sum() for f, w in zip(ref, ws)).backward()
for li in range(2):
    e, o = rec[li], orec[li]
    T = e['qkv'].shape[0]
    qkv = e['qkv'].float().cpu().view(T, 3, 2, 64)
    d = e['dqkv'].float().cpu().view(T, 3, 2, 64)
    print('layer', li, 'q %.4f k %.4f v %.4f out %.4f dout %.4f | dq %.4f dk %.4f dv %.4f' % (
        rel(qkv[:, 0], o['q']), rel(qkv[:, 1], o['k']), rel(qkv[:, 2], o['v']), rel(e['out'].float().cpu().view(T, 2, 64), o['out']),
        rel(e['dout'].float().cpu().view(T, 2, 64), o['out'].grad),
        rel(d[:, 0], o['q'].grad), rel(d[:, 1], o['k'].grad), rel(d[:, 2], o['v'].grad)))
    if li == 1:
        for r in range(T):
            print('   row %2d dq %.3f dk %.3f dv %.3f |dout| %.3e ref|dq| %.3e' % (r, rel(d[r, 0], o['q'].grad[r]), rel(d[r, 1], o['k'].grad[r]), rel(d[r, 2], o['v'].grad[r]),
                  e['dout'][r].float().norm().item(), o['q'].grad[r].norm().item()))
