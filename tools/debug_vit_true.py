import sys, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from argparse import Namespace
from tests._gpu_common import randomize_, rel
from oracle import vividmed as O
from mmmm_amd.models.cogvlm.visual import TransformerLayer
from mmmm_amd.models.lora import LoraConfig, Linear
dev = torch.device('cuda:0')
d, H, f = 1792, 16, 15360
cfgn = Namespace(hidden_size=d, num_heads=H, intermediate_size=f, layer_norm_eps=1e-6)
layer = TransformerLayer(cfgn)
for mod in layer.modules():
    if isinstance(mod, Linear): mod.add_lora(LoraConfig(r=64, lora_alpha=8, lora_dropout=0.0, use_rslora=True))
randomize_(layer, 5)
mul = float(os.environ.get('QKV_MUL', '1'))
with torch.no_grad():
    layer.attention.query_key_value.weight.mul_(mul)
layer.to(dev).to(torch.bfloat16).train()
lens = [401, 401]
T = sum(lens)
g = torch.Generator().manual_seed(0)
x0 = torch.randn(T, d, generator=g)
dy0 = torch.randn(T, d, generator=g)
x = x0.to(dev).bfloat16().requires_grad_()
cu = torch.tensor([0, 401, 802], dtype=torch.int32, device=dev)
y = layer(x, cu, 401)
y.backward(dy0.to(dev).bfloat16())
sd = {('L.' + k): v.detach().float().cpu().requires_grad_() for k, v in layer.state_dict().items()}
ocfg = O.Cfg(vocab_size=8, hidden_size=8, intermediate_size=8, num_hidden_layers=1, num_attention_heads=1,
             vision=O.VisionCfg(hidden_size=d, num_heads=H, num_hidden_layers=1, intermediate_size=f, layer_norm_eps=1e-6))
xr = x0.bfloat16().float().requires_grad_()
yr = O.vit_layer(sd, ocfg, 'L', xr, lens)
yr.backward(dy0.bfloat16().float())
print('y', rel(y.float(), yr), 'dx', rel(x.grad.float(), xr.grad))
for n, p in layer.named_parameters():
    if p.grad is not None and sd['L.' + n].grad is not None:
        print(f'{rel(p.grad.float(), sd["L." + n].grad):.4f} {n}')
