import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import torch.nn.functional as F
from tests.test_model_gpu import tiny_config
from tests._gpu_common import oracle_cfg, oracle_state, randomize_, rel
from mmmm_amd.models.mmmm import MMMMForCausalLM, VisionArgs
from mmmm_amd.models.lora import LoraConfig, Linear
from mmmm_amd.utils import apply_lora
from mmmm_amd import functional as Fh, kernels as K
from oracle import vividmed as O

dev = torch.device('cuda:0')
torch.manual_seed(0)
# 1) plain LoRA linear, qkv shape
for (M, N, Kd, bias) in [(43, 384, 128, True), (43, 128, 128, True), (64, 384, 128, True), (43, 384, 128, False)]:
    lin = Linear(Kd, N, bias=bias)
    lin.add_lora(LoraConfig(lora_dropout=0.0))
    randomize_(lin, 1)
    lin.to(dev).to(torch.bfloat16)
    x = torch.randn(M, Kd, device=dev).bfloat16().requires_grad_()
    y = lin(x)
    dy = torch.randn(M, N, device=dev).bfloat16()
    y.backward(dy)
    xf = x.detach().float().clone().requires_grad_()
    W, A, B = lin.weight.float(), lin.A.detach().float().clone().requires_grad_(), lin.B.detach().float().clone().requires_grad_()
    yr = F.linear(xf, W, lin.bias.float() if bias else None) + F.linear(F.linear(xf, A), B)
    yr.backward(dy.float())
    print('linear', M, N, Kd, bias, 'y %.4f dx %.4f dA %.4f dB %.4f' % (rel(y.float(), yr), rel(x.grad.float(), xf.grad), rel(lin.A.grad.float(), A.grad), rel(lin.B.grad.float(), B.grad)))

# 2) ViT layer
m = MMMMForCausalLM(tiny_config(), vision_override=VisionArgs(pos_embed_shape=(2, 2, 4), patch_size=(4, 8, 8)))
apply_lora(m, LoraConfig(r=64, lora_alpha=8, lora_dropout=0.0, use_rslora=True))
randomize_(m, 123)
m.to(dev).to(torch.bfloat16)
m.train()
layer = m.model.vision.transformer.layers[1]
lens = [9, 17, 17]
cu = torch.tensor([0, 9, 26, 43], dtype=torch.int32, device=dev)
x = torch.randn(43, 128, device=dev).bfloat16().requires_grad_()
y = layer(x, cu, 17)
dy = torch.randn(43, 128, device=dev).bfloat16()
y.backward(dy)
sd = {k: v.requires_grad_(v.is_floating_point()) for k, v in oracle_state(m).items()}
xf = x.detach().float().cpu().requires_grad_()
yr = O.vit_layer(sd, oracle_cfg(m.config), 'model.vision.transformer.layers.1', xf, lens)
yr.backward(dy.float().cpu())
print('layer y %.4f dx %.4f' % (rel(y.float(), yr), rel(x.grad.float(), xf.grad)))
for name, p in layer.named_parameters():
    full = 'model.vision.transformer.layers.1.' + name
    if p.grad is not None:
        print('  %.4f %s' % (rel(p.grad.float(), sd[full].grad), name))
