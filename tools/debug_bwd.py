import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from tests.test_model_gpu import tiny_config, make_inputs, run_oracle
from tests._gpu_common import cpu, oracle_cfg, oracle_state, randomize_, rel
from mmmm_amd.models.mmmm import MMMMForCausalLM, VisionArgs
from mmmm_amd.models.lora import LoraConfig
from mmmm_amd.utils import apply_lora
from oracle import vividmed as O

dev = torch.device('cuda:0')
m = MMMMForCausalLM(tiny_config(), vision_override=VisionArgs(pos_embed_shape=(2, 2, 4), patch_size=(4, 8, 8)))
apply_lora(m, LoraConfig(r=64, lora_alpha=8, lora_dropout=0.0, use_rslora=True))
randomize_(m, 123)
m.to(dev).to(torch.bfloat16)
m.train()
batch, _ = make_inputs(dev, seed=5)

# ---- vision only
feats = m.model.vision(batch['image'], batch['patch_size'], batch['pool_size'])
g = torch.Generator().manual_seed(1)
ws = [torch.randn(f.shape, generator=g) for f in feats]
loss = sum((f.float() * w.to(dev)).sum() for f, w in zip(feats, ws))
loss.backward()
sd = {k: v.requires_grad_(v.is_floating_point()) for k, v in oracle_state(m).items()}
ref = O.vision_forward(sd, oracle_cfg(m.config), [x.float() for x in cpu(batch['image'])], batch['patch_size'], batch['pool_size'])
sum((f * w).sum() for f, w in zip(ref, ws)).backward()
print('VISION ONLY')
for name, p in m.named_parameters():
    if p.grad is not None and sd[name].grad is not None:
        print(f'{rel(p.grad.float(), sd[name].grad):8.4f} {name}')
