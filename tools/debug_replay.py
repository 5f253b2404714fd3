"""which parameter gradients differ between two replays of the same full-size step? (determinism check)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from mmmm_amd.ddp import BucketedGradAllReduce
from mmmm_amd.models.lora import ActivationBudget, StepState

dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
import os
w = bench.WORKLOADS[os.environ.get('WORKLOAD', 'phase-vg-448')]      # tool-side knob
model, tok = bench.build(w, dev, float(sys.argv[1]) if len(sys.argv) > 1 else 1.0)
names = {id(p): n for n, p in model.named_parameters()}
trainable = [p for p in model.parameters() if p.requires_grad]
ddp = BucketedGradAllReduce(trainable, world_size=1)
batch = bench.make_batch(w, tok, 2, dev, seed=11)
ActivationBudget.limit = 200 << 30


def run(limit=200 << 30):
    ActivationBudget.limit = limit
    StepState.step = 7
    ddp.zero_grad()
    loss = model.training_step(batch)
    loss.backward()
    ddp.finish()
    torch.cuda.synchronize()
    return loss.item(), {names[id(p)]: p.grad.detach().float().clone() for p in trainable}


l0, g0 = run()
l1, g1 = run()
if 'nosidestream' in sys.argv:          # third run with every weight-gradient kernel on the main stream
    import mmmm_amd.functional as Fh
    Fh.WGRAD_SIDE_STREAM = False
    model.concurrent_heads = False
l2, g2 = run(None) if 'ckpt' in sys.argv else run()      # 'ckpt': third run recomputes every layer (reference mode)
print('loss', l0, l1, l2)
rows = []
for n in g0:
    d01 = (g0[n] - g1[n]).norm().item()
    d12 = (g1[n] - g2[n]).norm().item()
    nb = g0[n].norm().item()
    rows.append((max(d01, d12) / max(nb, 1e-30), n, nb, d01, d12))
rows.sort(reverse=True)
print('params', len(rows), 'with any difference', sum(1 for r in rows if r[0] > 0))
pat = sys.argv[2] if len(sys.argv) > 2 else ''
for key in ('lm_head.weight', 'model.norm.weight', 'vg_proj.0.weight', 'vg_proj.2.weight', 'model.embed_tokens.weight', 'model.layers.31.mlp.language_mlp.down_proj.lora_B.default.weight', 'model.layers.31.input_layernorm.weight', 'sam.mask_decoder.output_hypernetworks_mlps.0.layers.0.weight', 'isam_model.mask_decoder.transformer.layers.0.mlp.lin1.weight', 'sam.image_encoder.blocks.11.mlp.linear2.weight', 'sam.image_encoder.blocks.0.attn.qkv.weight', 'model.vision.linear_proj.linear_proj.lora_B.default.weight', 'model.vision.transformer.layers.62.mlp.fc2.lora_B.default.weight'):
    for r in rows:
        if r[1] == key:
            print(f'KEY {r[0]:.3e}  |g|={r[2]:.3e} d01={r[3]:.2e} d12={r[4]:.2e}  {r[1]}')
for r in [r for r in rows if pat in r[1]][:40]:
    print(f'{r[0]:.3e}  |g|={r[2]:.3e} d01={r[3]:.2e} d12={r[4]:.2e}  {r[1]}')

# anomaly scan: a parameter whose replay noise is far above that of its neighbours (same block) points at a race rather than
# at conditioning (the signature of the dy race: lora_B of dense / down_proj at O(1) while the rest of the layer sat at 1e-3)
import re, statistics
groups = {}
for r in rows:
    if r[2] < 1e-6:
        continue
    m = re.match(r'(.*?(?:layers|blocks)\.\d+)\.', r[1])
    groups.setdefault(m.group(1) if m else r[1].split('.')[0], []).append(r)
flagged = 0
for g, rs in groups.items():
    if len(rs) < 4:
        continue
    med = statistics.median(r[0] for r in rs)
    for r in rs:
        if r[0] > 20 * max(med, 1e-7):
            flagged += 1
            print(f'ANOMALY {r[1]}: {r[0]:.2e} vs block median {med:.2e}')
print('anomalies:', flagged)
