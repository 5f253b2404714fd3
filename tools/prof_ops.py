"""which Python call sites launch the small ATen kernels (mul / add / fill / copy)? one training step under torch.profiler"""
import sys, collections
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from torch.profiler import profile, ProfilerActivity
from mmmm_amd.ddp import BucketedGradAllReduce

dev = torch.device('cuda:0')
w = bench.WORKLOADS['phase-vg-448']
model, tok = bench.build(w, dev, depth_scale=0.07)
trainable = [p for p in model.parameters() if p.requires_grad]
ddp = BucketedGradAllReduce(trainable, world_size=1)
opt = torch.optim.AdamW(trainable, lr=5e-5, weight_decay=0.01, fused=True)
batch = bench.make_batch(w, tok, 8, dev, seed=0)
from mmmm_amd.models.lora import ActivationBudget
ActivationBudget.limit = 160 << 30      # as bench.py after its planning step: every layer kept

def step():
    ddp.zero_grad()
    loss = model.training_step(batch)
    loss.backward()
    ddp.finish()
    ddp.clip_grad_norm_(1.0)
    opt.step()

step(); step()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    step()
cnt = collections.Counter()
for e in prof.events():
    if len(sys.argv) > 1 and e.name != sys.argv[1]:
        continue
    if e.name in ('aten::mul', 'aten::add', 'aten::add_', 'aten::fill_', 'aten::copy_', 'aten::mul_', 'aten::zeros', 'aten::cat', 'aten::clone', 'aten::index', 'aten::sum'):
        cnt[(e.name, str(e.input_shapes)[:100])] += 1
for (n, s), c in cnt.most_common(60):
    print(f'{c:6d} {n:12s} {s}')
