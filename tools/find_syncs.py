"""report every host<->device synchronisation of one training step (torch sync debug mode)"""
import sys, warnings
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from mmmm_amd.ddp import BucketedGradAllReduce
from mmmm_amd.models.lora import ActivationBudget
dev = torch.device('cuda', 0)
w = bench.WORKLOADS['phase-vg-448']
model, tok = bench.build(w, dev, 0.07)
trainable = [p for p in model.parameters() if p.requires_grad]
ddp = BucketedGradAllReduce(trainable, world_size=1)
opt = torch.optim.AdamW(trainable, lr=5e-5, weight_decay=0.01, fused=True)
batch = bench.make_batch(w, tok, 8, dev, seed=0)
ActivationBudget.limit = 1 << 40
def step():
    ddp.zero_grad(); loss = model.training_step(batch); loss.backward(); ddp.finish(); ddp.clip_grad_norm_(1.0); opt.step()
step(); step()
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode('warn')
with warnings.catch_warnings(record=True) as ws:
    warnings.simplefilter('always')
    step()
torch.cuda.set_sync_debug_mode('default')
import collections
c = collections.Counter()
for wn in ws:
    c[(wn.filename.replace(str(Path(__file__).resolve().parents[1]), ''), wn.lineno, str(wn.message)[:60])] += 1
for k, v in c.most_common(40): print(v, k)
# first synchronising op inside backward, with its traceback
import traceback
ddp.zero_grad()
loss = model.training_step(batch)
torch.cuda.set_sync_debug_mode('error')
try:
    with torch.autograd.detect_anomaly(check_nan=False):
        loss.backward()
except Exception as e:
    traceback.print_exc(limit=12)
torch.cuda.set_sync_debug_mode('default')
