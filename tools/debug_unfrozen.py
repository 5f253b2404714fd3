import sys, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from mmmm_amd.ddp import BucketedGradAllReduce
dev = torch.device('cuda', 0)
w = bench.WORKLOADS['phase-vg-448']
model, tok = bench.build(w, dev, float(os.environ.get('DS', '0.07')))
if os.environ.get('FREEZE'):
    model.sam.requires_grad_(False); model.isam_model.requires_grad_(False)
trainable = [p for p in model.parameters() if p.requires_grad]
names = {id(p): n for n, p in model.named_parameters()}
ddp = BucketedGradAllReduce(trainable, world_size=1)
opt = torch.optim.AdamW(trainable, lr=5e-5, weight_decay=0.01, fused=True)
batch = bench.make_batch(w, tok, 8, dev, seed=0)
for step in range(int(os.environ.get("STEPS", "2"))):
    ddp.zero_grad()
    try:
        loss = model.training_step(batch)
    except Exception as e:
        print('training_step failed:', type(e).__name__, e); break
    loss.backward()
    ddp.finish()
    bad = [names[id(p)] for p in trainable if p.grad is not None and not torch.isfinite(p.grad).all()]
    if step == 0:
        top = sorted(((p.grad.float().norm().item(), names[id(p)], tuple(p.shape)) for p in trainable if p.grad is not None), reverse=True)[:14]
        for t in top: print('   ', t, flush=True)
    gn = ddp.clip_grad_norm_(1.0)
    print('step', step, 'loss', loss.item(), 'gnorm', gn.item(), 'nonfinite grads', len(bad), bad[:6], flush=True)
    opt.step()
    badp = [names[id(p)] for p in trainable if not torch.isfinite(p).all()]
    print('   nonfinite params', len(badp), badp[:6], flush=True)
