"""Host wall time of a training step's parts, each step started from an idle GPU (no profiler), for variants of host-side mechanisms:
   python3 tools/host_wall.py [n_steps] [--no-hooks] [--set MODULE.NAME=VALUE ...]
--no-hooks removes the gradient-exchange readiness hooks after construction (MEASUREMENT ONLY: buckets are then launched by finish())."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from mmmm_amd.ddp import BucketedGradAllReduce
from mmmm_amd.optim import FlatAdamW
from mmmm_amd.models.lora import ActivationBudget

args = sys.argv[1:]
n = int(args[0]) if args and args[0].isdigit() else 6
sets = [a.split('=', 1) for a in args if '=' in a and not a.startswith('--')]
bench.apply_sets([f'{k}={v}' for k, v in sets])
dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
w = bench.WORKLOADS['phase-vg-448']
model, tok = bench.build(w, dev, 1.0)
trainable = [p for p in model.parameters() if p.requires_grad]
ddp = BucketedGradAllReduce(trainable, world_size=1)
opt = FlatAdamW(ddp, lr=5e-5, weight_decay=0.01, max_grad_norm=1.0)
batch = bench.make_batch(w, tok, 8, dev, seed=0)
ActivationBudget.limit = 160 << 30
if '--no-hooks' in args:
    for h in ddp._hooks:
        h.remove()


def step():
    ddp.zero_grad()
    loss = model.training_step(batch)
    loss.backward()
    ddp.finish()
    opt.step()


for _ in range(3):
    step()
rows = []
for _ in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ddp.zero_grad(); t1 = time.perf_counter()
    loss = model.training_step(batch); t2 = time.perf_counter()
    loss.backward(); t3 = time.perf_counter()
    ddp.finish(); opt.step(); t4 = time.perf_counter()
    rows.append([(b - a) * 1e3 for a, b in ((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t0, t4))])
torch.cuda.synchronize()
med = [sorted(r[i] for r in rows)[len(rows) // 2] for i in range(5)]
mn = [min(r[i] for r in rows) for i in range(5)]
print(' '.join(args), '| median ms: zero_grad %.1f forward %.1f backward %.1f finish+opt %.1f total %.1f | min total %.1f (forward %.1f backward %.1f)' % (*med, mn[4], mn[1], mn[2]))
