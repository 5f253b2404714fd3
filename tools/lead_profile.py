"""Where in the step does the GPU wait for the host? Every n-th native launch of one step records (host clock, HIP event); after the
step, lead = (GPU time at the event) - (host time at the enqueue), both relative to the step's start (the GPU is idle at the start, so
lead(0) ~ 0). A region where the lead is ~0 is host-bound: the GPU runs each launch the moment it is enqueued. No profiler attached.
usage: python tools/lead_profile.py [stride]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import bench  # noqa: E402
from mmmm_amd import hip  # noqa: E402
from mmmm_amd.ddp import BucketedGradAllReduce  # noqa: E402
from mmmm_amd.models.lora import ActivationBudget  # noqa: E402
from mmmm_amd.optim import FlatAdamW  # noqa: E402

stride = int(sys.argv[1]) if len(sys.argv) > 1 else 25
dev = torch.device('cuda', 0)
w = bench.WORKLOADS['phase-vg-448']
model, tok = bench.build(w, dev, 1.0)
trainable = [p for p in model.parameters() if p.requires_grad]
ddp = BucketedGradAllReduce(trainable, world_size=1)
opt = FlatAdamW(ddp, lr=5e-5, weight_decay=0.01, max_grad_norm=1.0)
batch = bench.make_batch(w, tok, 8, dev, seed=0)
ActivationBudget.limit = 160 << 30

marks, count, phase = [], [0], ['']
main = torch.cuda.current_stream(dev)
_call = hip.call


def call(name, *args):
    rc = _call(name, *args)
    count[0] += 1
    if marks is not None and count[0] % stride == 0 and torch.cuda.current_stream(dev) == main:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(main)
        marks.append((time.perf_counter(), ev, name, phase[0], count[0]))
    return rc


def step():
    phase[0] = 'zero'
    ddp.zero_grad()
    phase[0] = 'forward'
    loss = model.training_step(batch)
    phase[0] = 'backward'
    loss.backward()
    phase[0] = 'finish'
    ddp.finish()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
hip.call = call
import mmmm_amd.kernels as K  # noqa: E402
K.hip.call = call
for rep in range(2):
    marks.clear()
    count[0] = 0
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(main)
    step()
    step()                     # two steps back to back: does the host's lead survive the step boundary?
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'rep {rep}: 2 steps: host enqueue {1e3 * (t1 - t0):.1f} ms, GPU done {1e3 * (t2 - t0):.1f} ms, {count[0]} native launches')
rows = [(1e3 * (t - t0), e0.elapsed_time(ev), name, ph, n) for t, ev, name, ph, n in marks]
print('launch#  phase      host ms   gpu ms   lead ms  kernel')
prev = None
starved = 0.0
for h, g, name, ph, n in rows:
    lead = g - h
    print(f'{n:7d}  {ph:9s} {h:8.1f} {g:8.1f} {lead:8.2f}  {name}')
    if prev is not None and lead < 0.3 and prev[1] < 0.3:
        starved += h - prev[0]
    prev = (h, lead)
print(f'host time between consecutive samples that both had < 0.3 ms of lead (GPU starved): {starved:.1f} ms')
