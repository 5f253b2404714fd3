"""cProfile of the HOST side of training steps (enqueue only): where does the Python time of a step go?
usage: python tools/host_profile.py [n_steps]"""
import cProfile
import pstats
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from mmmm_amd.ddp import BucketedGradAllReduce
from mmmm_amd.optim import FlatAdamW

dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
w = bench.WORKLOADS['phase-vg-448']
model, tok = bench.build(w, dev, 1.0)
trainable = [p for p in model.parameters() if p.requires_grad]
ddp = BucketedGradAllReduce(trainable, world_size=1)
opt = FlatAdamW(ddp, lr=5e-5, weight_decay=0.01, max_grad_norm=1.0)
batch = bench.make_batch(w, tok, 8, dev, seed=0)
from mmmm_amd.models.lora import ActivationBudget
ActivationBudget.limit = 160 << 30          # what bench.py's planning step arrives at: every layer kept, nothing recomputed


def step():
    ddp.zero_grad()
    loss = model.training_step(batch)
    loss.backward()
    ddp.finish()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3

# wall time of the host side per part, each from an idle GPU, no profiler
import time
parts = {'zero_grad': 0.0, 'forward': 0.0, 'backward': 0.0, 'finish+opt': 0.0}
cpu = {'forward': 0.0, 'backward': 0.0}
for _ in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ddp.zero_grad(); t1 = time.perf_counter(); c1 = time.process_time()
    loss = model.training_step(batch); t2 = time.perf_counter(); c2 = time.process_time()
    loss.backward(); t3 = time.perf_counter(); c3 = time.process_time()
    ddp.finish(); opt.step(); t4 = time.perf_counter()
    for k, v in zip(parts, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
        parts[k] += v * 1e3 / n
    cpu['forward'] += (c2 - c1) * 1e3 / n
    cpu['backward'] += (c3 - c2) * 1e3 / n
print('host wall per step (ms, no profiler):', {k: round(v, 1) for k, v in parts.items()}, 'sum', round(sum(parts.values()), 1))
print('process CPU time of the same parts (all threads; below the wall time = the host was WAITING, e.g. on a full launch queue):',
      {k: round(v, 1) for k, v in cpu.items()})

# the backward's Python runs on autograd's device thread: a second profiler is switched on from inside that thread (first custom
# backward it runs) — cProfile attaches to the calling thread only
import threading
import mmmm_amd.functional as Fh
pr_bwd = cProfile.Profile()
_main = threading.get_ident()
_armed = [False]
_orig = Fh._Linear.backward


def _arm(ctx, *a):
    if threading.get_ident() != _main and not _armed[0]:
        _armed[0] = True
        pr_bwd.enable()
    return _orig(ctx, *a)


Fh._Linear.backward = staticmethod(_arm)
pr = cProfile.Profile()
for _ in range(n):
    torch.cuda.synchronize()          # every profiled step starts on an idle GPU: a launch that finds the queue full WAITS inside hip.call, and
    pr.enable()                       # three steps enqueued back to back (host 150 ms, GPU 300 ms per step) measured that wait, not the host
    step()
    pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
st.print_callers('rsub|argsort|Event.synchronize')
if _armed[0]:
    print('==== autograd thread (backward) ====')
    sb = pstats.Stats(pr_bwd)
    sb.sort_stats('tottime').print_stats(40)
