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
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
st.print_callers('rsub|argsort|Event.synchronize')
