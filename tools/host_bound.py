"""Is the step host-bound? time until step() returns (all launches enqueued) vs time until the GPU is done."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from mmmm_amd.ddp import BucketedGradAllReduce
from mmmm_amd.models.lora import ActivationBudget
dev = torch.device('cuda', 0)
w = bench.WORKLOADS['phase-vg-448']
model, tok = bench.build(w, dev, 1.0)
trainable = [p for p in model.parameters() if p.requires_grad]
ddp = BucketedGradAllReduce(trainable, world_size=1)
opt = torch.optim.AdamW(trainable, lr=5e-5, weight_decay=0.01, fused=True)
batch = bench.make_batch(w, tok, 8, dev, seed=0)
ActivationBudget.limit = 1 << 40
def step(marks):
    ddp.zero_grad(); marks.append(time.perf_counter())
    loss = model.training_step(batch); marks.append(time.perf_counter())
    loss.backward(); marks.append(time.perf_counter())
    ddp.finish(); ddp.clip_grad_norm_(1.0); opt.step(); marks.append(time.perf_counter())
for _ in range(3): step([])
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); m = []
    step(m)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'host enqueue {1e3*(t1-t0):.1f} ms (zero {1e3*(m[0]-t0):.1f}, fwd {1e3*(m[1]-m[0]):.1f}, bwd {1e3*(m[2]-m[1]):.1f}, opt {1e3*(m[3]-m[2]):.1f}) | GPU done {1e3*(t2-t0):.1f} ms')
import cProfile, pstats, io
pr = cProfile.Profile()
ddp.zero_grad()
pr.enable()
loss = model.training_step(batch)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
print(s.getvalue()[:6000])
