"""full-step replay (forward + backward + bucket finish, no optimizer) of one full-size workload: which logged scalar differs between
identical runs? usage: debug_step_replay.py <workload> [budget_gb|none]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from mmmm_amd.ddp import BucketedGradAllReduce
from mmmm_amd.models.lora import ActivationBudget, StepState
dev = torch.device('cuda:0')
wl = sys.argv[1] if len(sys.argv) > 1 else 'model-hr-3d'
budget = None if len(sys.argv) > 2 and sys.argv[2] == 'none' else (int(sys.argv[2]) if len(sys.argv) > 2 else 120) << 30
model, tok = bench.build(bench.WORKLOADS['phase-vg-448'], dev, 1.0)
ddp = BucketedGradAllReduce([p for p in model.parameters() if p.requires_grad], world_size=1)
batch = bench.make_batch(bench.WORKLOADS[wl], tok, 2, dev, seed=11)
runs = []
for i in range(4):
    ActivationBudget.limit = budget
    StepState.step = 7
    ddp.zero_grad()
    loss = model.training_step(batch)
    loss.backward()
    ddp.finish()
    torch.cuda.synchronize()
    runs.append({k: float(v) for k, v in model.logged.items() if torch.is_tensor(v)})
    print('plan', ActivationBudget.last_plan, 'reserved GB', torch.cuda.memory_reserved() >> 30)
for k in runs[0]:
    vals = [r[k] for r in runs]
    print(f'{k:50s}', ' '.join(f'{v:.8f}' for v in vals), '' if max(vals) == min(vals) else '   <-- differs')
