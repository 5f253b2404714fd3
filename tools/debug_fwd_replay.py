"""forward replay of one full-size step: which logged scalar differs between identical runs? usage: debug_fwd_replay.py <workload>"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from mmmm_amd.models.lora import ActivationBudget, StepState
dev = torch.device('cuda:0')
wl = sys.argv[1] if len(sys.argv) > 1 else 'model-hr-3d'
model, tok = bench.build(bench.WORKLOADS['phase-vg-448'], dev, 1.0)
batch = bench.make_batch(bench.WORKLOADS[wl], tok, 2, dev, seed=11)
ActivationBudget.limit = None
runs = []
for i in range(4):
    StepState.step = 7
    with torch.no_grad():
        loss = model.training_step(batch)
    torch.cuda.synchronize()
    runs.append({k: float(v) for k, v in model.logged.items() if torch.is_tensor(v)})
for k in runs[0]:
    vals = [r[k] for r in runs]
    print(f'{k:50s}', ' '.join(f'{v:.8f}' for v in vals), '' if max(vals) == min(vals) else '   <-- differs')
