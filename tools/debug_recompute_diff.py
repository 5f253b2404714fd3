"""keep-every-activation vs recompute-every-layer on the phase-vlm mixed batch: which gradients differ, by how much (bit-level)."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from tests.test_fullsize_gpu import _build, _batch_for, run_step
dev = torch.device('cuda', 0)
model, ddp, _ = _build(dev, 'phase-vlm-448')
batch = _batch_for(model, 'phase-vlm-mixed', dev)
keep = 200 << 30
l0, g0 = run_step(model, ddp, batch, 3, keep)
for what, budget in (('replay', keep), ('recompute', None), ('recompute again', None), ('keep again', keep)):
    l, g = run_step(model, ddp, batch, 3, budget)
    diff = [(n, ((g[n] - g0[n]).norm() / g0[n].norm().clamp_min(1e-30)).item()) for n in g0 if not torch.equal(g[n], g0[n])]
    lora = [d for d in diff if 'lora' in d[0]]
    print(what, 'loss equal', torch.equal(l, l0), 'differing', len(diff), 'of which lora', len(lora))
    for n, e in lora[:12]:
        print('   ', n, f'{e:.2e}')
    layers = sorted({n.split('.layers.')[1].split('.')[0] + ('v' if 'vision.' in n else 'd') for n, _ in lora if '.layers.' in n})
    print('    lora layers:', layers[:40])
