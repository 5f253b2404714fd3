"""do the operands of the grouped factor-gradient launches differ between keep-all and recompute-all? (bit checksums per item)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from tests.test_fullsize_gpu import _build, _batch_for, run_step
from mmmm_amd import kernels as K
dev = torch.device('cuda', 0)
model, ddp, _ = _build(dev, 'phase-vlm-448')
batch = _batch_for(model, 'phase-vlm-mixed', dev)
names = {p.grad.data_ptr(): n for n, p in model.named_parameters() if p.requires_grad}
orig = K.tn_skinny_group
rec = {}
cur = [None]
def cs(t):
    return int(t.contiguous().view(torch.int16).to(torch.int64).sum().item())
def wrapped(items):
    for it in items:
        W, S, out = it[0], it[1], it[2]
        rec[cur[0]].setdefault(names[out.data_ptr()], []).append((tuple(W.shape), W.stride(), W.data_ptr() % 256, cs(W), cs(S), cs(out), it[3], it[5], it[6], it[7]))
    orig(items)
    for it in items:
        rec[cur[0]][names[it[2].data_ptr()]].append(cs(it[2]))
K.tn_skinny_group = wrapped
import mmmm_amd.functional as Fh
for mode, budget in (('keep', 200 << 30), ('recompute', None)):
    cur[0] = mode
    rec[mode] = {}
    run_step(model, ddp, batch, 3, budget)
n = 0
for k in rec['keep']:
    a, b = rec['keep'][k], rec['recompute'].get(k)
    if a != b:
        n += 1
        if n <= 6:
            print(k, '\n  keep     ', a, '\n  recompute', b)
print('items', len(rec['keep']), 'differing records', n)
