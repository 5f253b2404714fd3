"""Distribution of the ViT-E MLP pre-activations in the benchmark's random-init model: share of values outside the GELU table."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from mmmm_amd import kernels as K
dev = torch.device('cuda:0')
w = bench.WORKLOADS['phase-vg-448']
model, tok = bench.build(w, dev, 0.25)
batch = bench.make_batch(w, tok, 8, dev, 0)
seen = []
orig = K.gemm
def spy(*a, **kw):
    out = orig(*a, **kw)
    o = out[1] if isinstance(out, tuple) else out
    if o.dim() == 2 and o.shape[1] == 15360 and o.dtype == torch.bfloat16:
        f = o.float().abs()
        seen.append((len(seen), float((f < 2 ** -16).float().mean()), float((f == 0).float().mean()), float((f >= 16).float().mean()), float(f.mean())))
    return out
K.gemm = spy
loss = model.training_step(batch, 0)
loss.backward()
torch.cuda.synchronize()
for s in seen[:6] + seen[-4:]:
    print('gemm %d: tiny %.5f zero %.5f big %.5f mean|h| %.4g' % s)
