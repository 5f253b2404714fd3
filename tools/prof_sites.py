"""Which Python call sites launch the FORWARD's ATen ops, and which autograd nodes the backward's? One training step at a reduced
decoder / ViT depth (the grounding heads and losses are full size): a TorchDispatchMode records the innermost mmmm_amd frame of
every aten op of the forward; torch.profiler names the autograd node of every op of the backward that launched a kernel.
usage: python tools/prof_sites.py [depth_scale]"""
import collections
import sys
import traceback
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

import bench  # noqa: E402
from mmmm_amd.ddp import BucketedGradAllReduce  # noqa: E402
from mmmm_amd.models.lora import ActivationBudget  # noqa: E402
from mmmm_amd.optim import FlatAdamW  # noqa: E402

dev = torch.device('cuda:0')
w = bench.WORKLOADS['phase-vg-448']
model, tok = bench.build(w, dev, depth_scale=float(sys.argv[1]) if len(sys.argv) > 1 else 0.07)
trainable = [p for p in model.parameters() if p.requires_grad]
ddp = BucketedGradAllReduce(trainable, world_size=1)
opt = FlatAdamW(ddp, lr=5e-5, weight_decay=0.01, max_grad_norm=1.0)
batch = bench.make_batch(w, tok, 8, dev, seed=0)
ActivationBudget.limit = 160 << 30
VIEWS = ('view', 'reshape', 'expand', 'permute', 'transpose', 't.default', 'slice', 'select', 'unsqueeze', 'squeeze', 'alias', 'detach',
         'as_strided', 'unbind', 'split', '_unsafe_view', 'sym_', 'size', 'stride', 'is_', 'empty', 'lift_fresh', 'unflatten', 'chunk', 'narrow')


class Sites(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.count = collections.Counter()
        self.views = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace('aten.', '')
        # slices / selects of tensors that require grad: each becomes a zeros + copy (+ add) in the backward (SliceBackward0 / SelectBackward0)
        if name.startswith(('slice', 'select', 'narrow', 'index.Tensor', 'unbind', 'split')) and args and isinstance(args[0], torch.Tensor) and args[0].requires_grad:
            site = '?'
            for fr in reversed(traceback.extract_stack(limit=24)):
                if 'mmmm_amd/' in fr.filename and not fr.filename.endswith(('kernels.py', 'hip.py')):
                    site = f"{fr.filename.split('mmmm_amd/')[-1]}:{fr.lineno} {fr.name}"
                    break
            self.views[(site, name)] += 1
        if not any(name.startswith(v) for v in VIEWS):
            site = '?'
            for fr in reversed(traceback.extract_stack(limit=24)):
                if 'mmmm_amd/' in fr.filename and not fr.filename.endswith(('kernels.py', 'hip.py')):
                    site = f"{fr.filename.split('mmmm_amd/')[-1]}:{fr.lineno} {fr.name}"
                    break
            self.count[(site, name)] += 1
        return func(*args, **(kwargs or {}))


def step(mode=None):
    ddp.zero_grad()
    if mode is not None:
        with mode:
            loss = model.training_step(batch)
    else:
        loss = model.training_step(batch)
    loss.backward()
    ddp.finish()
    opt.step()


step()
step()
sites = Sites()
step(sites)
print('== forward: aten ops by innermost mmmm_amd frame (views excluded)')
by_site = collections.Counter()
for (site, name), c in sites.count.items():
    by_site[site] += c
for site, c in by_site.most_common(60):
    ops = ', '.join(f'{n} x{k}' for (s, n), k in sorted(sites.count.items(), key=lambda kv: -kv[1]) if s == site)
    print(f'{c:6d}  {site}   [{ops[:150]}]')

print('== forward: slices / selects of tensors that require grad (their backward is zeros + copy per use)')
for (site, name), c in sites.views.most_common(40):
    print(f'{c:6d}  {site}   [{name}]')

with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
bwd, bwd_k = collections.Counter(), collections.Counter()
for e in prof.events():
    if e.device_type.name != 'CPU' or not e.kernels or any(c.kernels for c in e.cpu_children):
        continue
    p, node = e.cpu_parent, None
    while p is not None:
        if p.name.startswith('autograd::engine::evaluate_function'):
            node = p.name.split(': ')[-1]
            break
        p = p.cpu_parent
    if node is not None:
        bwd[(node, e.name)] += 1
        bwd_k[(node, e.name)] += len(e.kernels)
print('== backward: ops that launched kernels, by autograd node')
for k, c in bwd.most_common(50):
    print(f'{c:6d} ops {bwd_k[k]:6d} kernels  {k[1]:28s} {k[0]}')
