"""Histogram of the autograd graph of ONE training step of the benchmarked workload: how many backward nodes of which kind does the engine walk
(host enqueue of the backward pass: ~100 ms per step)?   usage (GPU box): python3 tools/graph_nodes.py [workload]"""
import collections
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from mmmm_amd.ddp import BucketedGradAllReduce
from mmmm_amd.models.lora import ActivationBudget

dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
name = sys.argv[1] if len(sys.argv) > 1 else 'phase-vg-448'
w = bench.WORKLOADS[name]
model, tok = bench.build(w, dev, 1.0)
trainable = [p for p in model.parameters() if p.requires_grad]
ddp = BucketedGradAllReduce(trainable, world_size=1)
batch = bench.make_batch(w, tok, bench.ALSO_BATCH.get(name, 8), dev, seed=0)
ActivationBudget.limit = 160 << 30
ddp.zero_grad()
loss = model.training_step(batch)
seen, stack, hist = set(), [loss.grad_fn], collections.Counter()
edges = 0
while stack:
    f = stack.pop()
    if f is None or f in seen:
        continue
    seen.add(f)
    hist[type(f).__name__] += 1
    for g, _ in f.next_functions:
        edges += 1
        if g is not None and g not in seen:
            stack.append(g)
print(f'{name}: {len(seen)} backward nodes, {edges} edges')
for k, v in hist.most_common(60):
    print(f'{v:6d}  {k}')
# every view node with what consumes and what produces it
pairs = collections.Counter()
for f in seen:
    for g, _ in f.next_functions:
        if g is not None and type(g).__name__ in ('ViewBackward0', 'ReshapeAliasBackward0', 'UnsafeViewBackward0', 'TBackward0', 'SliceBackward0', 'SelectBackward0'):
            prod = g.next_functions[0][0]
            pairs[(type(g).__name__, type(f).__name__, type(prod).__name__ if prod is not None else 'None')] += 1
print('view node: consumer <- view <- producer')
for (v, c, p_), n in pairs.most_common(40):
    print(f'{n:6d}  {c} <- {v} <- {p_}')
# AccumulateGrad nodes by the node that feeds them
acc = collections.Counter()
for f in seen:
    for g, _ in f.next_functions:
        if g is not None and type(g).__name__ == 'AccumulateGrad':
            acc[type(f).__name__] += 1
print('AccumulateGrad edges by consumer:', dict(acc.most_common(20)))
loss.backward()
torch.cuda.synchronize()
