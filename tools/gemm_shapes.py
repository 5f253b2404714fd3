"""Per-shape accounting of every GEMM launch of one full training step (in situ: real operands, real cache state).

usage: python tools/gemm_shapes.py [--workload phase-vg-448] [--depth-scale 1.0]
Wraps kernels.gemm / lora_down / gemm_tn with HIP events and prints, per distinct signature, launches, total ms and
TFLOP/s. Debug tooling for directing kernel work; not part of the product path."""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mmmm_amd import kernels as K  # noqa: E402
from mmmm_amd.ddp import BucketedGradAllReduce  # noqa: E402
from mmmm_amd.models.lora import ActivationBudget  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='phase-vg-448')
    ap.add_argument('--depth-scale', type=float, default=1.0)
    ap.add_argument('--batch', type=int, default=8)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    w = bench.WORKLOADS[args.workload]
    model, tok = bench.build(w, dev, args.depth_scale)
    trainable = [p for p in model.parameters() if p.requires_grad]
    ddp = BucketedGradAllReduce(trainable, world_size=1)
    batch = bench.make_batch(w, tok, args.batch, dev, seed=0)
    ActivationBudget.limit = 1 << 40

    def step():
        ddp.zero_grad()
        model.training_step(batch).backward()
        ddp.finish()

    step()
    step()
    recs = []
    orig = {n: getattr(K, n) for n in ('gemm', 'lora_down', 'gemm_tn')}

    def wrap(name):
        f = orig[name]

        def g(*a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = f(*a, **kw)
            e1.record()
            if name == 'gemm':
                x, wt = a[0], a[1]
                k2 = kw['a2'].shape[1] if kw.get('a2') is not None else 0
                sig = ('gemm', str(x.dtype)[6:], x.shape[0], wt.shape[0], x.shape[1], k2,
                       'gated' if kw.get('w1') is not None else '', 'res' if kw.get('residual') is not None else '',
                       'bias' if kw.get('bias') is not None else '', 'drop' if kw.get('drop_p', 0) > 0 else '')
                fl = 2.0 * x.shape[0] * wt.shape[0] * (x.shape[1] + k2)
            elif name == 'lora_down':
                x, A = a[0], a[1]
                sig = ('lora_down', 'bf16', x.shape[0], A.shape[0], x.shape[1], 0, 'gated' if kw.get('counts') is not None else '', '', '',
                       'drop' if kw.get('drop_p', 0) > 0 else '')
                fl = 2.0 * x.shape[0] * A.shape[0] * x.shape[1]
            else:
                X, Y = a[0], a[1]
                sig = ('gemm_tn', 'bf16', X.shape[1], Y.shape[1], X.shape[0], 0, 'seg%d' % kw.get('segment', -1), '', '',
                       'drop' if kw.get('drop_p', 0) > 0 else '')
                fl = 2.0 * X.shape[0] * X.shape[1] * Y.shape[1]
            recs.append((sig, fl, e0, e1))
            return out
        return g

    for n in orig:
        setattr(K, n, wrap(n))
    step()
    torch.cuda.synchronize()
    for n, f in orig.items():
        setattr(K, n, f)
    agg = collections.OrderedDict()
    for sig, fl, e0, e1 in recs:
        a = agg.setdefault(sig, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += e0.elapsed_time(e1)
        a[2] += fl
    tot = sum(a[1] for a in agg.values())
    print(f'total {tot:.1f} ms over {len(recs)} launches')
    print('| op | dtype | M | N | K | K2 | flags | launches | total ms | share | avg us | TFLOP/s |')
    print('|---|---|---:|---:|---:|---:|---|---:|---:|---:|---:|---:|')
    for sig, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        flags = ' '.join(s for s in sig[6:] if s)
        print(f'| {sig[0]} | {sig[1]} | {sig[2]} | {sig[3]} | {sig[4]} | {sig[5]} | {flags} | {n} | {ms:.2f} | {ms / tot:.1%} | '
              f'{ms / n * 1e3:.1f} | {fl / ms / 1e9:.0f} |')


if __name__ == '__main__':
    main()
