"""Wall time of the phases of one training step WITHOUT a profiler attached: HIP events recorded on the main stream at the phase
boundaries (forward: end of the language-model forward, end of the heads' forward + losses; backward: tensor hooks where the
gradient crosses from the heads into the decoder and from the decoder into the ViT). rocprofv3 adds ~5 us per launch, which
inflates the launch-bound heads phase (2 400 launches); this does not.
usage: python tools/phase_times.py [--workload phase-vg-448] [--steps 6]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mmmm_amd.ddp import BucketedGradAllReduce  # noqa: E402
from mmmm_amd.models.lora import ActivationBudget  # noqa: E402
from mmmm_amd.optim import FlatAdamW  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='phase-vg-448')
    ap.add_argument('--steps', type=int, default=6)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    w = bench.WORKLOADS[args.workload]
    model, tok = bench.build(w, dev, 1.0)
    ddp = BucketedGradAllReduce([p for p in model.parameters() if p.requires_grad], world_size=1)
    opt = FlatAdamW(ddp, lr=5e-5, weight_decay=0.01, max_grad_norm=1.0)
    batches = [bench.make_batch(w, tok, 8, dev, seed=i) for i in range(4)]
    ActivationBudget.limit = 1 << 40
    marks = {}

    def mark(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks[name] = e

    vg = model.visual_grounding
    vis = model.model.vision.forward

    def vg_wrapped(token_ids, hidden, *a, **k):
        mark('lm_fwd_end')
        if hidden.requires_grad:
            hidden.register_hook(lambda g: (mark('heads_bwd_end'), None)[1])
        return vg(token_ids, hidden, *a, **k)

    def vis_wrapped(*a, **k):
        out = vis(*a, **k)
        mark('vit_fwd_end')
        for t in (out if isinstance(out, (list, tuple)) else [out]):
            if torch.is_tensor(t) and t.requires_grad:
                t.register_hook(lambda g: (None if 'dec_bwd_end' in marks else mark('dec_bwd_end'), None)[1])
        return out

    if w['sam']:
        model.visual_grounding = vg_wrapped
    model.model.vision.forward = vis_wrapped
    rows = []
    for i in range(args.steps + 3):
        marks.clear()
        torch.cuda.synchronize()
        mark('start')
        ddp.zero_grad()
        loss = model.training_step(batches[i % 4])
        mark('fwd_end')
        loss.backward()
        mark('bwd_end')
        ddp.finish()
        opt.step()
        mark('end')
        torch.cuda.synchronize()
        if i >= 3:
            order = ['start', 'vit_fwd_end', 'lm_fwd_end', 'fwd_end', 'heads_bwd_end', 'dec_bwd_end', 'bwd_end', 'end']
            order = [o for o in order if o in marks]
            rows.append({f'{a}->{b}': marks[a].elapsed_time(marks[b]) for a, b in zip(order[:-1], order[1:])} | {'total': marks['start'].elapsed_time(marks['end'])})
    keys = list(rows[0])
    print(' | '.join(f'{k}' for k in keys))
    for r in rows:
        print(' | '.join(f'{r[k]:.1f}' for k in keys))
    print('mean: ' + ' | '.join(f'{k} {sum(r[k] for r in rows) / len(rows):.1f}' for k in keys))


if __name__ == '__main__':
    main()
