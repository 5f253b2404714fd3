"""Timeline of ONE steady-state training step from a rocprofv3 kernel trace: where the wall time goes.

  python tools/step_timeline.py <kernel_trace.csv> [out.md]

A step is delimited by consecutive `adamw_k` groups (the optimizer is the last thing a step launches). For the last complete step:
wall time, union-busy / idle time of the device, per-queue busy time, phases by marker kernels (ViT forward -> ..., see MARKS), per-phase
kernel time by family, and the largest idle gaps."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
short = lambda n: re.sub(r'\(anonymous namespace\)::|void |at::native::', '', n).split('(')[0][:60]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '?'), short(r['Kernel_Name'])) for r in rows)
# step boundaries: end of the last adamw_k of a group (gap > 5 ms to the next adamw)
ad = [e for e in ev if e[3].startswith('adamw_k')]
ends = [a[1] for i, a in enumerate(ad) if i + 1 == len(ad) or ad[i + 1][0] - a[1] > 5_000_000]
assert len(ends) >= 3, 'need at least 3 optimizer steps in the trace'
t0, t1 = ends[-2], ends[-1]
st = [e for e in ev if t0 <= e[0] < t1]
out = []
P = out.append
P(f'step wall {(t1 - t0) / 1e6:.1f} ms, {len(st)} kernels')
busy = 0
cs, ce = st[0][0], st[0][1]
gaps = []
last = st[0][3]
for s, e, q, n in st[1:]:
    if s > ce:
        busy += ce - cs
        gaps.append((s - ce, ce - t0, last, n))
        cs, ce = s, e
    else:
        ce = max(ce, e)
    if e >= ce:
        last = n
busy += ce - cs
P(f'device busy (union over queues) {busy / 1e6:.1f} ms, idle {(t1 - t0 - busy) / 1e6:.1f} ms in {len(gaps)} gaps; sum of kernel durations {sum(e - s for s, e, _, _ in st) / 1e6:.1f} ms')
perq = collections.Counter()
for s, e, q, n in st:
    perq[q] += e - s
P('per-queue kernel time (ms): ' + ', '.join(f'q{q}: {v / 1e6:.1f}' for q, v in perq.most_common()))


def fam(n):
    for key, f in (('attn_f32', 'attn f32'), ('gemm256_k', 'gemm bf16'), ('gemm256w_k', 'gemm bf16'), ('gemm_nt_k<2', 'gemm bf16'), ('gemm_nt_k<4', 'gemm f32'), ('gemm_nt_f32p_k', 'gemm f32'), ('gemm_tn_k', 'gemm bf16'), ('gemm_tn_f32', 'gemm f32'), ('lora_', 'lora'), ('tn_', 'lora'),
                   ('attn16', 'attn bf16'), ('a32::fwd_k', 'attn bf16'), ('fwd_k<', 'attn bf16'), ('mfma_rate_k', 'ubench'), ('attn_f32', 'attn f32'), ('attn_delta', 'attn bf16'), ('transpose', 'transpose'), ('colsum', 'colsum'), ('norm', 'norm'),
                   ('ew_k', 'elementwise'), ('gelu_tab', 'elementwise'), ('silu', 'elementwise'), ('rope', 'elementwise'), ('gather', 'rows'), ('scatter', 'rows'), ('embedding', 'rows'),
                   ('ce_', 'ce'), ('adamw', 'adamw'), ('dice', 'loss'), ('upsample', 'upsample'), ('lsap', 'loss'), ('im2col', 'patch'), ('vectorized', 'ATen'),
                   ('elementwise', 'ATen'), ('reduce', 'ATen'), ('Cat', 'ATen'), ('index', 'ATen'), ('copyBuffer', 'copy'), ('fillBuffer', 'copy')):
        if key in n:
            return f
    return 'other'


# phases by marker kernels (first occurrence after the step start)
def first(pred, after=t0):
    for s, e, q, n in st:
        if s >= after and pred(n):
            return s
    return None


m_ce_f = first(lambda n: n.startswith('ce_fwd_k'))
m_ce_b = first(lambda n: n.startswith('ce_bwd_k'))
m_exp = first(lambda n: n.startswith('expert_index'))
m_adam = first(lambda n: n.startswith('adamw_k'))
marks = [('zero_grad, LoRA transposes', t0, m_exp), ('ViT-E + decoder forward', m_exp, m_ce_f), ('heads forward + losses + heads backward', m_ce_f, m_ce_b), ('backward (heads, LM, ViT)', m_ce_b, m_adam),
         ('clip + AdamW', m_adam, t1)]
# split backward at the last decoder attention backward kernel (attn16_dq_k<128>) = end of the LM backward
lm_b = [e for s, e, q, n in st if n.startswith(('attn16_dq_k<128', 'attn16_dq_ds_k<128', 'attn16_dkv_k<128')) and s >= (m_ce_b or t0)]
if lm_b:
    m_lmb = max(lm_b)
    marks[3:4] = [('backward: decoder', m_ce_b, m_lmb), ('backward: ViT-E', m_lmb, m_adam)]
DETAIL = []
P('')
P('| phase | wall ms | busy ms | kernel ms by family |')
P('|---|---:|---:|---|')
for name, a, b in marks:
    if a is None or b is None:
        continue
    seg = [(max(s, a), min(e, b), n) for s, e, q, n in st if s < b and e > a]
    seg.sort()
    bz = 0
    if seg:
        cs, ce = seg[0][0], seg[0][1]
        for s, e, n in seg[1:]:
            if s > ce:
                bz += ce - cs
                cs, ce = s, e
            else:
                ce = max(ce, e)
        bz += ce - cs
    fams = collections.Counter()
    for s, e, n in seg:
        fams[fam(n)] += e - s
    P(f'| {name} | {(b - a) / 1e6:.1f} | {bz / 1e6:.1f} | ' + ', '.join(f'{k} {v / 1e6:.1f}' for k, v in fams.most_common(9)) + ' |')
    cnt = collections.Counter(n for s, e, n in seg)
    dur = collections.Counter()
    for s, e, n in seg:
        dur[n] += e - s
    DETAIL.append(f'{name}: {len(seg)} launches; by count: ' + ', '.join(f'{k} x{v} ({dur[k] / 1e6:.1f} ms)' for k, v in cnt.most_common(16)))
P('')
for d in DETAIL:
    P(d)
P('')
P('kernel time by family over the step (ms): ' + ', '.join(f'{k} {v / 1e6:.1f}' for k, v in collections.Counter({f: sum(e - s for s, e, q, n in st if fam(n) == f) for f in set(fam(n) for *_, n in st)}).most_common()))
P('')
P('largest idle gaps (ms at offset ms: after -> before):')
for g, off, a, b in sorted(gaps, reverse=True)[:12]:
    P(f'  {g / 1e6:.2f} at {off / 1e6:.1f}: {a} -> {b}')
hist = collections.Counter()
for g, off, a, b in gaps:
    hist[min(int(g / 1000) // 10 * 10, 200)] += g
P('idle by gap length (us bucket -> ms): ' + ', '.join(f'{k}: {v / 1e6:.2f}' for k, v in sorted(hist.items())))
oth = collections.Counter()
for s, e, q, n in st:
    if fam(n) in ('other', 'ATen'):
        oth[n] += e - s
P('"other"/ATen kernels (ms): ' + ', '.join(f'{k} {v / 1e6:.2f}' for k, v in oth.most_common(14)))
txt = '\n'.join(out)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], 'w').write(txt + '\n')
