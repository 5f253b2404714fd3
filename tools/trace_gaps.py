"""Idle-gap analysis of a rocprofv3 kernel trace: union of kernel intervals over all queues, idle time, largest gaps with
the kernels around them, per-queue busy time. usage: python tools/trace_gaps.py <kernel_trace.csv> [t_begin_frac]"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')) for r in rows))
t0, t1 = ev[0][0], max(e[1] for e in ev)
# last 40 % of the trace = steady steps
cut = t0 + int((t1 - t0) * float(sys.argv[2]) if len(sys.argv) > 2 else (t1 - t0) * 0.6)
ev = [e for e in ev if e[0] >= cut]
span = max(e[1] for e in ev) - ev[0][0]
busy = 0; cur_s, cur_e = ev[0][0], ev[0][1]; gaps = []
last_name = ev[0][2]
for s, e, n, q in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, last_name, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    if e >= cur_e: last_name = n
busy += cur_e - cur_s
print(f'span {span/1e6:.1f} ms, busy (union) {busy/1e6:.1f} ms, idle {(span-busy)/1e6:.1f} ms in {len(gaps)} gaps')
perq = collections.Counter()
for s, e, n, q in ev: perq[q] += e - s
print('per-queue kernel time (ms):', {q: round(v / 1e6, 1) for q, v in perq.items()})
short = lambda n: re.sub(r'\(anonymous namespace\)::', '', n)[:60]
hist = collections.Counter()
for g, a, b in gaps:
    hist[min(int(g / 1000) // 5 * 5, 100)] += g
print('idle by gap length bucket (us -> ms):', {k: round(v / 1e6, 2) for k, v in sorted(hist.items())})
agg = collections.Counter()
for g, a, b in gaps: agg[(short(a), short(b))] += g
for (a, b), g in agg.most_common(15): print(f'{g/1e6:7.2f} ms  after {a}  before {b}')
