"""Per-(kernel, grid) durations of ONE steady training step from a rocprofv3 kernel trace: separates the shapes a kernel is launched
with (same name, different grids), and the gap in front of each launch on its queue.
  python tools/trace_by_grid.py <kernel_trace.csv> [out.md] [name filter regex]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
pat = re.compile(sys.argv[3]) if len(sys.argv) > 3 else None
short = lambda n: re.sub(r'\(anonymous namespace\)::|void |at::native::', '', n).split('(')[0][:70]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '?'), short(r['Kernel_Name']),
             int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1),
             int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 1)) or 1)) for r in rows)
ad = [e for e in ev if e[3].startswith('adamw_k')]
ends = [a[1] for i, a in enumerate(ad) if i + 1 == len(ad) or ad[i + 1][0] - a[1] > 5_000_000]
t0, t1 = ends[-2], ends[-1]
st = [e for e in ev if t0 <= e[0] < t1]
agg = collections.defaultdict(lambda: [0, 0, 0])       # (name, workgroups) -> [launches, total ns, total gap ns]
last_end = {}
for s, e, q, n, grid, wg in st:
    k = (n, grid // max(wg, 1))
    a = agg[k]
    a[0] += 1
    a[1] += e - s
    if q in last_end:
        a[2] += max(0, s - last_end[q])
    last_end[q] = max(last_end.get(q, 0), e)
out = [f'step wall {(t1 - t0) / 1e6:.1f} ms; per (kernel, workgroups): launches, total ms, avg us, avg gap in front (us)', '',
       '| kernel | workgroups | launches | total ms | avg us | avg gap us |', '|---|---:|---:|---:|---:|---:|']
for (n, wgs), (c, t, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if pat and not pat.search(n):
        continue
    if t < 200_000 and not pat:
        continue
    out.append(f'| `{n}` | {wgs} | {c} | {t / 1e6:.2f} | {t / c / 1e3:.1f} | {g / c / 1e3:.1f} |')
text = '\n'.join(out)
print(text)
if len(sys.argv) > 2 and sys.argv[2] != '-':
    open(sys.argv[2], 'w').write(text + '\n')
