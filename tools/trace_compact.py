"""keep (start, end, queue, short kernel name) of the last `frac` of a rocprofv3 kernel trace: small enough to carry back from the GPU box
usage: python tools/trace_compact.py <kernel_trace.csv> <out.csv> [frac]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id', '?'), r['Kernel_Name']) for r in rows)
t0, t1 = ev[0][0], max(e[1] for e in ev)
cut = t1 - int((t1 - t0) * float(sys.argv[3] if len(sys.argv) > 3 else 0.25))
short = lambda n: re.sub(r'\(anonymous namespace\)::|void |at::native::', '', n).split('(')[0][:70]
with open(sys.argv[2], 'w', newline='') as f:
    w = csv.writer(f)
    for s, e, q, n in ev:
        if s >= cut:
            w.writerow((s - cut, e - cut, q, short(n)))
