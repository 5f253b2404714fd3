"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv into a short, committed summary under profiles/."""
import csv
import re
import sys
from pathlib import Path


def short(n: str) -> str:
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    if n.startswith('at::native::'):
        m = re.match(r'at::native::(?:\(anonymous namespace\)::)?([\w:]+)<?', n)
        core = m.group(1) if m else n[:60]
        tag = ''
        for key in ('FillFunctor', 'CUDAFunctor_add', 'MulFunctor', 'copy_kernel', 'normal_kernel', 'uniform_kernel', 'FusedAdam', 'LpNorm',
                    'upsample_trilinear3d_backward', 'upsample_trilinear3d', 'max_pool3d', 'CatArray', 'index', 'sigmoid', 'BinaryOpScalar'):
            if key in n:
                tag = f'[{key}]'
                break
        return f'ATen {core}{tag}'
    return n.split('(')[0][:100]


def main(src: str, dst: str, title: str):
    rows = list(csv.DictReader(open(src)))
    agg = {}
    for r in rows:
        k = short(r['Name'])
        a = agg.setdefault(k, [0, 0.0])
        a[0] += int(r['Calls'])
        a[1] += float(r['TotalDurationNs'])
    tot = sum(v[1] for v in agg.values())
    out = [f'# {title}', '', f'source: rocprofv3 --kernel-trace --stats (kernel_stats.csv), total kernel time {tot / 1e6:.1f} ms', '',
           '| kernel | calls | total ms | share | avg us |', '|---|---:|---:|---:|---:|']
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        out.append(f'| `{k}` | {c} | {t / 1e6:.2f} | {100 * t / tot:.2f}% | {t / c / 1e3:.1f} |')
    Path(dst).write_text('\n'.join(out) + '\n')
    print('\n'.join(out[:30]))


if __name__ == '__main__':
    main(*sys.argv[1:4])
