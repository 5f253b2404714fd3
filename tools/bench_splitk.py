"""Sweep of the split-K count of the fp32 GEMM on the few-tiles / long-K weight-gradient shapes of the grounding heads.
usage: python tools/bench_splitk.py      (prints us per launch for S = 1..; the default policy's pick is marked)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmmm_amd import kernels as K  # noqa: E402


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev = torch.device('cuda', 0)
    for M, N, Kd in ((384, 768, 12544), (768, 384, 12544), (768, 768, 3136), (5, 96, 12544), (96, 96, 12544), (256, 768, 3136), (768, 256, 6272)):
        a = torch.randn(M, Kd, device=dev)
        w = torch.randn(N, Kd, device=dev)
        out = torch.zeros(M, N, device=dev)
        for split in (0, 2):
            row = []
            for S in (1, 2, 3, 4, 6, 8, 10, 12, 16, 24, 32, 48, 64):
                if S > Kd // 128:
                    break
                row.append(f'{S}:{timeit(lambda: K.gemm(a, w, out=out, accumulate=True, f32_split=split, ksplit_override=S)):.0f}')
            d = timeit(lambda: K.gemm(a, w, out=out, accumulate=True, f32_split=split))
            print(f'[{M} x {N} x {Kd}] f32_split={split} default {d:.0f} us | ' + ' '.join(row), flush=True)


if __name__ == '__main__':
    main()
