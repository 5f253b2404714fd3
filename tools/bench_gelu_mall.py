import os, sys
sys.path.insert(0, '/root/repo')
import torch
from mmmm_amd import kernels as K
dev = torch.device('cuda:0')
D, F = 1792, 15360
w1 = (torch.randn(F, D, device=dev) * 0.02).bfloat16()
for M in (6280, 3140, 1570):
    x = [torch.randn(M, D, device=dev).bfloat16() for _ in range(2)]
    # a big unrelated buffer written between iterations would flush the MALL; here the producer's output is the most recent thing written
    def timed(n=20):
        tot_g = tot_f = 0.0
        for i in range(n + 3):
            a, b, c = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            a.record(); h = K.gemm(x[i % 2], w1); b.record(); g = K.gelu(h); c.record()
            torch.cuda.synchronize()
            if i >= 3:
                tot_g += a.elapsed_time(b); tot_f += b.elapsed_time(c)
        return tot_g / n * 1e3, tot_f / n * 1e3
    tg, tf = timed()
    print(f'M={M}: fc1 {tg:.1f} us, gelu right after {tf:.1f} us ({M * F * 4 / tf / 1e6:.2f} TB/s)', flush=True)
