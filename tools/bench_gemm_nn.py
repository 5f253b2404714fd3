"""dgrad GEMM dx = dy W: NT kernel on a transposed weight copy vs the NN operand form reading W as stored (b_nn), on the step's shapes"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from mmmm_amd import kernels as K
from tools.bench_kernels import timeit
dev = torch.device('cuda:0')
# (M, N_w = contraction, K_w = output)
for M, Nw, Kw in [(6280, 5376, 1792), (6280, 1792, 1792), (6280, 15360, 1792), (6280, 1792, 15360), (3648, 12288, 4096), (3648, 4096, 4096),
                  (3648, 11008, 4096), (3648, 4096, 11008)]:
    dy = torch.randn(M, Nw, device=dev).bfloat16()
    W = (torch.randn(Nw, Kw, device=dev) / Nw ** 0.5).bfloat16()
    Wt = K.transpose(W)
    u = (torch.randn(M, 64, device=dev) * 0.3).bfloat16()
    At = (torch.randn(Kw, 64, device=dev) * 0.1).bfloat16()
    kw = dict(a2=u, b2=At, alpha2=0.7, drop_p=0.05, drop_seed=5)
    t_nt = timeit(lambda: K.gemm(dy, Wt, **kw), iters=20)
    t_nn = timeit(lambda: K.gemm(dy, W, b_nn=True, **kw), iters=20)
    fl = 2 * M * Kw * (Nw + 64)
    print(f'dy [{M} x {Nw}] . W [{Nw} x {Kw}]: NT {t_nt*1e3:.0f} us ({fl/t_nt/1e9:.0f} TF)  NN {t_nn*1e3:.0f} us ({fl/t_nn/1e9:.0f} TF)', flush=True)
