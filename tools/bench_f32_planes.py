"""Would the heads' fp32 GEMMs be faster as ONE launch of the 256-column bf16 kernel over pre-split [hi | lo] planes (K' = 3 K, fp32 output)?
Times the bf16 kernel on K' = 3 K random operands against the fp32 three-product kernel on the same shapes (round 5: no — DESIGN.md section 3)."""
import sys
sys.path.insert(0, '.')
import torch
from mmmm_amd import kernels as K
from tools.bench_kernels import timeit
dev = torch.device('cuda:0')
for M in (3136, 8192, 12544):
    for N, Kd in ((3072, 768), (768, 3072), (768, 768), (2304, 768)):
        a = torch.randn(M, Kd, device=dev)
        w = torch.randn(N, Kd, device=dev) / 30
        out = torch.empty(M, N, device=dev)
        t32 = min(timeit(lambda: K.gemm(a, w, out=out, f32_split=2), iters=20) for _ in range(2))
        ab = torch.randn(M, 3 * Kd, device=dev).bfloat16()
        wb = (torch.randn(N, 3 * Kd, device=dev) / 30).bfloat16()
        tb = min(timeit(lambda: K.gemm(ab, wb, out=out), iters=20) for _ in range(2))
        # the plane split of the activation: one pass, read 4 B write 4 B per element -> approximate with a cast kernel
        ts = min(timeit(lambda: K.cast(a, torch.bfloat16), iters=20) for _ in range(2))
        print(f'[{M} x {N}] K {Kd}: fp32 3-product kernel {t32*1e3:6.1f} us | bf16 256-kernel on K\'=3K planes, fp32 out {tb*1e3:6.1f} us (+ split pass ~{2*ts*1e3:4.1f} us)', flush=True)
