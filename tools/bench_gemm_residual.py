"""GEMMs whose epilogue adds a residual (decoder o_proj / down_proj forward; the post-norm blocks' input-gradient GEMMs under
functional.FORK_LINEAR), with and without it — what the residual costs the epilogue. Run once per library build (VM_LIB_PATH)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from mmmm_amd import kernels as K
from tools.bench_kernels import timeit
dev = torch.device('cuda:0')
for name, M, N, Kd in [('lm.o_proj', 3648, 4096, 4096), ('lm.down', 3648, 4096, 11008), ('vit.dqkv', 6150, 1792, 5376), ('vit.dproj', 6150, 1792, 1792),
                       ('vit.dfc1', 6150, 1792, 15360)]:
    a = torch.randn(M, Kd, device=dev).bfloat16()
    w = (torch.randn(N, Kd, device=dev) / 64).bfloat16()
    r = torch.randn(M, N, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t0 = timeit(lambda: K.gemm(a, w, out=out), iters=40)
    t1 = timeit(lambda: K.gemm(a, w, out=out, residual=r), iters=40)
    t2 = timeit(lambda: torch.add(out, r), iters=40)
    print(f'{name:10s} [{M} x {N}] x {Kd}: plain {t0*1e3:6.1f} us, with residual {t1*1e3:6.1f} us (+{(t1-t0)*1e3:.1f}), a separate add {t2*1e3:.1f} us', flush=True)
