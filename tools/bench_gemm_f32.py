import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import ctypes as C
import torch
from mmmm_amd import kernels as K, hip
from mmmm_amd.hip import ptr, stream, dtype_code
from tools.bench_kernels import timeit
dev = torch.device('cuda:0')
def gemm_raw(a, w, out, ksplit):
    g = hip.GemmArgs()
    g.A, g.lda = ptr(a), a.stride(0); g.B, g.ldb = ptr(w), w.stride(0); g.C, g.ldc = ptr(out), out.stride(0)
    g.M, g.N, g.K = a.shape[0], w.shape[0], a.shape[1]; g.split = -1; g.out_dtype = dtype_code(out.dtype); g.alpha2 = 1.0; g.alpha = 1.0
    g.ksplit = ksplit
    hip.call('vm_gemm_f32', C.addressof(g), stream())
for M, N, Kd in [(3136, 768, 3072), (3136, 3072, 768), (3136, 768, 768), (768, 768, 3136), (3136, 2304, 768)]:
    a = torch.randn(M, Kd, device=dev); w = torch.randn(N, Kd, device=dev) / 32
    out = torch.zeros(M, N, device=dev)
    ref = a.double() @ w.double().T
    for mode, ks in ((0, 1), (2, 1), (3, 1), (2, 4)):
        K.gemm_f32_mode(mode)
        out.zero_(); gemm_raw(a, w, out, ks)
        err = ((out.double() - ref).norm() / ref.norm()).item()
        ms = timeit(lambda: gemm_raw(a, w, out, ks), iters=20)
        print(f'M={M} N={N} K={Kd} mode={mode} ksplit={ks}: {ms*1e3:.0f} us {2*M*N*Kd/ms/1e9:.0f} TF err {err:.1e}', flush=True)
K.gemm_f32_mode(3)
