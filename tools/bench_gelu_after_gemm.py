"""GELU forward / backward timed INSIDE the sequence that produces its input (fc1 GEMM -> gelu; fc2-dgrad GEMM -> gelu_bwd) on one ViT-E
block's shapes: what the Infinity Cache still holds of the producer's output depends on the order the consumer walks it in. Variants are
separate builds of the library (tools/build_variant_lib.sh, -DVM_EXP_GELU_REV / -DVM_EXP_GELU_NT) selected through VM_LIB_PATH: run once
per build inside one gpurun call."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from mmmm_amd import kernels as K
dev = torch.device('cuda:0')
M, D, F = 6280, 1792, 15360
x = [torch.randn(M, D, device=dev).bfloat16() for _ in range(2)]
w1 = (torch.randn(F, D, device=dev) * 0.02).bfloat16()
dyo = [torch.randn(M, D, device=dev).bfloat16() for _ in range(2)]
w2 = (torch.randn(F, D, device=dev) * 0.02).bfloat16()      # fc2 weight [D, F] transposed for the NT dgrad: dg = dy w2t^T with w2t [F, D]
hs = [torch.randn(M, F, device=dev).bfloat16() for _ in range(2)]
tag = os.environ.get('VM_LIB_PATH', 'default').split('_')[-1].replace('.so', '')
def timed(fn_pre, fn, n=20):
    for i in range(3):
        fn(fn_pre(i), i)
    torch.cuda.synchronize()
    tot = 0.0
    for i in range(n):
        t = fn_pre(i)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(t, i); b.record()
        torch.cuda.synchronize()
        tot += a.elapsed_time(b)
    return tot / n * 1e3
us_f = timed(lambda i: K.gemm(x[i % 2], w1), lambda h, i: K.gelu(h))
us_b = timed(lambda i: K.gemm(dyo[i % 2], w2), lambda dg, i: K.gelu_bwd(hs[i % 2], dg))
print(f'{tag}: gelu fwd after fc1 {us_f:.1f} us; gelu bwd after fc2-dgrad {us_b:.1f} us', flush=True)
