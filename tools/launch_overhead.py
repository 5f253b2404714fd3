"""Host cost of one launch as the training step pays it (the step enqueues ~4 200 C-ABI calls + ~1 900 ATen launches in ~150 ms):
the same tiny kernel through (a) hip.call = ctypes on the shipped library, (b) the tensor-level wrapper in kernels.py, (c) an ATen op,
(d) an autograd.Function around the wrapper (forward only / forward + backward), with the GPU kept idle (tiny problem sizes: the loop
measures the host, not a full queue). Companion: tools/ubench/launch_bench (the same from C++, without Python).
usage (GPU box): python3 tools/launch_overhead.py [N = 20000]"""
import sys
import time

import torch

sys.path.insert(0, '.')
from mmmm_amd import hip, kernels as K  # noqa: E402


def loop(name, n, f):
    for _ in range(200):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'{name:70s} host {(t1 - t0) / n * 1e6:6.2f} us per call   (enqueue + drain {(t2 - t0) / n * 1e6:6.2f})', flush=True)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    dev = torch.device('cuda:0')
    a = torch.randn(2048, device=dev).bfloat16()
    b = torch.randn(2048, device=dev).bfloat16()
    y = torch.empty_like(a)
    pa, pb, py = a.data_ptr(), b.data_ptr(), y.data_ptr()
    st = hip.stream()
    fn = getattr(hip.lib(), 'vm_add')
    loop('ctypes function object, arguments ready (vm_add, 2048 bf16)', n, lambda: fn(pa, pb, py, 2048, 0, st))
    loop('hip.call(name, ...) with ready arguments', n, lambda: hip.call('vm_add', pa, pb, py, 2048, 0, st))
    loop('hip.call + hip.ptr() x 3 + hip.stream()', n, lambda: hip.call('vm_add', hip.ptr(a), hip.ptr(b), hip.ptr(y), 2048, 0, hip.stream()))
    loop('kernels.add(a, b) (allocates the output)', n, lambda: K.add(a, b))
    loop('torch.add(a, b, out=y) (ATen)', n, lambda: torch.add(a, b, out=y))
    loop('a + b (ATen, allocates)', n, lambda: a + b)
    loop('torch.empty_like(a)', n, lambda: torch.empty_like(a))
    x = torch.randn(64, 256, device=dev).bfloat16()
    w = torch.randn(256, 256, device=dev).bfloat16()
    loop('kernels.gemm(x [64 x 256], w [256 x 256]) (struct of 40 fields + launch)', n, lambda: K.gemm(x, w))
    g = hip.GemmArgs()
    loop('hip.GemmArgs() alone', n, lambda: hip.GemmArgs())
    from mmmm_amd import functional as F
    xr = x.clone().requires_grad_(True)
    wp = torch.nn.Parameter(w.clone())
    loop('functional.linear forward (autograd.Function, no LoRA)', n // 4, lambda: F.linear(xr, wp))

    def fb():
        out = F.linear(xr, wp)
        out.backward(out)
    loop('functional.linear forward + backward (dx, dW: 3 GEMM launches, engine round trip)', n // 8, fb)
    ln_w = torch.nn.Parameter(torch.ones(256, device=dev).bfloat16())

    def chain():
        h = xr
        for _ in range(8):
            h = F.linear(h, wp)
        h.backward(h)
    loop('chain of 8 linears forward + backward (per chain: 8 + 16 launches, 8 nodes)', n // 32, chain)


if __name__ == '__main__':
    main()
