"""Per-tile fixed cost of the 256x256 GEMM: time vs K at fixed M, N (isolated, back-to-back launches, operands rotated through
several buffers so that launches do not find their operands in the Infinity Cache). time = rounds * (overhead + K-tiles * t_k):
the slope is the steady-state K-tile time, the intercept the per-workgroup prologue + epilogue + dispatch.
VM_GEMM_DEBUG bits (read once per process): 32 no epilogue, 64 epilogue without stores, 1 no operand traffic."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from mmmm_amd import kernels as K
dev = torch.device('cuda:0')
_tile = int(os.environ.get('GEMM_TILE', '0'))          # tool-side knob: GEMM_TILE=128|192|256|-192 -> vm_gemm_force_tile_
if _tile:
    from mmmm_amd import hip as _hip
    assert _hip.lib().vm_gemm_force_tile_(_tile) == 0
tag = f"tile={_tile or 'auto'} dbg={os.environ.get('VM_GEMM_DEBUG','0')}"
for M, N in ((6280, 15360), (6280, 1792), (3648, 4096)):
    pts = []
    for Kd in (256, 512, 1024, 1792, 4096, 8192):
        nb = max(2, min(8, int(600e6 / ((M + N) * Kd * 2))))
        As = [torch.randn(M, Kd, device=dev).bfloat16() for _ in range(nb)]
        Ws = [(torch.randn(N, Kd, device=dev) / 64).bfloat16() for _ in range(nb)]
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        for i in range(3):
            K.gemm(As[i % nb], Ws[i % nb], out=out)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 24
        a.record()
        for i in range(iters):
            K.gemm(As[i % nb], Ws[i % nb], out=out)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / iters * 1e3
        pts.append((Kd, us))
        print(tag, f'M={M} N={N} K={Kd}: {us:.1f} us  {2.0 * M * N * Kd / us / 1e6:.0f} TF', flush=True)
        del As, Ws
    # least-squares line through the points
    n = len(pts); sx = sum(k for k, _ in pts); sy = sum(u for _, u in pts); sxx = sum(k * k for k, _ in pts); sxy = sum(k * u for k, u in pts)
    slope = (n * sxy - sx * sy) / (n * sxx - sx * sx); icpt = (sy - slope * sx) / n
    print(tag, f'M={M} N={N}: {slope * 64:.3f} us per K-tile of 64 (all rounds), intercept {icpt:.1f} us', flush=True)
