"""one GEMM shape launched back to back for a few seconds (for tools/clock_sample.sh, and to separate what a launch costs at the power cap with WARM operands
from what it costs inside the step, where every weight comes from HBM and every output goes to fresh memory):
    python3 tools/gemm_sustain.py M N K seconds [w4] [cold | coldw | colda | coldc]
cold: weights, activations and outputs rotate over pools far larger than the 256 MB Infinity Cache (coldw / colda / coldc: only that operand rotates).
touch (with coldw): every launch is preceded by a pass that READS its weight (does a producer that pulls the weight into the Infinity Cache make the GEMM warm?);
touchonly: that pass alone (its own cost).
Features of the step's launches: ext (LoRA K-extension, r = 64, scale 1), drop (ext + the input-gradient dropout mask, p = 0.05), bias, res (residual)."""
import sys
import time
import torch
sys.path.insert(0, '.')
from mmmm_amd import kernels as K
M, N, Kd, secs = (int(a) for a in sys.argv[1:5])
K.GEMM_W4 = 1 if 'w4' in sys.argv else 0
from mmmm_amd import hip as _hip
if 't192' in sys.argv:
    _hip.call('vm_gemm_force_tile_', 192)
if 't256' in sys.argv:
    _hip.call('vm_gemm_force_tile_', 256)
dev = torch.device('cuda:0')
cw = 'cold' in sys.argv or 'coldw' in sys.argv
ca = 'cold' in sys.argv or 'colda' in sys.argv
cc = 'cold' in sys.argv or 'coldc' in sys.argv
pool = lambda nbytes, on: max(1, int(1.5e9 // nbytes)) if on else 1
ws = [(torch.randn(N, Kd, device=dev) / Kd ** 0.5).bfloat16() for _ in range(pool(N * Kd * 2, cw))]
as_ = [torch.randn(M, Kd, device=dev).bfloat16() for _ in range(pool(M * Kd * 2, ca))]
cs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(pool(M * N * 2, cc))]
kw = {}
if 'ext' in sys.argv or 'drop' in sys.argv:
    kw.update(a2=torch.randn(M, 64, device=dev).bfloat16(), b2=(torch.randn(N, 64, device=dev) * 0.05).bfloat16(), alpha2=1.0)
if 'drop' in sys.argv:
    kw.update(drop_p=0.05, drop_seed=1234)
if 'bias' in sys.argv:
    kw.update(bias=torch.randn(N, device=dev).bfloat16())
if 'res' in sys.argv:
    kw.update(residual=torch.randn(M, N, device=dev).bfloat16())
touch = 'touch' in sys.argv or 'touchonly' in sys.argv
sink = torch.zeros((), device=dev, dtype=torch.int64)
t0 = time.time()
i = 0
while time.time() - t0 < secs:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        if touch:
            sink.add_(ws[i % len(ws)].view(torch.int32)[:, ::32].sum())        # one dword per 128-byte line of the weight
        if 'touchonly' not in sys.argv:
            K.gemm(as_[i % len(as_)], ws[i % len(ws)], out=cs[i % len(cs)], **kw)
        i += 1
    e1.record()
    torch.cuda.synchronize()
    print(f'{2.0 * M * N * Kd * 200 / (e0.elapsed_time(e1) * 1e-3) / 1e12:.0f} TFLOP/s ({e0.elapsed_time(e1) * 5:.1f} us)', flush=True)
