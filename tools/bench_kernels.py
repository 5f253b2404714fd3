"""Micro-benchmarks of the hot kernels on the shapes of BASELINE config 2 (448^2, B=8, T=256)."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mmmm_amd import kernels as K  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    dev = torch.device('cuda:0')
    res = []
    shapes = [
        ('lm.qkv', 3648, 12288, 4096), ('lm.dense', 3648, 4096, 4096), ('lm.gate', 3648, 11008, 4096),
        ('lm.down', 3648, 4096, 11008), ('lm.head', 3648, 32008, 4096),
        ('vit.qkv', 6280, 5376, 1792), ('vit.dense', 6280, 1792, 1792), ('vit.fc1', 6280, 15360, 1792),
        ('vit.fc2', 6280, 1792, 15360), ('sq.4096', 4096, 4096, 4096), ('sq.8192', 8192, 8192, 8192),
    ]
    for name, M, N, Kd in shapes:
        a = torch.randn(M, Kd, device=dev).bfloat16()
        w = (torch.randn(N, Kd, device=dev) / 64).bfloat16()
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ms = timeit(lambda: K.gemm(a, w, out=out))
        ms_t = timeit(lambda: torch.matmul(a, w.T, out=out))
        tf = 2 * M * N * Kd / ms / 1e9
        res.append(dict(kernel='gemm_bf16', name=name, M=M, N=N, K=Kd, ms=ms, tflops=tf, hipblaslt_ms=ms_t,
                        hipblaslt_tflops=2 * M * N * Kd / ms_t / 1e9))
        print(res[-1], flush=True)
    # grouped + lora
    M, N, Kd = 3648, 12288, 4096
    a = torch.randn(M, Kd, device=dev).bfloat16()
    w0 = (torch.randn(N, Kd, device=dev) / 64).bfloat16()
    w1 = (torch.randn(N, Kd, device=dev) / 64).bfloat16()
    t = torch.randn(M, 64, device=dev).bfloat16()
    b0 = torch.randn(N, 64, device=dev).bfloat16()
    b1 = torch.randn(N, 64, device=dev).bfloat16()
    counts = torch.tensor([1576, M, 0, 0], dtype=torch.int32, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ms = timeit(lambda: K.gemm(a, w0, w1=w1, a2=t, b2=b0, b2_1=b1, counts=counts, out=out))
    res.append(dict(kernel='gemm_bf16_gated_lora', name='lm.qkv', ms=ms, tflops=2 * M * N * (Kd + 64) / ms / 1e9))
    print(res[-1], flush=True)
    # f32 gemm
    for name, M, N, Kd in [('sam.qkv', 6272, 2304, 768), ('sam.fc1', 6272, 3072, 768)]:
        a = torch.randn(M, Kd, device=dev)
        w = torch.randn(N, Kd, device=dev) / 32
        ms = timeit(lambda: K.gemm(a, w))
        res.append(dict(kernel='gemm_f32', name=name, ms=ms, tflops=2 * M * N * Kd / ms / 1e9))
        print(res[-1], flush=True)
    # attention
    for name, lens, H, hd, causal in [('vit', [785] * 8, 16, 112, False), ('lm', [456] * 8, 32, 128, True),
                                      ('vit3d', [2049] * 4, 16, 112, False)]:
        rows = sum(lens)
        qkv = torch.randn(rows, 3 * H * hd, device=dev).bfloat16()
        q, k, v = qkv[:, :H * hd], qkv[:, H * hd:2 * H * hd], qkv[:, 2 * H * hd:]
        cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=dev)
        ms = timeit(lambda: K.attn_fwd(q, k, v, cu, max(lens), H, hd, hd ** -0.5, causal))
        fl = sum(4 * l * l * hd * H for l in lens) * (0.5 if causal else 1)
        out, lse = K.attn_fwd(q, k, v, cu, max(lens), H, hd, hd ** -0.5, causal)
        dout = torch.randn_like(out)
        msb = timeit(lambda: K.attn_bwd(q, k, v, out, lse, dout, cu, max(lens), H, hd, hd ** -0.5, causal))
        res.append(dict(kernel='attn_bf16', name=name, fwd_ms=ms, fwd_tflops=fl / ms / 1e9, bwd_ms=msb,
                        bwd_tflops=2.5 * fl / msb / 1e9))
        print(res[-1], flush=True)
    # rowwise
    x = torch.randn(3648, 4096, device=dev).bfloat16()
    w = torch.ones(4096, device=dev).bfloat16()
    ms = timeit(lambda: K.rmsnorm_fwd(x, w, 1e-6))
    res.append(dict(kernel='rmsnorm_fwd', ms=ms, gbps=2 * x.numel() * 2 / ms / 1e6))
    print(res[-1], flush=True)
    Path('gpurun_out').mkdir(exist_ok=True)
    Path('gpurun_out/bench_kernels.json').write_text(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
