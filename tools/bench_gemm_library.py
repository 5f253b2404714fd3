"""How far is the hand-written bf16 GEMM from the vendor library on the step's shapes? torch.mm (hipBLASLt / rocBLAS behind ATen) on the same
random operands, PLAIN products only (no LoRA extension, no expert segments, no epilogue) — a yardstick, not a product path."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from mmmm_amd import kernels as K
from tools.bench_kernels import timeit
dev = torch.device('cuda:0')
shapes = [('vit qkv', 6280, 5376, 1792), ('vit dense', 6280, 1792, 1792), ('vit fc1', 6280, 15360, 1792), ('vit fc2', 6280, 1792, 15360),
          ('dec qkv', 3648, 12288, 4096), ('dec dense', 3648, 4096, 4096), ('dec gate', 3648, 11008, 4096), ('dec down', 3648, 4096, 11008),
          ('grg dense', 4128, 4096, 4096), ('sq 8192', 8192, 8192, 8192)]
for name, M, N, Kd in shapes:
    a = [torch.randn(M, Kd, device=dev).bfloat16() for _ in range(3)]
    w = [(torch.randn(N, Kd, device=dev) / 64).bfloat16() for _ in range(3)]
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    i = [0]
    def ours():
        i[0] += 1
        K.gemm(a[i[0] % 3], w[i[0] % 3], out=out)
    def lib():
        i[0] += 1
        torch.mm(a[i[0] % 3], w[i[0] % 3].t(), out=out)
    res = {'ours': [], 'lib': []}
    for _ in range(3):
        res['ours'].append(timeit(ours, iters=20))
        res['lib'].append(timeit(lib, iters=20))
    fl = 2 * M * N * Kd
    print(f'{name:10s} [{M} x {N}] K {Kd}: hand-written {min(res["ours"])*1e3:7.1f} us {fl/min(res["ours"])/1e9:6.0f} TF   library {min(res["lib"])*1e3:7.1f} us {fl/min(res["lib"])/1e9:6.0f} TF', flush=True)
