"""The bf16 GEMM's work scheduler (csrc/gemm.hip: sched_plan) and the stream-K form of the 256-column kernel (csrc/gemm256.hip: gemm256sk_k).

Stream-K is built, correct and OFF by default (the compute side gains what the model predicts, the operand side loses
more — workgroups that share a panel no longer walk K in step); these tests force it through the C ABI (`vm_gemm_args.workspace` +
`vm_gemm_sched_mode_(2)`) so that the shipped kernel stays honest: whole tiles bit-identical to the one-tile-per-workgroup kernel, split
tiles within one bf16 ulp, every launch bit-identical to the previous one on the same operands (slab hand-off through sc1 stores / loads
and an epoch flag: a stale slab would show up as a different tile)."""
import ctypes as C
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def K():
    from mmmm_amd import kernels
    return kernels


@pytest.fixture()
def streamk():
    from mmmm_amd import hip
    hip.lib().vm_gemm_sched_mode_(2)
    yield
    hip.lib().vm_gemm_sched_mode_(0)


def _operands(dev, M, N, Kd, K2, seed):
    g = torch.Generator(device='cpu').manual_seed(seed)
    a = torch.randn(M, Kd, generator=g).to(dev).bfloat16()
    w = (torch.randn(N, Kd, generator=g) / math.sqrt(Kd)).to(dev).bfloat16()
    w1 = (torch.randn(N, Kd, generator=g) / math.sqrt(Kd)).to(dev).bfloat16()
    a2 = torch.randn(M, K2, generator=g).to(dev).bfloat16() if K2 else None
    b2 = (torch.randn(N, K2, generator=g) * 0.05).to(dev).bfloat16() if K2 else None
    b21 = (torch.randn(N, K2, generator=g) * 0.05).to(dev).bfloat16() if K2 else None
    bias = torch.randn(N, generator=g).to(dev).bfloat16()
    res = torch.randn(M, N, generator=g).to(dev).bfloat16()
    return a, w, w1, a2, b2, b21, bias, res


# (M, N, K, K2, rows of segment 0 or None): 288 tiles of 256 rows (the phase-grg-3d decoder shape that used to fall onto 128 x 128 tiles),
# fewer tiles than CUs (every tile split three ways), whole rounds + a remainder, ragged N, a K of two tiles
SHAPES = [(4128, 4096, 1024, 64, 2064), (1500, 1792, 2048, 64, None), (6280, 5376, 256, 64, None), (2049, 3000, 512, 0, 700), (777, 2040, 128, 64, None)]


@pytest.mark.parametrize('M,N,Kd,K2,split', SHAPES)
def test_stream_k_equals_one_tile_per_workgroup_within_one_ulp_and_replays_bit_for_bit(dev, K, streamk, M, N, Kd, K2, split):
    from mmmm_amd import hip
    a, w, w1, a2, b2, b21, bias, res = _operands(dev, M, N, Kd, K2, M + N)
    counts = torch.tensor([split, M], dtype=torch.int32, device=dev) if split is not None else None
    kw = dict(w1=w1 if split is not None else None, a2=a2, b2=b2, b2_1=b21 if split is not None else None, bias=bias, bias1=bias if split is not None else None,
              residual=res, counts=counts)
    ref = K.gemm(a, w, **kw)                                   # no workspace: one tile per workgroup
    ws = K.gemm_workspace()
    outs = [K.gemm(a, w, workspace=ws, **kw) for _ in range(4)]
    torch.cuda.synchronize()
    assert int(ws[-4:].view(torch.int32).item()) == 0, 'an owner gave up waiting for a slab'
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    d = (outs[0].float() - ref.float()).abs()
    # one bf16 ulp of the linear's output x = out - residual (rounded BEFORE the residual is added, as torch does) + one of the sum
    ulp = ((ref.float() - res.float()).abs() + ref.float().abs()) * 2.0 ** -7 + 2.0 ** -9
    assert bool((d <= ulp).all()), float((d / ulp).max())
    assert float((d > 0).float().mean()) < 0.05
    # and both agree with fp32 torch
    full = a.float() @ w.float().T
    if split is not None:
        full[split:] = a[split:].float() @ w1.float().T
    if K2:
        ext = a2.float() @ b2.float().T
        if split is not None:
            ext[split:] = a2[split:].float() @ b21.float().T
        full = full + ext
    full = (full + bias.float()).bfloat16().float() + res.float()
    assert ((outs[0].float() - full).norm() / full.norm()).item() < 4e-3


def test_stream_k_fp32_output_and_device_side_row_count(dev, K, streamk):
    """fp32 C (the split-K accumulate path's consumers) and a row count that only the device knows (the kernel counts the real tile rows)"""
    M, N, Kd = 3000, 2048, 1024
    a, w, w1, *_ = _operands(dev, M, N, Kd, 0, 5)
    counts = torch.tensor([1111, 2500], dtype=torch.int32, device=dev)       # 2500 of the 3000 rows are valid
    ref = torch.zeros(M, N, device=dev)
    K.gemm(a, w, w1=w1, counts=counts, out=ref)
    out = torch.zeros(M, N, device=dev)
    K.gemm(a, w, w1=w1, counts=counts, out=out, workspace=K.gemm_workspace())
    assert torch.equal(out[2500:], torch.zeros_like(out[2500:]))
    torch.testing.assert_close(out[:2500], ref[:2500], rtol=1e-5, atol=1e-5)
    full = a.float() @ w.float().T
    full[1111:] = a[1111:].float() @ w1.float().T
    torch.testing.assert_close(out[:2500], full[:2500], rtol=2e-3, atol=2e-3)


def test_stream_k_fp8_main_product_with_bf16_extension(dev, K, streamk):
    """the e4m3 form of the same kernel (vm_gemm_fp8): the row / column scales are applied AFTER the partial tiles are summed"""
    M, N, Kd = 1300, 1792, 2048
    g = torch.Generator(device='cpu').manual_seed(9)
    a = torch.randn(M, Kd, generator=g).to(dev).bfloat16()
    w = (torch.randn(N, Kd, generator=g) / math.sqrt(Kd)).to(dev).bfloat16()
    a8, sa, inv_sa = K.quant_rows_fp8(a)
    w8, sw, inv_sw = K.quant_rows_fp8(w)
    a2 = torch.randn(M, 64, generator=g).to(dev).bfloat16()
    b2 = (torch.randn(N, 64, generator=g) * 0.05).to(dev).bfloat16()
    a2s, b2s = K.scale_rows(a2, inv_sa), K.scale_rows(b2, inv_sw)
    bias = torch.randn(N, generator=g).to(dev).bfloat16()
    ref = K.gemm_fp8(a8, sa, w8, sw, a2=a2s, b2=b2s, bias=bias)
    outs = [K.gemm_fp8(a8, sa, w8, sw, a2=a2s, b2=b2s, bias=bias, workspace=K.gemm_workspace()) for _ in range(3)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    d = (outs[0].float() - ref.float()).abs()
    assert bool((d <= ref.float().abs() * 2.0 ** -7 + 2.0 ** -9).all()) and float((d > 0).float().mean()) < 0.05
    full = (a8.view(torch.float8_e4m3fn).float() * sa[:, None]) @ (w8.view(torch.float8_e4m3fn).float() * sw[:, None]).T + a2.float() @ b2.float().T + bias.float()
    assert ((outs[0].float() - full).norm() / full.norm()).item() < 1e-2


def test_scheduler_never_puts_a_chip_filling_shape_on_the_small_tile_kernel(dev):
    """round 4's chooser sent the decoder's N = 4096 linears at 4128 / 4176 rows (288 tiles of 256 rows = 1.06 rounds) to the 128 x 128 kernel"""
    from mmmm_amd import hip
    plan = hip.lib().vm_gemm_plan_
    plan.argtypes = [C.c_int] * 6 + [C.c_void_p] * 2
    for M in (3648, 4128, 4176, 6280, 8 * 2049, 4 * 4609, 4 * 3137):
        for N, Kd in ((4096, 4096), (4096, 11008), (11008, 4096), (12288, 4096), (1792, 1792), (5376, 1792), (15360, 1792), (1792, 15360)):
            for seg in (0, 1):
                kind, rows = C.c_int(), C.c_int()
                plan(M, N, Kd, 64, seg, 0, C.addressof(kind), C.addressof(rows))
                t256 = (-(-M // 256) + seg) * -(-N // 256)
                if t256 >= 200:          # more than 200 workgroups' worth of 256 x 256 tiles
                    assert (kind.value == 1 and rows.value in (192, 256)) or (kind.value, rows.value) == (3, 256), (M, N, Kd, seg, kind.value)


# ---- plan kind 3: the FULL 256-row tiles in one launch, the rows behind each segment's last full tile in a second one (128 x 128 tiles)
TAIL_SHAPES = [(4128, 4096, 2048, 64, 2064), (4176, 4096, 2048, 64, 4104), (4128, 12288, 1024, 64, 2064), (4128, 4096, 2048, 64, 2048), (4128, 4096, 2048, 0, 0),
               (4128, 4096, 2048, 64, 4128), (4100, 4096, 2048, 64, 130), (4128, 4096, 128, 64, 2064), (4128, 4096, 128, 0, 2064)]


@pytest.mark.parametrize('M,N,Kd,K2,split', TAIL_SHAPES)
def test_full_plus_tails_launches_equal_the_single_launch_bit_for_bit(dev, K, M, N, Kd, K2, split):
    """every row is computed exactly once — full tiles by the 256-row kernel (partial tiles' workgroups leave), the < 256 leftover rows of
    each segment by the 128 x 128 kernel — with the same K order and the same instruction per product: identical bits. Segment boundaries
    on a tile edge, an empty segment, a segment that is ALL tail, LoRA extension with the dgrad dropout mask, bias and residual included."""
    from mmmm_amd import hip
    lib = hip.lib()
    a, w, w1, a2, b2, b21, bias, res = _operands(dev, M, N, Kd, K2, 7 * M + N + Kd)
    counts = torch.tensor([split, M], dtype=torch.int32, device=dev)
    kw = dict(w1=w1, a2=a2, b2=b2, b2_1=b21 if K2 else None, bias=bias, bias1=bias, residual=res, counts=counts,
              drop_p=0.05 if K2 else 0.0, drop_seed=1234)
    plan = lib.vm_gemm_plan_
    plan.argtypes = [C.c_int] * 6 + [C.c_void_p] * 2
    kind, rows = C.c_int(), C.c_int()
    plan(M, N, Kd, K2, 1, 0, C.addressof(kind), C.addressof(rows))
    assert lib.vm_gemm_tails_mode_(2) == 0          # 2: the two-launch plan wherever it is legal (the short-K shapes exercise the ring's start-up and drain)
    try:
        plan(M, N, Kd, K2, 1, 0, C.addressof(kind), C.addressof(rows))
        assert kind.value == 3, 'the shape is meant to take the two-launch plan'
        two = K.gemm(a, w, **kw)
    finally:
        lib.vm_gemm_tails_mode_(1)
    assert lib.vm_gemm_tails_mode_(0) == 0
    try:
        one = K.gemm(a, w, **kw)
    finally:
        lib.vm_gemm_tails_mode_(1)
    torch.cuda.synchronize()
    assert torch.equal(two, one), (two.float() - one.float()).abs().max().item()
