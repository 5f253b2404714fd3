"""fp8 path of BASELINE configs[4] (model-hr, "fp8 MFMA path"): the e4m3 quantiser and the fp8 GEMM against exact references, then the
model-level tolerance, which is DEFINED against the build's own bf16 path (the reference has no fp8 arithmetic)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
F8 = torch.float8_e4m3fn


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.fixture(scope='module')
def K():
    from mmmm_amd import kernels
    return kernels


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float32])
def test_quant_rows_fp8_matches_torch(dev, K, dt):
    torch.manual_seed(0)
    x = (torch.randn(300, 1792, device=dev) * torch.logspace(-3, 2, 300, device=dev)[:, None]).to(dt)
    x[7] = 0
    nrows = torch.tensor([290], dtype=torch.int32, device=dev)
    x8, sc, inv = K.quant_rows_fp8(x, nrows)
    amax = x.float().abs().amax(1)
    ref_sc = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    ref_sc[290:] = 1.0
    assert torch.allclose(sc, ref_sc, rtol=1e-6) and torch.allclose(inv, 1 / ref_sc, rtol=1e-6)
    mult = torch.where(amax > 0, 448.0 / amax, torch.ones_like(amax))
    ref8 = (x.float() * mult[:, None]).to(F8).view(torch.uint8)
    ref8[290:] = 0
    assert torch.equal(x8, ref8)
    # dequantised rows reproduce x to e4m3 precision (3 mantissa bits: 2^-4 relative per element at worst)
    deq = x8.view(F8).float() * sc[:, None]
    assert rel(deq[:290], x.float()[:290]) < 0.04


def _ref(a8, sa, w8, sw, a2=None, b2=None, alpha2=1.0, bias=None, residual=None):
    y = (a8.view(F8).double() @ w8.view(F8).double().T) * sa.double()[:, None] * sw.double()[None]
    if a2 is not None:
        y = y + alpha2 * (a2.double() * sa.double()[:, None]) @ (b2.double() * sw.double()[:, None]).T
    if bias is not None:
        y = y + bias.double()
    if residual is not None:
        y = y + residual.double()
    return y


@pytest.mark.parametrize('M,N,Kd', [(300, 520, 256), (1000, 1792, 1792), (6280, 1792, 1792), (777, 4096, 11008 // 128 * 128)])
def test_gemm_fp8_matches_dequantised_reference(dev, K, M, N, Kd):
    torch.manual_seed(1)
    x = torch.randn(M, Kd, device=dev).bfloat16()
    w = (torch.randn(N, Kd, device=dev) / Kd ** 0.5).bfloat16()
    a8, sa, inv_a = K.quant_rows_fp8(x)
    w8, sw, inv_w = K.quant_rows_fp8(w)
    bias = torch.randn(N, device=dev)
    out = K.gemm_fp8(a8, sa, w8, sw, bias=bias, out_dtype=torch.float32)
    assert rel(out, _ref(a8, sa, w8, sw, bias=bias)) < 2e-6            # exact products, fp32 accumulation: summation order only
    # against the unquantised product: two e4m3 operands with per-row scales
    assert rel(out, x.double() @ w.double().T + bias.double()) < 0.06
    # LoRA extension (pre-divided operands) + bf16 output + residual
    t = (torch.randn(M, 64, device=dev) * 0.3).bfloat16()
    B = (torch.randn(N, 64, device=dev) * 0.1).bfloat16()
    t_div, B_div = (t.float() * inv_a[:, None]).bfloat16(), (B.float() * inv_w[:, None]).bfloat16()
    res = torch.randn(M, N, device=dev).bfloat16()
    out = K.gemm_fp8(a8, sa, w8, sw, a2=t_div, b2=B_div, alpha2=0.7, residual=res)
    ref = _ref(a8, sa, w8, sw, a2=t_div, b2=B_div, alpha2=0.7).float().bfloat16().double() + res.double()
    assert rel(out, ref) < 4e-3                                          # two bf16 roundings of the output


def test_gemm_fp8_gated_segments_and_dropout_equal_the_bf16_kernel_on_exact_inputs(dev, K):
    """integer-valued operands with unit scales are exact in e4m3 AND in bf16: the fp8 kernel must then reproduce the bf16 kernel bit for
    bit in every structural feature they share — two row segments from device counts, the K-extension, its dropout mask, bias"""
    torch.manual_seed(2)
    M, N, Kd = 700, 512, 384
    xi = torch.randint(-4, 5, (M, Kd), device=dev).float()
    w0 = torch.randint(-3, 4, (N, Kd), device=dev).float()
    w1 = torch.randint(-3, 4, (N, Kd), device=dev).float()
    t = torch.randint(-2, 3, (M, 64), device=dev).float()
    b0 = torch.randint(-2, 3, (N, 64), device=dev).float()
    b1 = torch.randint(-2, 3, (N, 64), device=dev).float()
    bias0, bias1 = torch.randn(N, device=dev).bfloat16(), torch.randn(N, device=dev).bfloat16()
    counts = torch.tensor([300, 650, 0, 0], dtype=torch.int32, device=dev)
    ones_m, ones_n = torch.ones(M, device=dev), torch.ones(N, device=dev)
    f8 = lambda z: z.to(F8).view(torch.uint8)
    for drop in (0.0, 0.25):
        got = K.gemm_fp8(f8(xi), ones_m, f8(w0), ones_n, w1_8=f8(w1), sw1=ones_n, a2=t.bfloat16(), b2=b0.bfloat16(), b2_1=b1.bfloat16(), alpha2=0.5,
                         bias=bias0, bias1=bias1, counts=counts, drop_p=drop, drop_seed=99)
        want = K.gemm(xi.bfloat16(), w0.bfloat16(), w1=w1.bfloat16(), a2=t.bfloat16(), b2=b0.bfloat16(), b2_1=b1.bfloat16(), alpha2=0.5,
                      bias=bias0, bias1=bias1, counts=counts, drop_p=drop, drop_seed=99)
        assert torch.equal(got[:650], want[:650]), drop
