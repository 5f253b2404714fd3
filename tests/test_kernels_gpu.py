"""Kernel-level numerics: every HIP entry point against a plain PyTorch fp32 reference of the same op.

Tolerances: fp32 kernels 1e-5 relative (fp32 accumulation order differs), bf16 kernels are compared
after rounding the fp32 reference to bf16 the way torch would (2^-8 relative per element)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.float(), b.float()
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


def max_err(a, b):
    return (a.float() - b.float()).abs().max().item()


@pytest.fixture(scope='module')
def K():
    from mmmm_amd import kernels
    return kernels


def test_arch(dev):
    import ctypes
    from mmmm_amd import hip
    buf = ctypes.create_string_buffer(64)
    hip.call('vm_device_arch', ctypes.addressof(buf), 64)
    assert buf.value.decode().startswith('gfx950')


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize('M,N,K_', [(128, 128, 64), (256, 384, 512), (200, 136, 192), (1, 8, 64), (777, 1000, 4096), (3648, 1792, 1792)])
def test_gemm_bf16_plain(dev, K, M, N, K_):
    g = torch.Generator(device='cpu').manual_seed(M * 7 + N)
    a = torch.randn(M, K_, generator=g).to(dev).bfloat16()
    w = (torch.randn(N, K_, generator=g) / math.sqrt(K_)).to(dev).bfloat16()
    out = K.gemm(a, w)
    ref = (a.float() @ w.float().T)
    assert rel_err(out, ref) < 4e-3
    assert max_err(out, ref.bfloat16()) <= 2 ** -6 * ref.abs().max().item() + 1e-3


@pytest.mark.parametrize('tile', ['192', '256'])
def test_gemm_bf16_suite_with_forced_big_tiles(dev, tile):
    """the tile chooser picks the 256-row / 192-row big-tile kernels only for shapes that fill the chip; `vm_gemm_force_tile_` (called by
    the `dev` fixture of the child run, tests/conftest.py) forces them, so the bf16 GEMM tests are re-run in a child process on every
    small / ragged shape too"""
    import os
    import subprocess
    import sys
    if os.environ.get('VM_TEST_CHILD'):
        pytest.skip('already inside the forced-tile child')
    env = dict(os.environ, VM_TEST_GEMM_TILE=tile, VM_TEST_CHILD='1')
    r = subprocess.run([sys.executable, '-m', 'pytest', __file__, '-m', 'gpu', '-q', '-x', '-k', 'gemm_bf16 and not forced'],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_gemm_bf16_asymmetric_identity(dev, K):
    # A = I check with an asymmetric B: catches row/col swaps in the C write
    n = 128
    a = torch.eye(n, device=dev).bfloat16()
    w = torch.arange(n * n, device=dev, dtype=torch.float32).view(n, n).remainder(251).bfloat16()
    out = K.gemm(a, w)
    assert torch.equal(out.float(), w.float().T)


def test_gemm_bf16_f32_out_bias_act_residual(dev, K):
    M, N, K_ = 300, 264, 256
    a = torch.randn(M, K_, device=dev).bfloat16()
    w = (torch.randn(N, K_, device=dev) / 16).bfloat16()
    bias = torch.randn(N, device=dev).bfloat16()
    res = torch.randn(M, N, device=dev).bfloat16()
    from mmmm_amd import hip
    out = K.gemm(a, w, bias=bias, residual=res, act=hip.ACT_GELU)
    lin = (a.float() @ w.float().T + bias.float()).bfloat16().float()
    ref = (torch.nn.functional.gelu(lin).bfloat16().float() + res.float())
    assert rel_err(out, ref) < 4e-3
    out32 = K.gemm(a, w, out_dtype=torch.float32, bias=bias.float(), act=hip.ACT_RELU)
    ref32 = torch.relu(a.float() @ w.float().T + bias.float())
    assert rel_err(out32, ref32) < 1e-5


def test_gemm_bf16_lora_extension(dev, K):
    M, N, K_, r = 260, 384, 512, 64
    a = torch.randn(M, K_, device=dev).bfloat16()
    w = (torch.randn(N, K_, device=dev) / 20).bfloat16()
    t = torch.randn(M, r, device=dev).bfloat16()
    b = (torch.randn(N, r, device=dev) / 8).bfloat16()
    out = K.gemm(a, w, a2=t, b2=b, alpha2=0.5)
    ref = a.float() @ w.float().T + 0.5 * (t.float() @ b.float().T)
    assert rel_err(out, ref) < 4e-3


def test_gemm_bf16_lora_extension_dropout_mask(dev, K):
    # extension accumulator masked by the same (seed, index) hash as vm_dropout
    M, N, r = 130, 256, 64
    a = torch.zeros(M, 64, device=dev).bfloat16()
    w = torch.zeros(N, 64, device=dev).bfloat16()
    t = torch.randn(M, r, device=dev).bfloat16()
    b = torch.randn(N, r, device=dev).bfloat16()
    p, seed = 0.25, 1234567
    out = K.gemm(a, w, a2=t, b2=b, drop_p=p, drop_seed=seed, out_dtype=torch.float32)
    ones = torch.ones(M, N, device=dev)
    mask = K.dropout(ones, p, seed)           # keep/(1-p) pattern of element (m, n)
    ref = (t.float() @ b.float().T) * mask
    assert rel_err(out, ref) < 1e-5
    frac = (mask == 0).float().mean().item()
    assert abs(frac - p) < 0.02


@pytest.mark.parametrize('split', [0, 1, 100, 128, 300, 517])
def test_gemm_bf16_two_expert_segments(dev, K, split):
    M, N, K_ = 517, 320, 256
    a = torch.randn(M, K_, device=dev).bfloat16()
    w0 = (torch.randn(N, K_, device=dev) / 16).bfloat16()
    w1 = (torch.randn(N, K_, device=dev) / 16).bfloat16()
    ref = torch.cat([a[:split].float() @ w0.float().T, a[split:].float() @ w1.float().T])
    out = K.gemm(a, w0, w1=w1, split=split)
    assert rel_err(out, ref) < 4e-3
    # device-side counts with a larger upper bound: rows past counts[1] must stay untouched
    Mmax = 640
    a_big = torch.zeros(Mmax, K_, device=dev).bfloat16()
    a_big[:M] = a
    counts = torch.tensor([split, M, 0, 0], dtype=torch.int32, device=dev)
    out_big = torch.full((Mmax, N), 7.0, device=dev).bfloat16()
    K.gemm(a_big, w0, w1=w1, counts=counts, out=out_big)
    assert rel_err(out_big[:M], ref) < 4e-3
    assert torch.all(out_big[M:] == 7.0)


@pytest.mark.parametrize('M,N,K_', [(128, 128, 32), (300, 200, 768), (9, 768, 768), (784, 3072, 768)])
def test_gemm_f32(dev, K, M, N, K_):
    a = torch.randn(M, K_, device=dev)
    w = torch.randn(N, K_, device=dev) / math.sqrt(K_)
    bias = torch.randn(N, device=dev)
    ref = (a.double() @ w.double().T + bias.double()).float()
    # arithmetic modes of vm_gemm_f32: exact f32 MFMA, split-bf16 with 6 products (fp32 products; the default of the fp32 islands),
    # split-bf16 with 3 products (16-bit products, opt-in)
    try:
        for mode, tol in ((0, 2e-6), (3, 2e-6), (2, 1.5e-5)):
            K.gemm_f32_mode(mode)
            out = K.gemm(a, w, bias=bias)
            assert rel_err(out, ref) < tol, (mode, rel_err(out, ref))
    finally:
        K.gemm_f32_mode(3)


@pytest.mark.parametrize('dt', [torch.float32, torch.bfloat16])
def test_gemm_split_k_small_output(dev, K, dt):
    """M x N of one tile with K in the thousands takes the split-K form (fp32 atomics into a zeroed C)"""
    M, N, Kd = 5, 96, 12544
    a = torch.randn(M, Kd, device=dev).to(dt)
    w = (torch.randn(N, Kd, device=dev) / 32).to(dt)
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev)
    out = K.gemm(a, w, bias=bias, residual=res, out_dtype=torch.float32)
    ref = a.double() @ w.double().T + bias.double() + res.double()
    assert rel_err(out, ref) < (2e-6 if dt == torch.float32 else 1e-5)


@pytest.mark.parametrize('M,N,Kd', [(384, 768, 12544), (768, 768, 3136), (2304, 768, 3136), (96, 40, 640)])
@pytest.mark.parametrize('split', [0, 2])
def test_gemm_f32_accumulate_into_existing_output(dev, K, M, N, Kd, split):
    """out += a w^T (the fp32 heads' weight gradients accumulate straight into their bucket slots): the few-tiles / long-K shapes go
    through the split-K atomics ON TOP of the existing content, the others through the epilogue's residual path"""
    a = torch.randn(M, Kd, device=dev)
    w = torch.randn(N, Kd, device=dev) / math.sqrt(Kd)
    out = torch.randn(M, N, device=dev)
    ref = out.double() + a.double() @ w.double().T
    got = K.gemm(a, w, out=out, accumulate=True, f32_split=split)
    assert got.data_ptr() == out.data_ptr()
    assert rel_err(out, ref) < (2e-6 if split == 0 else 1.5e-5)


@pytest.mark.parametrize('rows,cols', [(3136, 768), (3136, 3072), (100, 70), (64, 64), (65, 132), (12544, 384)])
def test_transpose_f32_vector_path(dev, K, rows, cols):
    """the 16-byte fp32 path (heads' weight gradients): exact transposition, zero K padding, column sums riding along"""
    x = torch.randn(rows, cols, device=dev)
    y = K.transpose(x, pad_to=64)
    rp = (rows + 63) // 64 * 64
    assert y.shape == (cols, rp) and torch.equal(y[:, :rows], x.T) and bool(torch.all(y[:, rows:] == 0))
    acc = torch.ones(cols, device=dev)
    y2 = K.transpose(x, pad_to=64, colsum_out=acc)
    assert torch.equal(y2, y)
    assert rel_err(acc, 1 + x.double().sum(0)) < 1e-5
    xs = torch.randn(rows, cols + 4, device=dev)[:, :cols]          # a row pitch that is not the width
    assert torch.equal(K.transpose(xs, pad_to=64)[:, :rows], xs.T)


def test_transpose(dev, K):
    for dt in (torch.bfloat16, torch.float32):
        x = torch.randn(130, 200, device=dev).to(dt)
        y = K.transpose(x, pad_to=64)
        assert y.shape == (200, 192)
        assert torch.equal(y[:, :130], x.T)
        assert torch.all(y[:, 130:] == 0)
        n = torch.tensor([100], dtype=torch.int32, device=dev)
        y = K.transpose(x, pad_to=64, nrows=n)
        assert torch.equal(y[:, :100], x[:100].T) and torch.all(y[:, 100:] == 0)


def test_transpose_batched(dev, K):
    """one launch over a descriptor table of differently shaped matrices (the LoRA factors of a model)"""
    for dt in (torch.bfloat16, torch.float32):
        shapes = [(64, 4096), (4096, 64), (64, 1792), (5, 3), (130, 200), (64, 64)]
        src = [torch.randn(r, c, device=dev).to(dt) for r, c in shapes]
        dst = [torch.full((c, r), 7.0, device=dev, dtype=dt) for r, c in shapes]
        desc = torch.tensor([[a.data_ptr(), b.data_ptr(), a.shape[0], a.shape[1], a.stride(0), b.stride(0)]
                             for a, b in zip(src, dst)], dtype=torch.int64).to(dev)
        for tiles in (1, 7, 64):
            for b in dst:
                b.fill_(7.0)
            K.transpose_batched(desc, len(src), tiles, dt)
            for a, b in zip(src, dst):
                assert torch.equal(b, a.T)


# ------------------------------------------------------------------ norms
@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('cols', [64, 256, 768, 1792, 2048, 4096, 5120])
def test_norms_with_register_rows_equal_the_generic_kernels_bit_for_bit(dev, K, dt, cols):
    """rows of up to 8 x 64 vectors are held in registers across the passes (rowwise.hip: *_r_k); same per-lane order of additions as the
    run-time-loop kernels, so outputs, statistics and input gradients are identical bits (the fp32 RMSNorm — not on the training path — to one ulp; 5120 bf16 columns / 4096 fp32 take the generic path
    either way); ragged row counts, the residual operand of the LayerNorm forward and the forked residual gradient of both backwards included"""
    from mmmm_amd import hip
    rows = 203
    x = torch.randn(rows, cols, device=dev).to(dt)
    w = (1 + 0.1 * torch.randn(cols, device=dev)).to(dt)
    b = (0.1 * torch.randn(cols, device=dev)).to(dt)
    res = torch.randn(rows, cols, device=dev).to(dt)
    dy = torch.randn(rows, cols, device=dev).to(dt)
    add = torch.randn(rows, cols, device=dev).to(dt)
    nrows = torch.tensor([190], dtype=torch.int32, device=dev)
    out = []
    try:
        for on in (1, 0):
            assert hip.lib().vm_norm_register_rows_(on) == 0
            y, rstd = K.rmsnorm_fwd(x, w, 1e-6, nrows)
            dx, _ = K.rmsnorm_bwd(x, w, dy, rstd, nrows, need_dw=False, dx_add=add)
            y2, mean, rstd2 = K.layernorm_fwd(x, w, b, 1e-5, residual=res)
            dx2, _, _ = K.layernorm_bwd(x, w, dy, mean, rstd2, need_dw=False, dx_add=add)
            y3, _, _ = K.layernorm_fwd(x, None, None, 1e-5)
            out.append((y[:190], rstd[:190], dx[:190], y2, mean, rstd2, dx2, y3))
    finally:
        hip.lib().vm_norm_register_rows_(1)
    names = ('rms y', 'rms rstd', 'rms dx', 'ln y', 'ln mean', 'ln rstd', 'ln dx', 'ln y (no affine)')
    bad = {n: (a.float() - c.float()).abs().max().item() for n, a, c in zip(names, *out) if not torch.equal(a, c)}
    if dt == torch.float32:      # hipcc contracts the fp32 RMS sum of squares into FMAs in one of the two loop forms only: one ulp of rstd
        bad = {n: e for n, e in bad.items() if not (n.startswith('rms') and e < 2e-6)}
    assert not bad, bad


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('cols', [64, 4096])
def test_rmsnorm(dev, K, dt, cols):
    rows, eps = 77, 1e-6
    x = torch.randn(rows, cols, device=dev).to(dt).requires_grad_()
    w = (1 + 0.1 * torch.randn(cols, device=dev)).to(dt).requires_grad_()
    xf, wf = x.detach().float().requires_grad_(), w.detach().float().requires_grad_()
    var = xf.pow(2).mean(-1, keepdim=True)
    ref = wf * (xf * torch.rsqrt(var + eps))
    dy = torch.randn(rows, cols, device=dev).to(dt)
    ref.backward(dy.float())
    y, rstd = K.rmsnorm_fwd(x.detach(), w.detach(), eps)
    tol = 5e-3 if dt == torch.bfloat16 else 1e-5
    assert rel_err(y, ref) < tol
    dx, dw = K.rmsnorm_bwd(x.detach(), w.detach(), dy, rstd)
    assert rel_err(dx, xf.grad) < tol
    assert rel_err(dw, wf.grad) < 1e-4
    # the gradient of a residual branch that forked off x, summed in by the kernel (fp32 sum, one rounding)
    add = torch.randn(rows, cols, device=dev).to(dt)
    dx2, _ = K.rmsnorm_bwd(x.detach(), w.detach(), dy, rstd, need_dw=False, dx_add=add)
    assert rel_err(dx2, xf.grad + add.float()) < tol


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float32])
@pytest.mark.parametrize('cols', [192, 768, 1792])
def test_layernorm(dev, K, dt, cols):
    rows, eps = 133, 1e-6
    x = torch.randn(rows, cols, device=dev).to(dt)
    w = (1 + 0.1 * torch.randn(cols, device=dev)).to(dt)
    b = (0.1 * torch.randn(cols, device=dev)).to(dt)
    res = torch.randn(rows, cols, device=dev).to(dt)
    xf, wf, bf = (t.float().clone().requires_grad_() for t in (x, w, b))
    ref = torch.nn.functional.layer_norm(xf, (cols,), wf, bf, eps)
    dy = torch.randn(rows, cols, device=dev).to(dt)
    ref.backward(dy.float())
    y, mean, rstd = K.layernorm_fwd(x, w, b, eps)
    tol = 5e-3 if dt == torch.bfloat16 else 1e-5
    assert rel_err(y, ref) < tol
    y2, _, _ = K.layernorm_fwd(x, w, b, eps, residual=res)
    assert rel_err(y2, ref.detach().to(dt).float() + res.float()) < tol
    dx, dw, db = K.layernorm_bwd(x, w, dy, mean, rstd)
    assert rel_err(dx, xf.grad) < tol
    assert rel_err(dw, wf.grad) < 1e-4 and rel_err(db, bf.grad) < 1e-4
    dx2, _, _ = K.layernorm_bwd(x, w, dy, mean, rstd, need_dw=False, dx_add=res)
    assert rel_err(dx2, xf.grad + res.float()) < tol


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float32])
def test_forked_norms_and_linear_sum_the_residual_gradient(dev, K, dt):
    """`fork`: the op returns its input as a second output; a block uses THAT as its residual, so both gradients of the input reach
    the op's backward, which sums them in its kernel (norms: dx_add; linear: the dgrad GEMM's residual epilogue). Same result as
    autograd's own accumulation."""
    from mmmm_amd import functional as Fh
    rows, cols = 200, 256
    tol = 1e-2 if dt == torch.bfloat16 else 1e-5
    x0 = torch.randn(rows, cols, device=dev).to(dt)
    w = (1 + 0.1 * torch.randn(cols, device=dev)).to(dt)
    b = (0.1 * torch.randn(cols, device=dev)).to(dt)
    W = (torch.randn(cols, cols, device=dev) / 16).to(dt)
    gy, gp = torch.randn(rows, cols, device=dev).to(dt), torch.randn(rows, cols, device=dev).to(dt)

    def run(op, fork):
        x = x0.clone().requires_grad_()
        h = x * 1                                   # a non-leaf input, as inside a model
        if fork:
            y, p = op(h, True)
        else:
            y, p = op(h, False), h
        torch.autograd.backward([y, p], [gy, gp])
        return y.detach(), x.grad

    ops = {
        'rms': lambda h, f: Fh.rms_norm(h, w, 1e-6, None, f),
        'ln': lambda h, f: Fh.layer_norm(h, w, b, 1e-6, None, f),
        'linear': lambda h, f: Fh.linear(h, W, meta=Fh.LinearMeta(fork=f), Wt0=K.transpose(W)),
    }
    for name, op in ops.items():
        y1, g1 = run(op, True)
        y0, g0 = run(op, False)
        assert torch.equal(y1, y0), name
        assert rel_err(g1, g0) < tol, (name, rel_err(g1, g0))
    # only the passed-through output used downstream: the op's own gradient is absent, the residual's still arrives
    x = x0.clone().requires_grad_()
    _, p = Fh.rms_norm(x * 1, w, 1e-6, None, True)
    p.backward(gp)
    assert torch.equal(x.grad, gp)


def test_vit_layer_with_the_linear_fork_matches_the_default(dev, K, monkeypatch):
    """functional.FORK_LINEAR (on by default since round 5): a post-norm ViT-E layer whose residual gradients ride in the dgrad GEMM epilogues gives the
    same output and, up to one bf16 rounding per sum, the same gradients as autograd's own adds"""
    from argparse import Namespace
    from mmmm_amd import functional as Fh
    from mmmm_amd.models.cogvlm.visual import TransformerLayer
    torch.manual_seed(5)
    cfg = Namespace(hidden_size=256, num_heads=4, intermediate_size=512, layer_norm_eps=1e-6)
    layer = TransformerLayer(cfg).to(dev).bfloat16()
    x0 = torch.randn(2 * 65, 256, device=dev).bfloat16()
    cu = Fh.cu_seqlens_tensor([65, 65], dev)
    gy = torch.randn_like(x0)
    res = []
    for on in (False, True):
        monkeypatch.setattr(Fh, 'FORK_LINEAR', on)
        layer.zero_grad()
        x = x0.clone().requires_grad_()
        y = layer(x * 1, cu, 65)
        y.backward(gy)
        res.append((y.detach(), x.grad, {n: p.grad.clone() for n, p in layer.named_parameters()}))
    (y0, g0, p0), (y1, g1, p1) = res
    assert torch.equal(y0, y1)
    assert rel_err(g1, g0) < 1e-2
    for n in p0:
        assert rel_err(p1[n], p0[n]) < 2e-2, n


@pytest.mark.parametrize('switch', ['NN_DGRAD', 'WGRAD_SIDE_STREAM', 'WGRAD_GROUP_OFF', 'F32_TN_WGRAD_OFF', 'FLAT_LINEAR_OFF', 'GEMM_W4'])
def test_measured_switches_give_the_default_results(dev, K, monkeypatch, switch):
    """Alternatives that were measured and left OFF ("dead ends") stay in the library behind module-level switches; each
    is exercised here on a ViT-E-shaped layer with LoRA (dropout on: the mask replay is part of every path) against the default path:
    NN_DGRAD (input gradient from the weight as stored),
    WGRAD_SIDE_STREAM (factor gradients on a side stream), the two old fallbacks WGRAD_GROUP = 0 / F32_TN_WGRAD = 0, [r6] models.lora.FLAT_LINEAR = False
    (a reshape and a view of the same shape around every tower linear: the same kernels, so bit-identical) and kernels.GEMM_W4 = 1 (the four-wave form of the
    256-column GEMM wherever the scheduler sends a launch there: bit-identical by construction, tests/test_gemm_w4_gpu.py)."""
    from argparse import Namespace
    from mmmm_amd import functional as Fh
    from mmmm_amd.ddp import BucketedGradAllReduce
    from mmmm_amd.models.cogvlm.visual import TransformerLayer
    from mmmm_amd.models.lora import LoraConfig
    from mmmm_amd.utils import apply_lora
    torch.manual_seed(11)
    cfg = Namespace(hidden_size=256, num_heads=4, intermediate_size=512, layer_norm_eps=1e-6)
    layer = TransformerLayer(cfg).to(dev).bfloat16()
    from mmmm_amd.models.lora import Linear as _Lin
    targets = [n for n, mod in layer.named_modules() if isinstance(mod, _Lin)]
    norms = [n for n, mod in layer.named_modules() if 'norm' in n.rsplit('.', 1)[-1]]
    apply_lora(layer, LoraConfig(r=64, lora_alpha=8, lora_dropout=0.05, use_rslora=True), targets, norms)
    assert len(targets) == 4
    with torch.no_grad():
        for n, p in layer.named_parameters():
            if 'lora_B' in n:
                p.normal_(0, 0.02)
    layer.train()
    x0 = torch.randn(2 * 65, 256, device=dev).bfloat16()
    cu = Fh.cu_seqlens_tensor([65, 65], dev)
    gy = torch.randn_like(x0)
    ddp = BucketedGradAllReduce([p for p in layer.parameters() if p.requires_grad], world_size=1)
    try:
        res = []
        for on in (False, True):
            if switch == 'WGRAD_GROUP_OFF':
                monkeypatch.setattr(Fh, 'WGRAD_GROUP', not on)
            elif switch == 'F32_TN_WGRAD_OFF':
                monkeypatch.setattr(Fh, 'F32_TN_WGRAD', not on)
            elif switch == 'FLAT_LINEAR_OFF':
                from mmmm_amd.models import lora as _lora
                monkeypatch.setattr(_lora, 'FLAT_LINEAR', not on)
            elif switch == 'GEMM_W4':
                monkeypatch.setattr(K, 'GEMM_W4', 1 if on else 0)
            else:
                monkeypatch.setattr(Fh, switch, on)
            ddp.zero_grad()
            x = x0.clone().requires_grad_()
            y = layer(x * 1, cu, 65)
            y.backward(gy)
            ddp.finish()
            torch.cuda.synchronize()
            res.append((y.detach().clone(), x.grad.clone(), {n: p.grad.clone() for n, p in layer.named_parameters() if p.requires_grad}))
        (y0, g0, p0), (y1, g1, p1) = res
        assert rel_err(y1, y0) < 1e-2 and rel_err(g1, g0) < 2e-2, (switch, rel_err(y1, y0), rel_err(g1, g0))
        for n in p0:
            assert rel_err(p1[n], p0[n]) < 3e-2, (switch, n, rel_err(p1[n], p0[n]))
        if switch in ('FLAT_LINEAR_OFF', 'GEMM_W4'):
            assert torch.equal(y1, y0) and torch.equal(g1, g0), switch          # (the norm gains' gradients go through fp32 atomics: not compared bit for bit)
    finally:
        ddp.remove()
        if switch == 'GEMM_W4':
            K._gemm_w4_applied[0] = -1          # the next kernels.gemm call re-applies the product's setting


# ------------------------------------------------------------------ rope
def _rope_ref(q, k, cos, sin, pos):
    def rot(x):
        x1, x2 = x[..., :x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
        return torch.cat((-x2, x1), dim=-1)
    c, s = cos[pos][:, None, :], sin[pos][:, None, :]
    return q * c + rot(q) * s, k * c + rot(k) * s


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float32])
def test_rope(dev, K, dt):
    rows, H, hd, npos = 50, 4, 128, 40
    qkv = torch.randn(rows, 3 * H * hd, device=dev).to(dt)
    pos = torch.randint(0, npos, (rows,), device=dev, dtype=torch.int32)
    inv = 1.0 / (10000 ** (torch.arange(0, hd, 2, device=dev).float() / hd))
    fr = torch.outer(torch.arange(npos, device=dev).float(), inv)
    emb = torch.cat((fr, fr), -1)
    cos, sin = emb.cos().contiguous(), emb.sin().contiguous()
    q = qkv[:, :H * hd].float().view(rows, H, hd)
    k = qkv[:, H * hd:2 * H * hd].float().view(rows, H, hd)
    qr, kr = _rope_ref(q, k, cos, sin, pos.long())
    work = qkv.clone()
    K.rope_(work, pos, cos, sin, H, hd)
    tol = 5e-3 if dt == torch.bfloat16 else 1e-6
    assert rel_err(work[:, :H * hd].view(rows, H, hd), qr) < tol
    assert rel_err(work[:, H * hd:2 * H * hd].view(rows, H, hd), kr) < tol
    assert torch.equal(work[:, 2 * H * hd:], qkv[:, 2 * H * hd:])
    # inverse is the transpose of the rotation: <R x, y> == <x, R^T y>
    if dt == torch.float32:
        y = torch.randn_like(qkv)
        ry = y.clone()
        K.rope_(ry, pos, cos, sin, H, hd, inverse=True)
        lhs = (work[:, :2 * H * hd] * y[:, :2 * H * hd]).sum()
        rhs = (qkv[:, :2 * H * hd] * ry[:, :2 * H * hd]).sum()
        assert abs(lhs - rhs) / abs(lhs) < 1e-4


# ------------------------------------------------------------------ elementwise
@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float32])
def test_activations(dev, K, dt):
    n = 8 * 1000 + 3
    g = torch.randn(n, device=dev).to(dt)
    u = torch.randn(n, device=dev).to(dt)
    d = torch.randn(n, device=dev).to(dt)
    gf, uf = g.float().clone().requires_grad_(), u.float().clone().requires_grad_()
    ref = torch.nn.functional.silu(gf) * uf
    ref.backward(d.float())
    tol = 5e-3 if dt == torch.bfloat16 else 1e-6
    assert rel_err(K.silu_mul(g, u), ref) < tol
    dg, du = K.silu_mul_bwd(g, u, d)
    assert rel_err(dg, gf.grad) < tol and rel_err(du, uf.grad) < tol
    xf = g.float().clone().requires_grad_()
    r2 = torch.nn.functional.gelu(xf)
    r2.backward(d.float())
    assert rel_err(K.gelu(g), r2) < tol
    assert rel_err(K.gelu_bwd(g, d), xf.grad) < tol
    y = torch.relu(g)
    assert rel_err(K.relu_bwd(y, d), torch.where(y > 0, d, torch.zeros_like(d))) < 1e-6
    assert rel_err(K.add(g, u), (g.float() + u.float()).to(dt)) < 1e-6


def test_dropout_is_deterministic_and_unbiased(dev, K):
    x = torch.ones(1 << 20, device=dev).bfloat16()
    a = K.dropout(x, 0.05, 42)
    b = K.dropout(x, 0.05, 42)
    c = K.dropout(x, 0.05, 43)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert abs((a == 0).float().mean().item() - 0.05) < 2e-3
    assert abs(a.float().mean().item() - 1.0) < 5e-3


def test_dropout_mask_independence(dev, K):
    """the (seed, index) hash behind every fused dropout: drop events of neighbouring elements, of the same element in
    neighbouring rows and of the same element under consecutive seeds are uncorrelated, for several seeds and rates"""
    R, Ccols = 2048, 4096
    x = torch.ones(R, Ccols, device=dev).bfloat16()
    n = R * Ccols
    for p in (0.05, 0.1, 0.5):
        tol = 5.0 * (p * (1 - p) / n) ** 0.5 + 1e-4         # 5 sigma of the drop frequency
        prev = None
        for seed in (0, 1, 2, 12345, (1 << 40) + 7):
            d = (K.dropout(x, p, seed) == 0).float()
            assert abs(d.mean().item() - p) < tol, (p, seed, d.mean().item())
            assert abs(d.mean(0) - p).max().item() < 6.0 * (p * (1 - p) / R) ** 0.5        # no dead / always-dropped column
            z = d - p
            var = p * (1 - p)
            for a, b in ((z[:, 1:], z[:, :-1]), (z[:, 4:], z[:, :-4]), (z[1:], z[:-1]), (z[:, ::2], z[:, 1::2])):
                assert abs((a * b).mean().item() / var) < 5e-3
            if prev is not None:
                assert abs((z * prev).mean().item() / var) < 5e-3
            prev = z


def test_gather_scatter_cast(dev, K):
    src = torch.randn(40, 64, device=dev).bfloat16()
    idx = torch.tensor([3, -1, 39, 0, 3], dtype=torch.int32, device=dev)
    g = K.gather_rows(src, idx)
    assert torch.equal(g[0], src[3]) and torch.all(g[1] == 0) and torch.equal(g[2], src[39]) and torch.equal(g[4], src[3])
    out = torch.zeros(50, 64, device=dev).bfloat16()
    sidx = torch.tensor([5, -1, 7], dtype=torch.int32, device=dev)
    K.scatter_rows(src[:3], sidx, out)
    assert torch.equal(out[5], src[0]) and torch.equal(out[7], src[2]) and out.float().abs().sum() == (src[0].float().abs().sum() + src[2].float().abs().sum())
    assert torch.equal(K.cast(src, torch.float32), src.float())
    assert torch.equal(K.cast(src.float(), torch.bfloat16), src)


def test_embedding_bwd(dev, K):
    V, h, n = 50, 64, 30
    ids = torch.randint(0, 10, (n,), device=dev, dtype=torch.int32)
    ids[4] = -1
    rows = torch.arange(n, device=dev, dtype=torch.int32)
    dout = torch.randn(n, h, device=dev)
    dw = torch.zeros(V, h, device=dev)
    K.embedding_bwd(dout, ids, rows, dw)
    ref = torch.zeros(V, h, device=dev)
    m = ids >= 0
    ref.index_add_(0, ids[m].long(), dout[m])
    assert rel_err(dw, ref) < 1e-6


# ------------------------------------------------------------------ cross entropy
@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float32])
def test_weighted_ce(dev, K, dt):
    rows, vocab, ld = 37, 32008, 32064
    buf = torch.zeros(rows, ld, device=dev).to(dt)
    buf[:, :vocab] = (3 * torch.randn(rows, vocab, device=dev)).to(dt)
    logits = buf[:, :vocab]
    labels = torch.randint(0, vocab, (rows,), device=dev)
    labels[::5] = -100
    lf = logits.float().clone().requires_grad_()
    ref = torch.nn.functional.cross_entropy(lf, labels, reduction='none')
    row_loss, lse = K.ce_fwd(buf, labels, vocab)
    assert rel_err(row_loss, ref) < 1e-5
    scale = torch.rand(rows, device=dev)
    (ref * scale).sum().backward()
    dl = K.ce_bwd(buf, labels, lse, scale * (labels >= 0), vocab)
    tol = 8e-3 if dt == torch.bfloat16 else 1e-5
    assert rel_err(dl[:, :vocab], lf.grad) < tol
    assert torch.all(dl[:, vocab:] == 0)


# ------------------------------------------------------------------ routing metadata
def test_expert_index_build(dev, K):
    B, L = 3, 12
    tt = torch.tensor([
        [0, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0],
        [0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0],
        [0, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 1],
    ], device=dev)
    am = torch.ones(B, L, dtype=torch.long, device=dev)
    am[1, 8:] = 0
    am[2, 10:] = 0
    r = K.expert_index_build(tt, am)
    pm = am.bool()
    vis = torch.zeros_like(pm)
    vis[:, :-1] = (tt[:, :-1] == 1) & (tt[:, 1:] == 1)
    lang = ~vis
    vis &= pm
    lang &= pm
    mask = r['expert_mask']
    assert torch.equal((mask & 1).bool(), vis) and torch.equal((mask & 2).bool(), lang)
    counts = r['counts'].tolist()
    assert counts[0] == int(vis.sum()) and counts[1] == int(pm.sum()) and counts[2] == int(pm.sum(1).max())
    assert r['cu_seqlens'].tolist() == [0, 12, 20, 30]
    rot = r['row_of_tok'].view(B, L)
    assert torch.all(rot[~pm] == -1)
    assert sorted(rot[pm].tolist()) == list(range(counts[1]))
    assert torch.all(rot[vis] < counts[0]) and torch.all(rot[lang] >= counts[0])
    # order preserved inside each segment
    assert rot[vis].tolist() == sorted(rot[vis].tolist()) and rot[lang].tolist() == sorted(rot[lang].tolist())
    tor = r['tok_of_row']
    flat = rot.flatten()
    for t in range(B * L):
        if flat[t] >= 0:
            assert tor[flat[t]] == t
    assert r['row_of_pos'][:counts[1]].tolist() == rot[pm].tolist()


# ------------------------------------------------------------------ im2col
def test_im2col(dev, K):
    img = torch.randn(3, 4, 32, 48, device=dev)
    cols = K.im2col3d(img, (2, 16, 16))
    w = torch.randn(5, 3, 2, 16, 16, device=dev)
    ref = torch.nn.functional.conv3d(img[None], w, stride=(2, 16, 16))[0]  # [5, 2, 2, 3]
    out = cols @ w.flatten(1).T
    assert rel_err(out.T.reshape(ref.shape), ref) < 1e-5


# ------------------------------------------------------------------ attention
def _attn_ref(q, k, v, cu, scale, causal):
    """q,k,v: [rows, H, hd] fp32; block-diagonal (causal) attention"""
    out = torch.zeros_like(q)
    for i in range(len(cu) - 1):
        s, e = cu[i], cu[i + 1]
        qi, ki, vi = (t[s:e].transpose(0, 1) for t in (q, k, v))
        sc = qi @ ki.transpose(1, 2) * scale
        if causal:
            L = e - s
            sc = sc.masked_fill(torch.ones(L, L, device=q.device).triu(1).bool(), float('-inf'))
        out[s:e] = (sc.softmax(-1) @ vi).transpose(0, 1)
    return out


@pytest.mark.parametrize('hd,H,causal', [(128, 2, True), (112, 3, False), (128, 1, False), (64, 2, True), (96, 1, True), (32, 2, False), (16, 1, True)])
@pytest.mark.parametrize('lens', [[130], [64, 1, 200], [456, 456], [785]])
def test_attention_bf16(dev, K, hd, H, causal, lens):
    cu = [0]
    for l in lens:
        cu.append(cu[-1] + l)
    rows = cu[-1]
    qkv = torch.randn(rows, 3, H, hd, device=dev).bfloat16()
    q, k, v = (qkv[:, i].reshape(rows, H * hd) for i in range(3))
    qkv2 = qkv.view(rows, 3 * H * hd)
    qv, kv, vv = qkv2[:, :H * hd], qkv2[:, H * hd:2 * H * hd], qkv2[:, 2 * H * hd:]
    cu_t = torch.tensor(cu, dtype=torch.int32, device=dev)
    scale = hd ** -0.5
    out, lse = K.attn_fwd(qv, kv, vv, cu_t, max(lens), H, hd, scale, causal)
    qf, kf, vf = (t.float().view(rows, H, hd).clone().requires_grad_() for t in (q, k, v))
    ref = _attn_ref(qf, kf, vf, cu, scale, causal)
    assert rel_err(out.view(rows, H, hd), ref) < 1e-2
    dout = torch.randn(rows, H * hd, device=dev).bfloat16()
    ref.backward(dout.float().view(rows, H, hd))
    dqkv = K.attn_bwd(qv, kv, vv, out, lse, dout, cu_t, max(lens), H, hd, scale, causal)
    assert rel_err(dqkv[:, 0].view(rows, H, hd), qf.grad) < 2e-2
    assert rel_err(dqkv[:, 1].view(rows, H, hd), kf.grad) < 2e-2
    assert rel_err(dqkv[:, 2].view(rows, H, hd), vf.grad) < 2e-2


def test_attention_bf16_row_indirection(dev, K):
    # expert-sorted physical rows: sequence position -> row through row_of_pos
    H, hd, lens = 2, 128, [70, 90]
    rows = sum(lens)
    perm = torch.randperm(rows, device=dev)
    qkv_seq = torch.randn(rows, 3 * H * hd, device=dev).bfloat16()
    qkv_phys = torch.empty_like(qkv_seq)
    qkv_phys[perm] = qkv_seq           # position i lives at row perm[i]
    cu_t = torch.tensor([0, 70, 160], dtype=torch.int32, device=dev)
    sl = lambda t: (t[:, :H * hd], t[:, H * hd:2 * H * hd], t[:, 2 * H * hd:])
    o_seq, _ = K.attn_fwd(*sl(qkv_seq), cu_t, 90, H, hd, hd ** -0.5, True)
    o_phys, _ = K.attn_fwd(*sl(qkv_phys), cu_t, 90, H, hd, hd ** -0.5, True, row_of_pos=perm.int())
    assert torch.equal(o_phys[perm], o_seq)
    # the backward follows the same indirection for q/k/v/out/dout rows and for the rows it writes
    _, lse = K.attn_fwd(*sl(qkv_seq), cu_t, 90, H, hd, hd ** -0.5, True)
    do_seq = torch.randn(rows, H * hd, device=dev).bfloat16()
    do_phys = torch.empty_like(do_seq)
    do_phys[perm] = do_seq
    g_seq = K.attn_bwd(*sl(qkv_seq), o_seq, lse, do_seq, cu_t, 90, H, hd, hd ** -0.5, True)
    g_phys = K.attn_bwd(*sl(qkv_phys), o_phys, lse, do_phys, cu_t, 90, H, hd, hd ** -0.5, True, row_of_pos=perm.int())
    assert torch.equal(g_phys[perm], g_seq)


@pytest.mark.parametrize('hd,H,causal,lens', [(112, 2, False, [785, 33]), (128, 2, True, [456, 130, 1, 64]), (64, 3, False, [200]), (16, 1, True, [129]),
                                              (112, 2, False, [2049, 700]), (112, 1, False, [4609]), (128, 1, True, [3000, 1044])])      # (the 3-D / high-resolution lengths: on the workspace path since round 5b)
def test_attention_backward_workspace_path_equals_recompute_path(dev, K, monkeypatch, hd, H, causal, lens):
    """with the dS^T scratch (vm_attn_bwd_workspace_bytes) dK / dV stores dS^T and dQ is a product over it; without it dQ recomputes
    S and dP. Same bf16 dS, same key order of the dQ sum: the two must agree bit for bit — ragged lengths, the causal mask, row indirection"""
    cu = [0]
    for l in lens:
        cu.append(cu[-1] + l)
    rows = cu[-1]
    qkv = torch.randn(rows, 3 * H * hd, device=dev).bfloat16()
    sl = lambda t: (t[:, :H * hd], t[:, H * hd:2 * H * hd], t[:, 2 * H * hd:])
    cu_t = torch.tensor(cu, dtype=torch.int32, device=dev)
    perm = torch.randperm(rows, device=dev).int()
    dout = torch.randn(rows, H * hd, device=dev).bfloat16()
    for rop in (None, perm):
        out, lse = K.attn_fwd(*sl(qkv), cu_t, max(lens), H, hd, hd ** -0.5, causal, row_of_pos=rop)
        monkeypatch.setattr(K, 'ATTN_DS_MAX_BYTES', 1 << 30)
        g_ws = K.attn_bwd(*sl(qkv), out, lse, dout, cu_t, max(lens), H, hd, hd ** -0.5, causal, row_of_pos=rop)
        monkeypatch.setattr(K, 'ATTN_DS_MAX_BYTES', 0)
        g_re = K.attn_bwd(*sl(qkv), out, lse, dout, cu_t, max(lens), H, hd, hd ** -0.5, causal, row_of_pos=rop)
        assert torch.equal(g_ws, g_re)


def test_hand_waited_kernels_are_deterministic_under_load(dev, K):
    """The kernels whose LDS reads are inline assembly behind hand-placed waits (attention backward with and without the dS^T workspace,
    the grouped LoRA factor gradients with the dropout pass) must give the same bits on every launch — a missing wait shows up as a
    launch-to-launch difference long before it shows up against a tolerance. Full-chip shapes, repeated while other work is queued."""
    torch.manual_seed(0)
    H, hd, lens = 16, 112, [785] * 8
    rows = sum(lens)
    qkv = torch.randn(rows, 3 * H * hd, device=dev).bfloat16()
    sl = lambda t: (t[:, :H * hd], t[:, H * hd:2 * H * hd], t[:, 2 * H * hd:])
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=dev)
    dout = torch.randn(rows, H * hd, device=dev).bfloat16()
    out, lse = K.attn_fwd(*sl(qkv), cu, 785, H, hd, hd ** -0.5, False)
    ref = K.attn_bwd(*sl(qkv), out, lse, dout, cu, 785, H, hd, hd ** -0.5, False).clone()
    M, C_ = 6280, 1792
    W = torch.randn(M, C_, device=dev).bfloat16()
    S = torch.randn(M, 64, device=dev).bfloat16()
    def group():                                                  # (one output slot per item: a slot may appear once per launch)
        o1 = torch.zeros(6, 64, C_, device=dev); o2 = torch.zeros(6, C_, 64, device=dev)
        items = []
        for j in range(6):
            items += [(W, S, o1[j], True, None, -1, 0.5, 0.05, 1234 + j), (W, S, o2[j], False, None, -1, 0.5, 0.0, 0)]
        K.tn_skinny_group(items)
        return o1, o2
    g1, g2 = group()
    # the decoder's form of the same kernels: head width 128, causal, sequence position -> row through `row_of_pos` (expert-sorted rows)
    H2, hd2, lens2 = 32, 128, [456] * 8
    rows2 = sum(lens2)
    qkv2 = torch.randn(rows2, 3 * H2 * hd2, device=dev).bfloat16()
    sl2 = lambda t: (t[:, :H2 * hd2], t[:, H2 * hd2:2 * H2 * hd2], t[:, 2 * H2 * hd2:])
    cu2 = torch.tensor([0] + list(torch.tensor(lens2).cumsum(0)), dtype=torch.int32, device=dev)
    rop = torch.randperm(rows2, device=dev).int()
    dout2 = torch.randn(rows2, H2 * hd2, device=dev).bfloat16()
    out2, lse2 = K.attn_fwd(*sl2(qkv2), cu2, 456, H2, hd2, hd2 ** -0.5, True, row_of_pos=rop)
    ref2 = K.attn_bwd(*sl2(qkv2), out2, lse2, dout2, cu2, 456, H2, hd2, hd2 ** -0.5, True, row_of_pos=rop).clone()
    filler = torch.randn(4096, 4096, device=dev).bfloat16()
    for i in range(12):
        K.gemm(filler, filler)                                   # something else in the queue around the launches under test
        got = K.attn_bwd(*sl(qkv), out, lse, dout, cu, 785, H, hd, hd ** -0.5, False)
        assert torch.equal(got, ref), f'attention backward differs on launch {i}'
        got2 = K.attn_bwd(*sl2(qkv2), out2, lse2, dout2, cu2, 456, H2, hd2, hd2 ** -0.5, True, row_of_pos=rop)
        assert torch.equal(got2, ref2), f'causal / row-indirected attention backward differs on launch {i}'
        h1, h2 = group()
        assert torch.equal(h1, g1) and torch.equal(h2, g2), f'grouped factor gradients differ on launch {i}'


def test_attention_rare_rescale_branch(dev, K):
    # spike one key late in the sequence so the running max jumps in a later tile (guide rule 26)
    H, hd, L = 1, 128, 256
    q = torch.randn(L, H * hd, device=dev).bfloat16() * 0.1
    k = torch.randn(L, H * hd, device=dev).bfloat16() * 0.1
    v = torch.randn(L, H * hd, device=dev).bfloat16()
    k[200] = (q[30].float() * 60).bfloat16()
    cu_t = torch.tensor([0, L], dtype=torch.int32, device=dev)
    out, _ = K.attn_fwd(q, k, v, cu_t, L, H, hd, 1.0, False)
    ref = _attn_ref(q.float().view(L, H, hd), k.float().view(L, H, hd), v.float().view(L, H, hd), [0, L], 1.0, False)
    assert rel_err(out.view(L, H, hd), ref) < 1e-2


# ------------------------------------------------------------------ LoRA kernels
@pytest.mark.parametrize('M,Kd', [(1, 128), (43, 384), (3648, 4096), (500, 1792), (6280, 15360), (70, 200), (129, 8)])
def test_lora_down(dev, K, M, Kd):
    x = torch.randn(M, Kd, device=dev).bfloat16()
    A = (torch.randn(64, Kd, device=dev) / math.sqrt(Kd)).bfloat16()
    t = K.lora_down(x, A)
    assert rel_err(t, x.float() @ A.float().T) < 4e-3
    # fused dropout == standalone dropout kernel followed by the projection
    p, seed = 0.05, 77
    t2 = K.lora_down(x, A, drop_p=p, drop_seed=seed)
    xd = K.dropout(x, p, seed)
    assert rel_err(t2, xd.float() @ A.float().T) < 4e-3
    assert rel_err(t2, t) > 1e-2 or M * Kd < 2000
    # the K-split partial sums are reduced in a fixed order: bit-identical replays (checkpoint recompute relies on it)
    assert torch.equal(t2, K.lora_down(x, A, drop_p=p, drop_seed=seed))


@pytest.mark.parametrize('split', [0, 7, 16, 100, 257])
def test_lora_down_two_segments(dev, K, split):
    M, Kd = 257, 256
    x = torch.randn(M, Kd, device=dev).bfloat16()
    A0 = (torch.randn(64, Kd, device=dev) / 16).bfloat16()
    A1 = (torch.randn(64, Kd, device=dev) / 16).bfloat16()
    ref = torch.cat([x[:split].float() @ A0.float().T, x[split:].float() @ A1.float().T])
    assert rel_err(K.lora_down(x, A0, A1, split=split), ref) < 4e-3
    counts = torch.tensor([split, M, 0, 0], dtype=torch.int32, device=dev)
    xb = torch.zeros(300, Kd, device=dev).bfloat16()
    xb[:M] = x
    assert rel_err(K.lora_down(xb, A0, A1, counts=counts)[:M], ref) < 4e-3


@pytest.mark.parametrize('M,P,Q', [(64, 128, 128), (100, 64, 384), (3648, 12288, 64), (3648, 64, 4096), (777, 200, 136), (3648, 32008, 256)])
def test_gemm_tn(dev, K, M, P, Q):
    X = torch.randn(M, P, device=dev).bfloat16()
    Y = torch.randn(M, Q, device=dev).bfloat16()
    ref = X.float().T @ Y.float()
    out = K.gemm_tn(X, Y)
    assert out.shape == (P, Q) and rel_err(out, ref) < 4e-3
    out2 = K.gemm_tn(X, Y, alpha=0.5, out_dtype=torch.float32)
    assert rel_err(out2, 0.5 * ref) < 3e-3 if True else None


@pytest.mark.parametrize('M,Cw', [(6280, 1792), (3648, 4096), (517, 200), (40, 64), (6280, 15360)])
def test_tn_skinny(dev, K, M, Cw):
    """LoRA factor gradients: out[c][n] / out[n][c] = alpha * W^T S, deterministic, optional accumulate and dropout"""
    W = torch.randn(M, Cw, device=dev).bfloat16()
    S = torch.randn(M, 64, device=dev).bfloat16()
    ref = W.float().T @ S.float()
    a = K.tn_skinny(W, S, transpose_out=False, alpha=0.5)
    assert a.shape == (Cw, 64) and rel_err(a, 0.5 * ref) < 4e-3
    b = K.tn_skinny(W, S, transpose_out=True, alpha=0.5, out_dtype=torch.float32)
    assert b.shape == (64, Cw) and b.dtype == torch.float32 and rel_err(b, 0.5 * ref.T) < 1e-5
    assert torch.equal(a, K.tn_skinny(W, S, transpose_out=False, alpha=0.5))        # fixed summation order
    base = torch.randn(64, Cw, device=dev).bfloat16()
    acc = base.clone()
    K.tn_skinny(W, S, transpose_out=True, out=acc, accumulate=True)
    assert rel_err(acc, base.float() + ref.T) < 6e-3
    p, seed = 0.1, 99
    Wd = K.dropout(W, p, seed)
    d = K.tn_skinny(W, S, transpose_out=True, drop_p=p, drop_seed=seed, out_dtype=torch.float32)
    # the fused kernels mask the streamed operand and apply 1/(1-p) to the fp32 result (vm_mask8): exact against that form,
    # and equal to dropout-then-contraction up to the bf16 rounding of x/(1-p) that the fused form does not perform
    keep = (K.dropout(torch.ones_like(W), p, seed) != 0).float()
    assert rel_err(d, ((W.float() * keep).T @ S.float()).T / (1 - p)) < 1e-5
    assert rel_err(d, (Wd.float().T @ S.float()).T) < 4e-3


def test_tn_skinny_segments(dev, K):
    M, Cw = 517, 256
    W = torch.randn(640, Cw, device=dev).bfloat16()
    S = torch.randn(640, 64, device=dev).bfloat16()
    counts = torch.tensor([200, M, 0, 0], dtype=torch.int32, device=dev)
    r0 = W[:200].float().T @ S[:200].float()
    r1 = W[200:M].float().T @ S[200:M].float()
    f = torch.float32
    assert rel_err(K.tn_skinny(W, S, transpose_out=False, counts=counts, segment=0, out_dtype=f), r0) < 1e-5
    assert rel_err(K.tn_skinny(W, S, transpose_out=False, counts=counts, segment=1, out_dtype=f), r1) < 1e-5
    assert rel_err(K.tn_skinny(W, S, transpose_out=False, counts=counts, segment=-1, out_dtype=f), r0 + r1) < 1e-5
    n = torch.tensor([M], dtype=torch.int32, device=dev)
    assert rel_err(K.tn_skinny(W, S, transpose_out=True, nrows=n, out_dtype=f), (r0 + r1).T) < 1e-5
    # both segments in one launch
    o0, o1 = K.tn_skinny(W, S, transpose_out=False, counts=counts, segment=2, out_dtype=f)
    assert rel_err(o0, r0) < 1e-5 and rel_err(o1, r1) < 1e-5
    a0, a1 = torch.ones(64, Cw, device=dev), torch.full((64, Cw), 2.0, device=dev)
    K.tn_skinny(W, S, transpose_out=True, counts=counts, segment=2, out=(a0, a1), accumulate=True, alpha=0.5)
    assert rel_err(a0, 1 + 0.5 * r0.T) < 1e-5 and rel_err(a1, 2 + 0.5 * r1.T) < 1e-5
    empty = torch.tensor([0, 0, 0, 0], dtype=torch.int32, device=dev)
    assert torch.count_nonzero(K.tn_skinny(W, S, transpose_out=True, counts=empty, segment=1, out_dtype=f)) == 0


@pytest.mark.parametrize('M,Kd,seg', [(6280, 1792, False), (3648, 4096, True), (300, 15360, False), (5, 4096, False), (1000, 11008, True)])
def test_lora_down_one_launch_equals_partials_plus_reduce_kernel_bit_for_bit(dev, K, M, Kd, seg):
    """round 5: the K-split rank-64 projection sums its partials in the launch that produced them (the workgroup whose ticket is the last of
    its row block, `sc1` stores / loads, no fence) instead of a second `lora_reduce_k` launch — same split order, same additions: every
    output bit equal to the two-kernel form, launch after launch on changing inputs (a stale partial from the previous launch, or a ticket
    that did not return to zero, would show up as a different row)"""
    from mmmm_amd import hip
    lib = hip.lib()
    g = torch.Generator(device=dev).manual_seed(M + Kd)
    A0 = (torch.randn(64, Kd, device=dev, generator=g) / 8).bfloat16()
    A1 = (torch.randn(64, Kd, device=dev, generator=g) / 8).bfloat16() if seg else None
    counts = torch.tensor([M // 3, M - 3, 0, 0], dtype=torch.int32, device=dev) if seg else None
    n = M - 3 if seg else M
    filler = torch.randn(4096, 4096, device=dev).bfloat16()
    try:
        for it in range(12):
            x = torch.randn(M, Kd, device=dev, generator=g).bfloat16()
            p_, seed = (0.05, 100 + it) if it % 2 else (0.0, 0)
            lib.vm_lora_down_two_kernels_(1)
            ref = K.lora_down(x, A0, A1, counts=counts, drop_p=p_, drop_seed=seed)
            lib.vm_lora_down_two_kernels_(2)
            if it % 3 == 0:
                K.gemm(filler, filler)                     # the projection's workgroups start while another kernel drains
            out = K.lora_down(x, A0, A1, counts=counts, drop_p=p_, drop_seed=seed)
            assert torch.equal(out[:n], ref[:n]), (it, (out[:n].float() - ref[:n].float()).abs().max().item())
        tt = x[:n].float()
        if p_ > 0:
            tt = K.dropout(x, p_, seed)[:n].float()
        full = tt @ A0.float().T
        if seg:
            full[M // 3:] = tt[M // 3:] @ A1.float().T
        assert rel_err(out[:n], full) < 6e-3
    finally:
        lib.vm_lora_down_two_kernels_(0)


@pytest.mark.parametrize('M,P,Q', [(3136, 768, 3072), (3136, 3072, 768), (777, 768, 768), (100, 64, 136), (2049, 2304, 768)])
@pytest.mark.parametrize('split', [2, 3])
def test_gemm_tn_f32(dev, K, M, P, Q, split):
    """fp32 weight gradient in TN form (vm_gemm_tn_f32): C += X^T Y on the operands as stored, bias gradient (column sums of X) on the
    side, against fp64; the three-product mode carries 16 mantissa bits per product, the six-product mode fp32 products"""
    g = torch.Generator(device=dev).manual_seed(M + P)
    X = torch.randn(M, P, device=dev, generator=g)
    Y = torch.randn(M, Q, device=dev, generator=g)
    base = torch.randn(P, Q, device=dev, generator=g)
    out = base.clone()
    cs = torch.ones(P, device=dev)
    K.gemm_tn_f32(X, Y, out, colsum_out=cs, f32_split=split)
    ref = base.double() + X.double().T @ Y.double()
    err = ((out.double() - ref).norm() / ref.norm()).item()
    assert err < (2e-5 if split == 2 else 2e-6), err
    assert rel_err(cs, 1 + X.sum(0)) < 1e-5
    # strided operands (column slices of wider buffers), second accumulation on top
    Xw = torch.randn(M, P + 8, device=dev, generator=g)
    out2 = out.clone()
    K.gemm_tn_f32(Xw[:, 8:], Y, out2, f32_split=split)
    ref2 = out.double() + Xw[:, 8:].double().T @ Y.double()
    assert ((out2.double() - ref2).norm() / ref2.norm()).item() < (2e-5 if split == 2 else 2e-6)


def test_tn_skinny_group(dev, K):
    """a batch of LoRA factor gradients in one launch (vm_tn_skinny_group_bf16): every item against a torch fp32 contraction — both
    output orientations, bf16 and fp32 slots, accumulation into non-zero slots, routed row segments from device counts (an empty
    one included), a ragged column count, the dropout mask of the forward, more items than one launch holds; deterministic"""
    g = torch.Generator(device=dev).manual_seed(3)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)
    counts = torch.tensor([200, 517, 0, 0], dtype=torch.int32, device=dev)
    empty = torch.tensor([0, 0, 0, 0], dtype=torch.int32, device=dev)
    specs = []          # (M, C, transpose_out, out dtype, counts, segment, alpha, drop_p)
    for i in range(28):
        specs.append(((785, 640, 1000, 333)[i % 4], (1792, 136, 64, 4096, 264)[i % 5], bool(i & 1), (torch.bfloat16, torch.float32)[(i >> 1) & 1],
                      None, -1, (1.0, 0.5)[i % 2], 0.1 if i % 3 == 2 else 0.0))
    specs += [(640, 256, False, torch.float32, counts, 0, 1.0, 0.0), (640, 256, True, torch.float32, counts, 1, 0.5, 0.0),
              (640, 256, True, torch.bfloat16, counts, 1, 1.0, 0.1), (640, 128, False, torch.float32, empty, 1, 1.0, 0.0)]
    items, refs, outs = [], [], []
    for j, (M, Cw, tr, dt, cnt, seg, alpha, p) in enumerate(specs):
        W, S = rn(M, Cw).bfloat16(), rn(M, 64).bfloat16()
        base = rn(64, Cw) if tr else rn(Cw, 64)
        out = base.to(dt).clone()
        lo, hi = 0, M
        if cnt is not None:
            c0, c1 = int(cnt[0]), int(cnt[1])
            lo, hi = (0, c0) if seg == 0 else (c0, c1)
        Wm = W.float()
        if p > 0:
            Wm = Wm * (K.dropout(torch.ones_like(W), p, 1000 + j) != 0).float() / (1 - p)
        r = alpha * (Wm[lo:hi].T @ S.float()[lo:hi])
        refs.append(out.float() + (r.T if tr else r))
        outs.append(out)
        items.append((W, S, out, tr, cnt, seg, alpha, p, 1000 + j))
    K.tn_skinny_group(items)
    for j, (o, r) in enumerate(zip(outs, refs)):
        assert rel_err(o, r) < (6e-3 if o.dtype == torch.bfloat16 else 2e-5), (j, specs[j], rel_err(o, r))
    # same inputs, same bits
    outs2 = [torch.zeros_like(o) for o in outs]
    outs3 = [torch.zeros_like(o) for o in outs]
    K.tn_skinny_group([(it[0], it[1], o, *it[3:]) for it, o in zip(items, outs2)])
    K.tn_skinny_group([(it[0], it[1], o, *it[3:]) for it, o in zip(items, outs3)])
    assert all(torch.equal(a, b) for a, b in zip(outs2, outs3))


def test_gemm_tn_segments_and_dropout(dev, K):
    M, P, Q = 517, 64, 256
    X = torch.randn(640, P, device=dev).bfloat16()
    Y = torch.randn(640, Q, device=dev).bfloat16()
    counts = torch.tensor([200, M, 0, 0], dtype=torch.int32, device=dev)
    r0 = X[:200].float().T @ Y[:200].float()
    r1 = X[200:M].float().T @ Y[200:M].float()
    assert rel_err(K.gemm_tn(X, Y, counts=counts, segment=0), r0) < 4e-3
    assert rel_err(K.gemm_tn(X, Y, counts=counts, segment=1), r1) < 4e-3
    assert rel_err(K.gemm_tn(X, Y, counts=counts, segment=-1), r0 + r1) < 4e-3
    n = torch.tensor([M], dtype=torch.int32, device=dev)
    assert rel_err(K.gemm_tn(X, Y, nrows=n), r0 + r1) < 4e-3
    p, seed = 0.1, 4242
    Yd = K.dropout(Y, p, seed)
    got = K.gemm_tn(X, Y, counts=counts, segment=1, drop_p=p, drop_seed=seed)
    assert rel_err(got, X[200:M].float().T @ Yd[200:M].float()) < 4e-3


# ------------------------------------------------------------------ optimizer
@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float32])
def test_flat_adamw_with_clipping_matches_torch(dev, K, dt):
    """FlatAdamW (vm_adamw over the gradient buckets, clip folded in) against clip_grad_norm_ + torch.optim.AdamW"""
    from mmmm_amd.ddp import BucketedGradAllReduce
    from mmmm_amd.optim import FlatAdamW
    torch.manual_seed(0)
    shapes = [(64, 96), (33,), (7, 5, 3), (128, 64)]
    mine = [torch.nn.Parameter(torch.randn(s, device=dev).to(dt)) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    ddp = BucketedGradAllReduce(mine, world_size=1, bucket_bytes=16384)
    opt = FlatAdamW(ddp, lr=3e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05, max_grad_norm=1.0)
    ropt = torch.optim.AdamW(ref, lr=3e-3, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.05, fused=True)
    for it in range(4):
        ddp.zero_grad()
        grads = [torch.randn(s, device=dev).to(dt) * (3.0 if it % 2 == 0 else 0.01) for s in shapes]     # clipped / not clipped
        for p, r, g in zip(mine, ref, grads):
            p.grad.copy_(g)
            r.grad = g.clone()
        ddp.finish()
        norm = opt.step()
        rnorm = torch.nn.utils.clip_grad_norm_(ref, 1.0)
        ropt.step()
        assert rel_err(norm, rnorm) < (5e-3 if dt == torch.bfloat16 else 1e-5)        # torch returns the norm rounded to bf16
        for p, r in zip(mine, ref):
            assert rel_err(p, r) < (4e-3 if dt == torch.bfloat16 else 2e-6), it
    ddp.remove()


def test_flat_adamw_param_groups_schedule_and_resume_match_torch(dev, K):
    """the reference's optimizer set-up (conf/phase-vg/fit.yaml:25-42): AdamW with a decayed and an undecayed
    (NoWeightDecayParameter) group, learning rate from the warm-up + cosine schedule stepped every `frequency` steps; plus
    checkpoint / resume through a torch.optim.AdamW-layout state dict, in both directions."""
    from mmmm_amd.ddp import BucketedGradAllReduce
    from mmmm_amd.optim import FlatAdamW, CosineLRSchedule
    from mmmm_amd.param import NoWeightDecayParameter
    torch.manual_seed(1)
    shapes = [(64, 96), (96,), (40, 8), (8,), (128, 64)]
    nodecay = [False, True, False, True, False]
    mk = lambda s, nd: (NoWeightDecayParameter if nd else torch.nn.Parameter)(torch.randn(s, device=dev))
    mine = [mk(s, nd) for s, nd in zip(shapes, nodecay)]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    sched = CosineLRSchedule(2e-3, t_initial=20, warmup_t=6, warmup_prefix=True, frequency=3)
    ddp = BucketedGradAllReduce(mine, world_size=1, bucket_bytes=16384)
    opt = FlatAdamW(ddp, lr=sched, betas=(0.9, 0.95), weight_decay=0.1, max_grad_norm=1.0)
    groups = [{'params': [r for r, nd in zip(ref, nodecay) if not nd], 'weight_decay': 0.1},
              {'params': [r for r, nd in zip(ref, nodecay) if nd], 'weight_decay': 0.0}]
    ropt = torch.optim.AdamW(groups, lr=2e-3, betas=(0.9, 0.95))

    def one_step(opt_, it):
        ddp.zero_grad()
        g = torch.Generator(device=dev).manual_seed(100 + it)
        grads = [torch.randn(s, device=dev, generator=g) * (2.0 if it % 2 else 0.02) for s in shapes]
        for p, r, gr in zip(mine, ref, grads):
            p.grad.copy_(gr)
            r.grad = gr.clone()
        ddp.finish()
        lr = sched(it)
        assert opt_.current_lr() == lr
        opt_.step()
        for grp in ropt.param_groups:
            grp['lr'] = lr
        torch.nn.utils.clip_grad_norm_(ref, 1.0)
        ropt.step()

    for it in range(8):
        one_step(opt, it)
    assert sched(0) == 0.0 and sched(2) == 0.0 and sched(3) > 0          # constructor value holds for the first `frequency` steps
    for p, r in zip(mine, ref):
        assert rel_err(p, r) < 3e-6
    # resume: FlatAdamW state -> a fresh FlatAdamW, and -> torch.optim.AdamW (same layout)
    sd = opt.state_dict()
    assert [len(g['params']) for g in sd['param_groups']] == [3, 2] and [g['weight_decay'] for g in sd['param_groups']] == [0.1, 0.0]
    ddp.remove()
    mine2 = [mk(s, nd) for s, nd in zip(shapes, nodecay)]
    with torch.no_grad():
        for a, b in zip(mine2, mine):
            a.copy_(b)
    ddp2 = BucketedGradAllReduce(mine2, world_size=1, bucket_bytes=16384)
    opt2 = FlatAdamW(ddp2, lr=sched, betas=(0.9, 0.95), weight_decay=0.1, max_grad_norm=1.0)
    opt2.load_state_dict(sd)
    assert opt2.step_count == 8
    import copy
    rsd = copy.deepcopy(ropt.state_dict())           # (torch's load_state_dict aliases same-device state tensors: keep a private copy)
    opt3_params = [mk(s, nd) for s, nd in zip(shapes, nodecay)]
    with torch.no_grad():
        for a, b in zip(opt3_params, ref):
            a.copy_(b)
    ddp3 = BucketedGradAllReduce(opt3_params, world_size=1, bucket_bytes=16384)
    opt3 = FlatAdamW(ddp3, lr=sched, betas=(0.9, 0.95), weight_decay=0.1, max_grad_norm=1.0)
    opt3.load_state_dict(copy.deepcopy(rsd))                             # a torch.optim.AdamW checkpoint loads too
    mine_saved, ddp_saved = mine, ddp
    for trial_params, trial_ddp, trial_opt in ((mine2, ddp2, opt2), (opt3_params, ddp3, opt3)):
        ref_c = [torch.nn.Parameter(r.detach().clone()) for r in ref]
        ropt_c = torch.optim.AdamW([{'params': [r for r, nd in zip(ref_c, nodecay) if not nd], 'weight_decay': 0.1},
                                    {'params': [r for r, nd in zip(ref_c, nodecay) if nd], 'weight_decay': 0.0}], lr=2e-3, betas=(0.9, 0.95))
        ropt_c.load_state_dict(copy.deepcopy(rsd))
        for it in range(8, 11):
            trial_ddp.zero_grad()
            g = torch.Generator(device=dev).manual_seed(100 + it)
            grads = [torch.randn(s, device=dev, generator=g) for s in shapes]
            for p, r, gr in zip(trial_params, ref_c, grads):
                p.grad.copy_(gr)
                r.grad = gr.clone()
            trial_ddp.finish()
            trial_opt.step()
            for grp in ropt_c.param_groups:
                grp['lr'] = sched(it)
            torch.nn.utils.clip_grad_norm_(ref_c, 1.0)
            ropt_c.step()
        for p, r in zip(trial_params, ref_c):
            assert rel_err(p, r) < 3e-6
        trial_ddp.remove()


def test_lsap_matches_scipy(dev, K):
    """device Hungarian matching == scipy.optimize.linear_sum_assignment, bit-exact, ties included: random float costs,
    small-integer costs (many optimal assignments), constant matrices, duplicated columns (the dummy negative columns of
    InstanceSamLoss), ragged sizes in one launch, padding entries"""
    import numpy as np
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(7)
    R, C = 8, 16
    probs, dims = [], []
    for t in range(600):
        nr = int(rng.integers(1, R + 1))
        nc = int(rng.integers(nr, C + 1))
        kind = t % 5
        if kind == 0:
            c = rng.random((nr, nc))
        elif kind == 1:
            c = rng.integers(0, 3, (nr, nc)).astype(np.float64)
        elif kind == 2:
            c = np.full((nr, nc), 0.5)
        elif kind == 3:
            c = rng.integers(0, 5, (nr, nc)) * 0.25
            c[:, nc // 2:] = c[:, nc // 2:nc // 2 + 1]
        else:
            c = rng.standard_normal((nr, nc)) * 1e3
        probs.append(c.astype(np.float32))
        dims.append((nr, nc))
    dims.insert(5, (0, 0)); probs.insert(5, np.zeros((1, 1), np.float32))          # padding entry
    cost = torch.zeros(len(probs), R, C)
    for i, c in enumerate(probs):
        cost[i, :c.shape[0], :c.shape[1]] = torch.from_numpy(c)
    got = K.lsap(cost.to(dev), torch.tensor(dims, dtype=torch.int32, device=dev), C).cpu()
    for i, (c, (nr, nc)) in enumerate(zip(probs, dims)):
        if nr == 0:
            continue
        row, col = linear_sum_assignment(c)
        assert row.tolist() == list(range(nr))
        assert got[i, :nr].tolist() == col.tolist(), (i, c, col, got[i])


@pytest.mark.parametrize('gamma,alpha', [(2.0, None), (2.0, 0.85), (1.5, 0.25), (0.0, None)])
def test_dice_focal_fused_matches_torch_form(dev, K, gamma, alpha):
    """vm_dice_focal_fwd / _bwd against the element-wise torch form of the same module (itself pinned to the reference by
    fixture F7): values and input gradients, reduce_batch on / off, missing target, an all-background row (clipped Dice
    denominator is not reachable with sigmoid > 0, but sum t = 0 is), sizes that are not a multiple of the chunk"""
    from mmmm_amd.models.loss import DiceFocalLoss
    g = torch.Generator().manual_seed(int(gamma * 10) + (0 if alpha is None else 7))
    m = DiceFocalLoss(dice_weight=2, focal_weight=3, focal_gamma=gamma, focal_alpha=alpha)
    for shape in [(3, 1, 4, 37, 53), (2, 2, 1, 300, 301), (1, 1, 1, 8, 8)]:
        x = (torch.randn(shape, generator=g) * 3).to(dev)
        t = (torch.rand(shape, generator=g) < 0.2).to(dev)
        t[0, 0] = False                                            # all-background row
        for target in (t, None):
            for reduce_batch in (True, False):
                xa = x.clone().requires_grad_()
                xb = x.double().clone().requires_grad_()
                got = m(xa, target, reduce_batch=reduce_batch, return_dict=True)
                # reference form in fp64 on the same device: the generic path of the same module (non-fp32 input)
                want = m(xb, target, reduce_batch=reduce_batch, return_dict=True)
                assert got.keys() == want.keys()
                for k in got:
                    assert got[k].shape == want[k].shape and rel_err(got[k], want[k]) < 2e-5, (k, shape, reduce_batch)
                w = torch.randn(got['total'].shape, generator=g).to(dev) if not reduce_batch else None
                (got['total'] * w).sum().backward() if w is not None else got['total'].backward()
                (want['total'] * w.double()).sum().backward() if w is not None else want['total'].backward()
                assert rel_err(xa.grad, xb.grad) < 2e-5, (shape, reduce_batch, target is None)
    # deterministic: fixed-order second stage
    a = m(x, t, return_dict=True)['total']
    assert torch.equal(a, m(x, t, return_dict=True)['total'])


@pytest.mark.parametrize('inp,out', [((1, 28, 28), (1, 112, 112)), ((1, 112, 112), (1, 448, 448)), ((8, 16, 16), (32, 64, 64)),
                                     ((2, 7, 9), (5, 30, 20)), ((4, 12, 12), (4, 12, 12)), ((3, 10, 10), (2, 5, 7))])
def test_upsample_trilinear_matches_aten(dev, K, inp, out):
    """vm_upsample_trilinear3d_fwd / _bwd == F.interpolate(mode='trilinear', align_corners=False) and its autograd, for 2D (depth 1)
    and 3D volumes, non-integer ratios, identity and down-sampling; the gather-form backward is bit-reproducible"""
    import torch.nn.functional as F
    from mmmm_amd import functional as Fh
    g = torch.Generator().manual_seed(sum(inp) + sum(out))
    x = torch.randn(3, *inp, generator=g).to(dev)
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    y = Fh.upsample_trilinear(xa, out)
    ref = F.interpolate(xb[None], out, mode='trilinear')[0]
    assert y.shape == ref.shape and rel_err(y, ref) < 1e-6
    w = torch.randn(y.shape, generator=g).to(dev)
    (y * w).sum().backward()
    (ref * w).sum().backward()
    assert rel_err(xa.grad, xb.grad) < 1e-5
    xc = x.clone().requires_grad_()
    (Fh.upsample_trilinear(xc, out) * w).sum().backward()
    assert torch.equal(xc.grad, xa.grad)
    # 5-D form [P, M, d, h, w] as the instance head uses it
    x5 = torch.randn(2, 3, *inp, generator=g).to(dev)
    assert rel_err(Fh.upsample_trilinear(x5, out), F.interpolate(x5, out, mode='trilinear')) < 1e-6


@pytest.mark.parametrize('M,N,Kc', [(300, 520, 256), (6280, 1792, 15360), (3648, 4096, 11008), (1000, 15360, 1792)])
def test_gemm_nn_weight_form_equals_nt(dev, K, M, N, Kc):
    """dx = dy W with the weight read as stored (`b_nn`: [contraction, output] rows through transposed LDS reads) against the NT kernel on the
    transposed copy: same operand values in the same k order -> bit-identical, with two row segments, K-extension, dropout mask"""
    torch.manual_seed(4)
    dy = torch.randn(M, Kc, device=dev).bfloat16()
    W = (torch.randn(Kc, N, device=dev) / Kc ** 0.5).bfloat16()            # [contraction, output]
    W1 = (torch.randn(Kc, N, device=dev) / Kc ** 0.5).bfloat16()
    want = K.gemm(dy, K.transpose(W))
    got = K.gemm(dy, W, b_nn=True)
    assert torch.equal(got, want)
    u = (torch.randn(M, 64, device=dev) * 0.3).bfloat16()
    At, At1 = (torch.randn(N, 64, device=dev) * 0.1).bfloat16(), (torch.randn(N, 64, device=dev) * 0.1).bfloat16()
    counts = torch.tensor([M // 3, M - 5, 0, 0], dtype=torch.int32, device=dev)
    kw = dict(a2=u, b2=At, b2_1=At1, alpha2=0.7, counts=counts, drop_p=0.1, drop_seed=77)
    want = K.gemm(dy, K.transpose(W), w1=K.transpose(W1), **kw)
    got = K.gemm(dy, W, w1=W1, b_nn=True, **kw)
    assert torch.equal(got[:M - 5], want[:M - 5])


def test_gelu_table_kernels_equal_the_arithmetic_kernels_on_every_bf16_value(dev, K):
    """Tensors of >= 2^20 bf16 elements take the table kernels (rowwise.hip gelu_tab_k); smaller ones the erf arithmetic
    (ew_k). Both must give the same bits for all 65 536 inputs, NaN / inf / denormals included, forward and backward."""
    bits = torch.arange(65536, dtype=torch.int32, device=dev).to(torch.int16)
    x = bits.view(torch.bfloat16)
    g = torch.Generator(device=dev).manual_seed(5)
    reps = 16
    xs = x.repeat(reps)                                             # 2^20 elements -> table path
    dy = torch.randn(xs.shape, device=dev, generator=g).bfloat16()
    # (half of the gradients: arbitrary bit patterns — denormals, huge values, inf, NaN — against every input value)
    odd = torch.randint(0, 65536, (xs.numel() // 2,), device=dev, generator=g, dtype=torch.int32).to(torch.int16).view(torch.bfloat16)
    dy[: odd.numel()] = odd
    y_tab, dx_tab = K.gelu(xs), K.gelu_bwd(xs, dy)
    for r in range(reps):                                           # 2^16 elements per call -> arithmetic path
        sl = slice(r * 65536, (r + 1) * 65536)
        y_ar, dx_ar = K.gelu(xs[sl].clone()), K.gelu_bwd(xs[sl].clone(), dy[sl].clone())
        assert torch.equal(y_tab[sl].view(torch.int16), y_ar.view(torch.int16))
        assert torch.equal(dx_tab[sl].view(torch.int16), dx_ar.view(torch.int16))
    # and against torch on the finite values (the arithmetic kernel's own contract)
    fin = torch.isfinite(x.float())
    ref = torch.nn.functional.gelu(x.float()[fin]).bfloat16()
    assert torch.equal(y_tab[:65536][fin].view(torch.int16), ref.view(torch.int16))
    # a ragged length exercises the scalar tail
    n = (1 << 20) + 5
    xr = torch.randn(n, device=dev, generator=g).bfloat16()
    dr = torch.randn(n, device=dev, generator=g).bfloat16()
    assert torch.equal(K.gelu(xr)[-5:], K.gelu(xr[-5:].clone()))
    assert torch.equal(K.gelu_bwd(xr, dr)[-5:], K.gelu_bwd(xr[-5:].clone(), dr[-5:].clone()))
    # a tensor longer than two sweeps of the capped grid (768 workgroups x 512 lanes x 8) runs the forward's two-vectors-per-trip loop,
    # its one-vector remainder (and many grid sweeps of the backward) and the scalar tail: every input value once more, against the arithmetic kernel
    from mmmm_amd import hip
    n = 2 * 768 * 512 * 8 + 77 * 4096 + 3
    xl = x.repeat(n // 65536 + 1)[:n].clone()
    dl = torch.randn(n, device=dev, generator=g).bfloat16()
    y_tab, dx_tab = K.gelu(xl), K.gelu_bwd(xl, dl)
    try:
        assert hip.lib().vm_gelu_table_(0) == 0
        y_ar, dx_ar = K.gelu(xl), K.gelu_bwd(xl, dl)
    finally:
        hip.lib().vm_gelu_table_(1)
    assert torch.equal(y_tab.view(torch.int16), y_ar.view(torch.int16))
    assert torch.equal(dx_tab.view(torch.int16), dx_ar.view(torch.int16))


def test_mfma_peak_probe_returns_a_plausible_rate(dev, K):
    """vm_ubench_mfma_bf16 (bench.py's roofline.peak_measured): bare bf16 MFMAs on random operands for a fraction of a second — an MI355X
    sustains between ~1.5 and 2.5 PFLOP/s there (nominal 2.5 at 2.4 GHz; the clock under matrix load is lower)"""
    r = K.ubench_mfma_bf16(0.4)
    assert 1000.0 < r < 2600.0, r


@pytest.mark.parametrize('dtype,n', [(torch.bfloat16, 8 * 1000003), (torch.float32, 4 * 250007), (torch.bfloat16, 64)])
def test_sumsq_partials_is_the_sum_of_squares_and_deterministic(dev, K, dtype, n):
    """vm_sumsq_partials (the gradient-norm pass of FlatAdamW): the per-workgroup partials add up to sum x^2 (fp32 accumulation against an
    fp64 reference) and two launches give the same bits (fixed element -> workgroup map, no atomics)"""
    torch.manual_seed(3)
    x = (torch.randn(n, device=dev) * 0.3).to(dtype)
    a = torch.full((1024,), float('nan'), device=dev)
    b = torch.full((1024,), float('nan'), device=dev)
    K.sumsq_partials(x, a)
    K.sumsq_partials(x, b)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.isfinite(a).all()
    ref = x.double().pow(2).sum().item()
    assert abs(a.double().sum().item() - ref) <= 2e-6 * ref
