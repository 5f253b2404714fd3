"""The four-wave form of the bf16 GEMM (csrc/gemm256w.hpp: one wave per SIMD, 128 x 128 wave tiles, persistent over the tile list) against the
eight-wave form (csrc/gemm256.hip: gemm256_k). The step runs it on the launches the scheduler gives 192-row tiles and on the 256-row launches without a scale / mask on the LoRA extension (`vm_gemm_w4_mode_(3)`, the default: -2.5 % per step;
profiles/r6_gemm_w4.txt); these tests select it for EVERY eligible launch through `vm_gemm_w4_mode_(1)` and compare with mode 0. Both forms add the K-tiles of an output element in the same order (LoRA extension tiles, then the main tiles, k ascending), so every
comparison is BIT FOR BIT: plain products, the LoRA K-extension with and without the input-gradient dropout mask, bias / residual epilogues, two-expert
row segments from device counts, ragged rows and columns, 256- and 192-row tiles, one tile per workgroup and several (the seam of the persistent loop:
the next tile's first K-tiles are requested before the epilogue and the epilogue's stores are never waited for), and repeated launches (determinism)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def K():
    from mmmm_amd import kernels
    return kernels


class w4:
    """context: bf16-output NT launches of the 256-column kernel run the four-wave form; `tile`: force 256- / 192-row tiles (0: the scheduler's choice)"""

    def __init__(self, tile: int = 0):
        self.tile = tile

    def __enter__(self):
        from mmmm_amd import hip
        assert hip.lib().vm_gemm_w4_mode_(1) == 0 and hip.lib().vm_gemm_force_tile_(self.tile) == 0
        return self

    def __exit__(self, *exc):
        from mmmm_amd import hip, kernels
        hip.lib().vm_gemm_w4_mode_(kernels.GEMM_W4)          # back to the product's setting (kernels.gemm only touches the mode when its own constant changes)
        hip.lib().vm_gemm_force_tile_(0)


class w8(w4):
    def __enter__(self):
        from mmmm_amd import hip
        assert hip.lib().vm_gemm_w4_mode_(0) == 0 and hip.lib().vm_gemm_force_tile_(self.tile) == 0
        return self


def _operands(dev, M, N, Kd, K2, seed):
    g = torch.Generator(device='cpu').manual_seed(seed)
    o = dict(a=torch.randn(M, Kd, generator=g).to(dev).bfloat16(),
             w=(torch.randn(N, Kd, generator=g) / math.sqrt(Kd)).to(dev).bfloat16(),
             w1=(torch.randn(N, Kd, generator=g) / math.sqrt(Kd)).to(dev).bfloat16(),
             bias=torch.randn(N, generator=g).to(dev).bfloat16(), res=torch.randn(M, N, generator=g).to(dev).bfloat16())
    if K2:
        o.update(a2=torch.randn(M, K2, generator=g).to(dev).bfloat16(), b2=(torch.randn(N, K2, generator=g) * 0.05).to(dev).bfloat16(),
                 b21=(torch.randn(N, K2, generator=g) * 0.05).to(dev).bfloat16())
    return o


def _both(K, tile, call):
    with w8(tile):
        ref = call().clone()
    with w4(tile):
        got = call().clone()
        again = call().clone()
    torch.cuda.synchronize()
    return ref, got, again


SHAPES = [  # M, N, K: several tiles per CU (the seam), one tile per CU, ragged rows and columns, K of two tiles (no third K-tile to prefetch)
    (6280, 5376, 1792), (3648, 4096, 4096), (1000, 1000, 512), (2049, 3000, 1024), (513, 264, 128), (8200, 1792, 256),
]


@pytest.mark.parametrize('tile', [256, 192])
@pytest.mark.parametrize('M,N,Kd', SHAPES)
def test_w4_plain_and_epilogues_bit_identical(dev, K, tile, M, N, Kd):
    o = _operands(dev, M, N, Kd, 0, seed=M + N)
    plain = None
    for kw in (dict(), dict(bias=o['bias']), dict(residual=o['res']), dict(bias=o['bias'], residual=o['res'])):
        ref, got, again = _both(K, tile, lambda: K.gemm(o['a'], o['w'], **kw))
        assert torch.equal(ref, got), (tile, M, N, Kd, sorted(kw), int((ref != got).sum()))
        assert torch.equal(got, again)
        plain = got if plain is None else plain
    # and against fp32 on the host of a corner (the eight-wave form is the reference of record; this keeps the pair from being wrong together)
    r = (o['a'][:64].float() @ o['w'][:64].float().t()).cpu()
    assert (plain[:64, :64].float().cpu() - r).abs().max() <= 2e-2 * r.abs().max()


@pytest.mark.parametrize('tile', [256, 192])
@pytest.mark.parametrize('drop_p', [0.0, 0.05])
def test_w4_lora_extension_and_dropout_mask_bit_identical(dev, K, tile, drop_p):
    """the K-extension tiles (t = x A^T times B^T, conf/lora.yaml r = 64) run first, then the scale + the input-gradient dropout mask on the accumulators,
    then the main tiles: the path every LoRA linear's forward (drop_p = 0: the mask sits in lora_down) and input gradient (drop_p = 0.05) takes"""
    for (M, N, Kd) in ((6280, 1792, 5376), (3648, 4096, 4096), (1000, 520, 256)):
        o = _operands(dev, M, N, Kd, 64, seed=7 * M + N)
        ref, got, again = _both(K, tile, lambda: K.gemm(o['a'], o['w'], a2=o['a2'], b2=o['b2'], alpha2=1.0 if drop_p == 0 else 0.75, drop_p=drop_p, drop_seed=1234,
                                                        residual=o['res']))
        assert torch.equal(ref, got), (tile, M, N, Kd, drop_p, int((ref != got).sum()))
        assert torch.equal(got, again)


@pytest.mark.parametrize('tile', [256, 192])
@pytest.mark.parametrize('split', [0, 51, 2048, 3648])
def test_w4_two_expert_segments_bit_identical(dev, K, tile, split):
    """token-type gated experts: rows [0, counts[0]) use (w, b2, bias), rows [counts[0], counts[1]) use (w1, b21, bias1); counts on the device"""
    M, N, Kd = 3648, 4096, 4096
    o = _operands(dev, M, N, Kd, 64, seed=split + 1)
    counts = torch.tensor([split, M - 13, 0, 0], dtype=torch.int32, device=dev)
    ref, got, again = _both(K, tile, lambda: K.gemm(o['a'], o['w'], w1=o['w1'], a2=o['a2'], b2=o['b2'], b2_1=o['b21'], bias=o['bias'], bias1=o['bias'].flip(0).contiguous(),
                                                    counts=counts))
    v = M - 13
    assert torch.equal(ref[:v], got[:v]) and torch.equal(got[:v], again[:v]), (tile, split, int((ref[:v] != got[:v]).sum()))
