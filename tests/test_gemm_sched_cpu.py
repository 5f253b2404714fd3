"""The bf16 GEMM's scheduler is host code (csrc/gemm.hip: sched_plan) and runs without a GPU: no chip-filling shape of BASELINE.json's six
workloads may land on the 128 x 128 tile kernel (round 4's rule sent the decoder's N = 4096 linears at 4128 / 4176 rows there), the
192-row form is chosen where it shortens the last round, and stream-K is never chosen by default (measured slower: DESIGN.md section 3)."""
import ctypes as C

import pytest


@pytest.fixture(scope='module')
def plan():
    from mmmm_amd import hip
    f = hip.lib().vm_gemm_plan_
    f.argtypes = [C.c_int] * 6 + [C.c_void_p] * 2
    f.restype = C.c_int

    def run(M, N, K, K2=64, seg=0, ws=0):
        kind, rows = C.c_int(), C.c_int()
        assert f(M, N, K, K2, seg, ws, C.addressof(kind), C.addressof(rows)) == 0
        return kind.value, rows.value
    return run


# decoder rows of the six workloads: phase-vg / vlm-448 8 x 456, phase-grg-3d 8 x 516, model-hr-2d 4 x 1044, model-hr-3d 4 x 836,
# phase-vlm-mixed 4 x 456 + 4 x 516 (+ text 128..512); ViT-E rows 8 x 785, 8 x 2049, 4 x 3137, 4 x 4609
DEC = (3648, 4128, 4176, 3344, 3888, 4500)
VIT = (6280, 16392, 12548, 18436, 4 * 785 + 4 * 2049)


def test_no_chip_filling_shape_on_the_small_tile_kernel(plan):
    for M in DEC:
        for N, K in ((4096, 4096), (4096, 11008), (11008, 4096), (12288, 4096), (4096, 12288)):
            kind, rows = plan(M, N, K, 64, 1)
            t256 = (-(-M // 256) + 1) * -(-N // 256)
            assert t256 < 200 or (kind == 1 and rows in (192, 256)) or (kind, rows) == (3, 256), (M, N, K, kind, rows)
    for M in VIT:
        for N, K in ((5376, 1792), (1792, 1792), (15360, 1792), (1792, 15360), (1792, 5376)):
            kind, rows = plan(M, N, K, 64, 0)
            assert (kind == 1 and rows in (192, 256)) or (kind, rows) == (3, 256), (M, N, K, kind, rows)      # (16 392 = 64 x 256 + 8 rows: full tiles + tails)


def test_known_choices(plan):
    assert plan(3648, 4096, 4096, 64, 1) == (1, 256)          # 16 x 16 = 256 tiles: one round of 256-row tiles
    assert plan(4128, 4096, 4096, 64, 1) == (3, 256)          # 288 tiles of 256 rows, 32 of them nearly empty: ONE round of full tiles + a tails launch
    assert plan(4176, 12288, 4096, 64, 1) == (3, 256)         # 16 x 48 = 768 full tiles = three rounds exactly, tails aside
    assert plan(3648, 11008, 4096, 64, 1)[0] == 1             # 688 tiles: three rounds with or without the partial tiles -> one launch
    assert plan(6280, 1792, 15360, 64, 0) == (1, 192)         # 175 tiles of 256 rows leave 81 CUs idle; 231 of 192 rows
    assert plan(6280, 15360, 1792, 64, 0) == (1, 256)         # 1 500 tiles: 6 rounds either way, the 256-row body is the faster one
    assert plan(456, 4096, 4096, 64, 1)[0] in (0, 1)           # a single sample: too few tiles for the cost model to matter
    assert plan(64, 512, 64, 0, 0) == (0, 0)                   # one K-tile: the small kernel


def test_stream_k_is_off_unless_forced(plan):
    from mmmm_amd import hip
    lib = hip.lib()
    assert plan(4128, 4096, 4096, 64, 1, ws=1)[0] in (1, 3)
    lib.vm_gemm_sched_mode_(2)
    try:
        assert plan(4128, 4096, 4096, 64, 1, ws=1) == (2, 256)
        assert plan(4128, 4096, 4096, 64, 1, ws=0)[0] in (1, 3)     # no workspace, no stream-K
        assert plan(3648, 12288, 4096, 64, 1, ws=1)[0] == 1    # 768 tiles: whole rounds, nothing to stream
    finally:
        lib.vm_gemm_sched_mode_(0)


def test_full_plus_tails_can_be_switched_off(plan):
    from mmmm_amd import hip
    lib = hip.lib()
    assert plan(4128, 4096, 4096, 64, 1) == (3, 256)
    assert lib.vm_gemm_tails_mode_(0) == 0
    try:
        assert plan(4128, 4096, 4096, 64, 1) == (1, 192)      # the one-launch choice of the cost model
    finally:
        lib.vm_gemm_tails_mode_(1)
