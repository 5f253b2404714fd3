"""Algebra behind the two forms of the LoRA dropout mask (mmmm_amd/csrc/vm_common.hpp), restated in numpy — the device code itself is
compared form against form on the GPU (tests/test_kernels_gpu.py: every masked kernel against `K.dropout`, which uses the 64-bit form).

* vm_hash4(seed, group) folds the high word of the group index into the low one; for group < 2^32 the fold is the identity, so
  vm_hash4w(vm_seed(seed), (uint32) group) — what the kernels use below 2^34 elements — is the same function.
* vm_keep_mask2 decides `field >= thr` for two 16-bit fields with three packed instructions on (field >> 1): exact for every field value
  when thr is even, which vm_drop_threshold guarantees."""
import numpy as np


def _fold(group: np.ndarray) -> np.ndarray:
    hi = (group >> np.uint64(32)).astype(np.uint32)
    lo = (group & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    with np.errstate(over='ignore'):
        return lo ^ (hi << np.uint32(17)) ^ (hi >> np.uint32(3)) ^ (hi * np.uint32(0x9E3779B9))


def test_high_word_fold_is_identity_below_2_to_32():
    rng = np.random.default_rng(0)
    g = rng.integers(0, 2 ** 32, size=100_000, dtype=np.uint64)
    assert np.array_equal(_fold(g), g.astype(np.uint32))
    g_hi = g + (np.uint64(1) << np.uint64(32))
    assert not np.array_equal(_fold(g_hi), g.astype(np.uint32))      # above it the fold does change the key


def test_packed_mask_equals_field_ge_threshold_for_even_thresholds():
    f = np.arange(65536, dtype=np.int64)
    for p in (0.0, 0.05, 0.1, 0.25, 0.5, 0.9):
        thr = int(np.float32(p) * np.float32(65536.0)) & ~1          # vm_drop_threshold
        c = (thr >> 1) - 1                                           # the splatted int16 constant
        d = (c - (f >> 1)).astype(np.int16)                          # v_pk_sub_i16 (both sides < 2^15: no wrap)
        assert np.all(c - (f >> 1) == d), 'the 16-bit subtraction must not wrap'
        keep_packed = d < 0                                          # v_pk_ashrrev_i16 15 -> 0xFFFF where negative
        assert np.array_equal(keep_packed, f >= thr), p


def test_threshold_of_the_benchmarked_dropout_is_unchanged():
    assert int(np.float32(0.05) * np.float32(65536.0)) == 3276       # already even: the masks of lora_dropout 0.05 did not move
