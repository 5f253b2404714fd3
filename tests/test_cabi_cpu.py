"""CPU-side checks of the C-ABI boundary: the library builds for gfx950, loads, and exports every
symbol that include/vividmed_hip.h declares. No kernel is launched here."""
import ctypes
import subprocess

import pytest

from mmmm_amd import hip


@pytest.fixture(scope='module')
def library():
    from mmmm_amd import build
    build.build(verbose=False)
    return hip.lib()


def test_header_declares_entry_points():
    syms = hip.declared_symbols()
    assert 'vm_gemm_bf16' in syms and 'vm_attn_fwd_bf16' in syms and 'vm_expert_index_build' in syms
    assert len(syms) >= 30


def test_library_exports_every_declared_symbol(library):
    for name in hip.declared_symbols():
        assert hasattr(library, name), name


def test_library_is_gfx950_code_object():
    out = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-objdump', '--offloading', str(hip.LIB_PATH)],
                         capture_output=True, text=True).stdout
    assert 'gfx950' in out


def test_struct_layouts_match_header_sizes(library):
    # the ctypes mirrors must be layout-compatible with the C structs (x86-64 SysV)
    assert ctypes.sizeof(hip.GemmArgs) % 8 == 0
    assert hip.GemmArgs.lda.offset == 8 and hip.GemmArgs.B.offset == 16
    assert ctypes.sizeof(hip.AttnArgs) % 8 == 0
    assert ctypes.sizeof(hip.AttnF32Args) == 216 and hip.AttnF32Args.f32_split.offset == 200 and hip.AttnF32Args.delta.offset == 192
    assert hip.AttnF32Args.causal.offset == 204 and hip.AttnF32Args.row_of_pos.offset == 208


def test_version(library):
    assert library.vm_version() >= 100


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(hip, '_lib', None)
    monkeypatch.setattr(hip, 'LIB_PATH', tmp_path / 'nope.so')
    with pytest.raises(hip.HipExtensionMissing):
        hip.lib()


def test_cpu_tensor_is_rejected():
    import torch
    with pytest.raises(hip.HipError):
        hip.ptr(torch.zeros(4))
