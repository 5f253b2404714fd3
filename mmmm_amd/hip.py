"""ctypes binding of the C ABI in include/vividmed_hip.h (libvividmed_hip.so).

This is the only place that touches the shared library. There is NO CPU fallback: importing the
symbols works anywhere (so the CPU test-suite can check that every declared entry point is exported),
but calling a kernel needs a gfx950 device, and a missing library raises immediately.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
import re

import torch

LIB_PATH = Path(os.environ.get('VM_LIB_PATH') or Path(__file__).resolve().parent / 'lib' / 'libvividmed_hip.so')   # VM_LIB_PATH: A/B builds
HEADER_PATH = Path(__file__).resolve().parents[1] / 'include' / 'vividmed_hip.h'

VM_BF16, VM_F32 = 0, 1
ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2
PROF_GEMM_BF16, PROF_GEMM_F32, PROF_ATTN, PROF_LORA = 0, 1, 2, 3


class HipExtensionMissing(RuntimeError):
    pass


class GemmArgs(C.Structure):
    _fields_ = [
        ('A', C.c_void_p), ('lda', C.c_int64),
        ('B', C.c_void_p), ('B_1', C.c_void_p), ('ldb', C.c_int64),
        ('A2', C.c_void_p), ('lda2', C.c_int64),
        ('B2', C.c_void_p), ('B2_1', C.c_void_p), ('ldb2', C.c_int64),
        ('K2', C.c_int32), ('alpha2', C.c_float),
        ('bias', C.c_void_p), ('bias_1', C.c_void_p),
        ('residual', C.c_void_p), ('ldr', C.c_int64),
        ('C', C.c_void_p), ('ldc', C.c_int64),
        ('M', C.c_int32), ('N', C.c_int32), ('K', C.c_int32),
        ('counts_dev', C.c_void_p),
        ('split', C.c_int32),
        ('act', C.c_int32),
        ('out_dtype', C.c_int32),
        ('drop_p', C.c_float), ('drop_seed', C.c_uint64),
        ('alpha', C.c_float),
        ('ksplit', C.c_int32),
        ('b_nn', C.c_int32),
        ('f32_split', C.c_int32),
        ('workspace', C.c_void_p), ('workspace_bytes', C.c_int64),
    ]


class AttnArgs(C.Structure):
    _fields_ = [
        ('q', C.c_void_p), ('k', C.c_void_p), ('v', C.c_void_p), ('out', C.c_void_p),
        ('ldq', C.c_int64), ('ldk', C.c_int64), ('ldv', C.c_int64), ('ldo', C.c_int64),
        ('lse', C.c_void_p),
        ('cu_seqlens', C.c_void_p), ('n_seq', C.c_int32),
        ('row_of_pos', C.c_void_p),
        ('total_pos_max', C.c_int32),
        ('max_seqlen', C.c_int32),
        ('n_heads', C.c_int32), ('head_dim', C.c_int32),
        ('scale', C.c_float),
        ('causal', C.c_int32),
        ('dout', C.c_void_p), ('lddo', C.c_int64),
        ('dq', C.c_void_p), ('dk', C.c_void_p), ('dv', C.c_void_p),
        ('lddq', C.c_int64), ('lddk', C.c_int64), ('lddv', C.c_int64),
        ('delta', C.c_void_p),
        ('workspace', C.c_void_p), ('workspace_bytes', C.c_int64),
    ]


class AttnF32Args(C.Structure):
    _fields_ = [
        ('q', C.c_void_p), ('k', C.c_void_p), ('v', C.c_void_p), ('out', C.c_void_p),
        ('q_bs', C.c_int64), ('q_ls', C.c_int64), ('k_bs', C.c_int64), ('k_ls', C.c_int64),
        ('v_bs', C.c_int64), ('v_ls', C.c_int64), ('o_bs', C.c_int64), ('o_ls', C.c_int64),
        ('lse', C.c_void_p),
        ('Bn', C.c_int32), ('Lq', C.c_int32), ('Lk', C.c_int32), ('n_heads', C.c_int32), ('head_dim', C.c_int32),
        ('scale', C.c_float),
        ('cu_seqlens', C.c_void_p), ('n_seq', C.c_int32),
        ('dout', C.c_void_p), ('do_bs', C.c_int64), ('do_ls', C.c_int64),
        ('dq', C.c_void_p), ('dk', C.c_void_p), ('dv', C.c_void_p),
        ('delta', C.c_void_p),
        ('f32_split', C.c_int32),
        ('causal', C.c_int32), ('row_of_pos', C.c_void_p),
    ]


TN_GROUP_MAX = 32


class TnGroupItem(C.Structure):
    """vm_tn_group_item (include/vividmed_hip.h)"""
    _fields_ = [
        ('W', C.c_void_p), ('ldw', C.c_int64), ('C', C.c_int32), ('M', C.c_int32),
        ('S', C.c_void_p), ('lds', C.c_int64),
        ('out', C.c_void_p), ('ldo', C.c_int64), ('out_f32', C.c_int32), ('transpose_out', C.c_int32),
        ('counts_dev', C.c_void_p), ('segment', C.c_int32), ('block0', C.c_int32),
        ('alpha', C.c_float), ('drop_p', C.c_float), ('seed', C.c_uint64),
    ]


def declared_symbols() -> list[str]:
    """Every function name declared in include/vividmed_hip.h."""
    text = HEADER_PATH.read_text()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\bint\s+(vm_\w+)\s*\(', text)))


def _prototypes() -> dict[str, list]:
    """argtypes for every entry point, parsed from the header (the header is the single source of truth)."""
    text = HEADER_PATH.read_text()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    protos = {}
    for m in re.finditer(r'\bint\s+(vm_\w+)\s*\(([^;{]*?)\)\s*;', text, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        argtypes = []
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                if '*' in a:
                    argtypes.append(C.c_void_p)
                elif 'uint64_t' in a:
                    argtypes.append(C.c_uint64)
                elif 'int64_t' in a:
                    argtypes.append(C.c_int64)
                elif 'float' in a:
                    argtypes.append(C.c_float)
                elif 'double' in a:
                    argtypes.append(C.c_double)
                else:
                    argtypes.append(C.c_int)
        protos[name] = argtypes
    return protos


_lib: C.CDLL | None = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise HipExtensionMissing(
                f'{LIB_PATH} is missing: build it with `python -m mmmm_amd.build` '
                '(there is no CPU fallback for the VividMed hot path)'
            )
        _lib = C.CDLL(str(LIB_PATH))
        for name, argtypes in _prototypes().items():
            try:
                fn = getattr(_lib, name)
            except AttributeError:
                if os.environ.get('VM_LIB_PATH'):      # an older A/B build (tools/build_ref_lib.sh) may lack the newest entry points
                    continue
                raise
            fn.restype = C.c_int
            fn.argtypes = argtypes
    return _lib


class HipError(RuntimeError):
    pass


_ERR = {-1: 'VM_ERR_BAD_ARG', -2: 'VM_ERR_UNSUPPORTED', -3: 'VM_ERR_LAUNCH'}


def check(rc: int, what: str):
    if rc != 0:
        raise HipError(f'{what} failed: {_ERR.get(rc, rc)}')


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.bfloat16:
        return VM_BF16
    if dt == torch.float32:
        return VM_F32
    raise TypeError(f'unsupported dtype {dt}')


def ptr(t: torch.Tensor | None) -> int | None:
    if t is None:
        return None
    if not t.is_cuda:
        raise HipError('the VividMed HIP path needs device tensors (no CPU fallback)')
    return t.data_ptr()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream() -> int:
    """hipStream_t of torch's current stream on the current device. The raw accessor avoids building a torch.cuda.Stream
    object per launch (9 us of host time each, ~5k launches per training step)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


_STREAM_OBJ: dict = {}


def current_stream_obj(device=None) -> 'torch.cuda.Stream':
    """torch's current stream on `device` as a torch.cuda.Stream — the SAME object for the same stream every time, so callers may compare
    with `is` and skip set insertions. `torch.cuda.current_stream()` builds a new wrapper per call (~4 us) and its __eq__ / __hash__ are
    Python: the backward pass asked ~5k times per step (22 ms of host time under cProfile, profiles/r5_host_profile_v2.txt)."""
    idx = device.index if (device is not None and device.index is not None) else torch.cuda.current_device()
    if _raw_stream is None:
        return torch.cuda.current_stream(idx)
    key = (idx, _raw_stream(idx))
    st = _STREAM_OBJ.get(key)
    if st is None:
        st = _STREAM_OBJ[key] = torch.cuda.current_stream(idx)
    return st


_fns: dict = {}


def call(name: str, *args):
    """Call a C-ABI entry point with raw arguments and raise on a non-zero status."""
    fn = _fns.get(name)
    if fn is None:
        fn = _fns[name] = getattr(lib(), name)
    rc = fn(*args)
    if rc != 0:
        check(rc, name)
