"""Tensor-level wrappers of the C ABI (one Python function per entry point, no autograd).

Every function enqueues on the current torch stream and returns immediately; there is no CPU path.
`mmmm_amd.functional` builds the autograd operators of the model on top of these.
"""
from __future__ import annotations

import os

import ctypes as C
import functools
import math

import torch

from . import hip
from .hip import ptr, dtype_code, stream


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


def _ld(t: torch.Tensor) -> int:
    """leading dimension (elements) of a 2-D row-major view with unit inner stride"""
    st = t.stride()              # (one call: this runs ~13k times per step)
    assert len(st) == 2 and st[1] == 1, (t.shape, st)
    return st[0] if t.shape[0] > 1 else max(st[0], t.shape[1])


# ------------------------------------------------------------------ GEMM
_GEMM_WS: dict = {}
GEMM_TAILS = True            # the scheduler may run a launch as full 256-row tiles + a tails launch (vm_gemm_tails_mode_); A/B: bench.py --set kernels.GEMM_TAILS=False
_gemm_tails_applied = [True]
GEMM_W4 = 3                  # which launches of the 256-column kernel run the four-wave persistent form (vm_gemm_w4_mode_): 0 none, 1 all bf16-output NT launches, 2 those the scheduler runs on
                             # 192-row tiles, 3 (default) those + the 256-row launches whose LoRA extension needs no scale / dropout mask (the forward launches) — inside the step, one call:
                             # 301.0 (0) / 295.6 (2) / 293.4 (3) ms; bit-identical results (profiles/r6_gemm_w4.txt); A/B: bench.py --set kernels.GEMM_W4=0
_gemm_w4_applied = [3]       # (the library's own default)


def gemm_workspace() -> torch.Tensor:
    """Scratch of the stream-K form of the bf16 / fp8 GEMM (vm_gemm_args.workspace): one zero-filled buffer per (device, stream) — launches
    on one stream are ordered, launches on different streams may overlap and must not share slabs. 64 MiB each on MI355X."""
    key = (torch.cuda.current_device(), stream())
    ws = _GEMM_WS.get(key)
    if ws is None:
        need = C.c_int64(0)
        hip.call('vm_gemm_workspace_bytes', C.addressof(need))
        ws = _GEMM_WS[key] = torch.zeros(need.value, dtype=torch.uint8, device='cuda')
    return ws


def gemm(
    a: torch.Tensor, w: torch.Tensor, *,
    w1: torch.Tensor | None = None,
    a2: torch.Tensor | None = None, b2: torch.Tensor | None = None, b2_1: torch.Tensor | None = None,
    alpha2: float = 1.0,
    bias: torch.Tensor | None = None, bias1: torch.Tensor | None = None,
    residual: torch.Tensor | None = None,
    out: torch.Tensor | None = None, out_dtype: torch.dtype | None = None,
    act: int = hip.ACT_NONE,
    counts: torch.Tensor | None = None, split: int = -1,
    drop_p: float = 0.0, drop_seed: int = 0, out_is_zero: bool = False, b_nn: bool = False,
    f32_split: int = 0, accumulate: bool = False, ksplit_override: int | None = None, workspace: torch.Tensor | None = None,
) -> torch.Tensor:
    """out[M,N] = act(a[M,K] @ w[N,K]^T + alpha2 * a2[M,K2] @ b2[N,K2]^T + bias) + residual

    `w1`/`b2_1`/`bias1`: weights of the second row segment (token-type gated experts); the segment
    boundary is `split` (host) or `counts[0]` with `counts[1]` valid rows (device int32 tensor).
    `accumulate` (with an fp32 `out`): out += product — through the split-K atomics when the output is a handful of tiles with a
    long contraction, else through the residual path of the epilogue.
    `f32_split` (fp32 operands): arithmetic of this call — 0 process default, 1 exact f32 MFMA, 2 / 3 split-bf16 with 3 / 6 products.
    `workspace` (bf16): `gemm_workspace()` — lets the library run the stream-K form where its scheduler asks for it (off by default: measured
    slower on every shape of the six workloads; `tests/test_gemm_sched_gpu.py` forces it).
    """
    if GEMM_TAILS != _gemm_tails_applied[0]:
        hip.call('vm_gemm_tails_mode_', int(bool(GEMM_TAILS)))
        _gemm_tails_applied[0] = GEMM_TAILS
    if GEMM_W4 != _gemm_w4_applied[0]:
        hip.call('vm_gemm_w4_mode_', int(GEMM_W4))
        _gemm_w4_applied[0] = GEMM_W4
    if b_nn:      # `w` is [K, N]: the contraction index is its row (a weight as stored, for dx = dy W); bf16, 256-column kernel only
        assert a.dim() == 2 and w.dim() == 2 and a.shape[1] == w.shape[0] and a.dtype == torch.bfloat16 and out is None, (a.shape, w.shape)
        M, K = a.shape
        N = w.shape[1]
    else:
        assert a.dim() == 2 and w.dim() == 2 and a.shape[1] == w.shape[1], (a.shape, w.shape)
        M, K = a.shape
        N = w.shape[0]
    f32 = a.dtype == torch.float32
    assert w.dtype == a.dtype
    if out_dtype is None:
        out_dtype = a.dtype
    # tiny M x N with a long contraction (weight gradients of the hyper-network mask products): one tile would walk
    # all of K alone, so K is split over workgroups that accumulate into a zeroed fp32 C
    ksplit = 0
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    if out is not None:
        out_dtype = out.dtype
    # (`out_is_zero`: the caller passes a ZEROED fp32 `out`, which the split-K form may accumulate into)
    if accumulate:
        assert out is not None and out.dtype == torch.float32 and residual is None
    if ((out is None or out_is_zero or accumulate) and out_dtype == torch.float32 and act == hip.ACT_NONE and a2 is None and counts is None and split < 0
            and tiles <= 48 and K >= 1024):
        # a handful of output tiles with a long contraction (weight gradients of the fp32 grounding heads). Only for small
        # outputs: the fp32 atomic epilogue runs at ~60 G atomics/s and already costs more than it saves at 150 tiles.
        # One workgroup walking all of K alone costs ~38 ns per K element (fp32 operands; ~7 ns bf16); S splits divide that by S but
        # add M*N*S fp32 atomics at ~60 G/s: the optimum is S = sqrt(t_K / t_atomics) (measured: [384 x 768 x 12544] 457 us
        # unsplit, 144 us with 23 splits, the model's 10 splits ~95 us; [768 x 768 x 3136] 118 us unsplit = 115 us with 11 splits)
        best = math.sqrt((2256.0 if f32 else 400.0) * K / (M * N))
        ksplit = max(1, min(64, K // (4 * (32 if f32 else 64)), -(-512 // tiles), int(best + 0.5)))
    if ksplit_override is not None:      # tools/bench_gemm_f32.py sweeps the split count
        ksplit = ksplit_override
    if accumulate and ksplit <= 1:
        residual = out
    if out is None:
        al = 8 if out_dtype == torch.bfloat16 else 4
        Np = (N + al - 1) // al * al                 # keep ldc aligned for the vector stores of the epilogue
        out = (torch.zeros if ksplit > 1 else torch.empty)(M, Np, dtype=out_dtype, device=a.device)
        if Np != N:
            out = out[:, :N]
    g = hip.GemmArgs()
    g.A, g.lda = ptr(a), _ld(a)
    g.B, g.B_1, g.ldb = ptr(w), ptr(w1), _ld(w)
    if w1 is not None:
        assert w1.shape == w.shape and _ld(w1) == _ld(w)
    if a2 is not None:
        assert b2 is not None and a2.shape[0] == M and b2.shape == (N, a2.shape[1])
        g.A2, g.lda2 = ptr(a2), _ld(a2)
        g.B2, g.B2_1, g.ldb2 = ptr(b2), ptr(b2_1), _ld(b2)
        g.K2 = a2.shape[1]
    else:
        g.K2 = 0
    g.alpha2 = alpha2
    if bias is not None:
        assert bias.dtype == out.dtype and bias.numel() == N
    g.bias, g.bias_1 = ptr(bias), ptr(bias1)
    if residual is not None:
        assert residual.dtype == out.dtype and residual.shape == out.shape
        g.residual, g.ldr = ptr(residual), _ld(residual)
    g.C, g.ldc = ptr(out), _ld(out)
    g.M, g.N, g.K = M, N, K
    g.counts_dev = ptr(counts)
    g.split = split
    g.act = act
    g.out_dtype = dtype_code(out.dtype)
    g.drop_p, g.drop_seed = drop_p, drop_seed & 0xFFFFFFFFFFFFFFFF
    g.alpha = 1.0
    g.ksplit = ksplit
    g.b_nn = 1 if b_nn else 0
    g.f32_split = f32_split if f32 else 0
    if workspace is not None:
        g.workspace, g.workspace_bytes = ptr(workspace), workspace.numel()
    hip.call('vm_gemm_f32' if f32 else 'vm_gemm_bf16', C.addressof(g), stream())
    return out


def quant_rows_fp8(x: torch.Tensor, nrows: torch.Tensor | None = None, need_inv: bool = True):
    """per-row e4m3 quantisation: -> (x8 uint8 [R, C], scale fp32 [R], inv_scale fp32 [R] | None)"""
    assert x.dim() == 2 and x.stride(1) == 1 and x.shape[1] % 16 == 0
    R, Cc = x.shape
    x8 = torch.empty(R, Cc, dtype=torch.uint8, device=x.device)
    scale = torch.empty(R, dtype=torch.float32, device=x.device)
    inv = torch.empty(R, dtype=torch.float32, device=x.device) if need_inv else None
    hip.call('vm_quant_rows_fp8', ptr(x), x.stride(0), ptr(x8), x8.stride(0), ptr(scale), ptr(inv), R, Cc, dtype_code(x.dtype), ptr(nrows), stream())
    return x8, scale, inv


def scale_rows(x: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    """bf16 [R, C] times an fp32 factor per row -> bf16"""
    assert x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1 and s.dtype == torch.float32 and s.numel() >= x.shape[0]
    out = torch.empty_like(x)
    hip.call('vm_scale_rows_bf16', ptr(x), x.stride(0), ptr(s), ptr(out), out.stride(0), x.shape[0], x.shape[1], stream())
    return out


def gemm_fp8(a8: torch.Tensor, sa: torch.Tensor, w8: torch.Tensor, sw: torch.Tensor, *, w1_8: torch.Tensor | None = None,
             sw1: torch.Tensor | None = None, a2: torch.Tensor | None = None, b2: torch.Tensor | None = None, b2_1: torch.Tensor | None = None,
             alpha2: float = 1.0, bias: torch.Tensor | None = None, bias1: torch.Tensor | None = None, residual: torch.Tensor | None = None,
             out_dtype: torch.dtype = torch.bfloat16, counts: torch.Tensor | None = None, drop_p: float = 0.0, drop_seed: int = 0,
             workspace: torch.Tensor | None = None) -> torch.Tensor:
    """out[M, N] = sa[m] sw[n] (a8 @ w8^T) + alpha2 (a2 @ b2^T) + bias (+ residual) with e4m3 a8 / w8 (vm_gemm_fp8). a2 / b2 must already
    be divided by sa / sw (functional._Linear does that)."""
    M, Kd = a8.shape
    N = w8.shape[0]
    assert a8.dtype == torch.uint8 and w8.dtype == torch.uint8 and w8.shape[1] == Kd and Kd % 128 == 0
    assert sa.dtype == torch.float32 and sa.numel() == M and sw.dtype == torch.float32 and sw.numel() == N
    al = 8 if out_dtype == torch.bfloat16 else 4
    Np = (N + al - 1) // al * al
    out = torch.empty(M, Np, dtype=out_dtype, device=a8.device)
    if Np != N:
        out = out[:, :N]
    g = hip.GemmArgs()
    g.A, g.lda = ptr(a8), a8.stride(0)
    g.B, g.B_1, g.ldb = ptr(w8), ptr(w1_8), w8.stride(0)
    if a2 is not None:
        assert b2 is not None and a2.shape[0] == M and b2.shape == (N, a2.shape[1]) and a2.dtype == torch.bfloat16
        g.A2, g.lda2 = ptr(a2), _ld(a2)
        g.B2, g.B2_1, g.ldb2 = ptr(b2), ptr(b2_1), _ld(b2)
        g.K2 = a2.shape[1]
    else:
        g.K2 = 0
    g.alpha2 = alpha2
    g.bias, g.bias_1 = ptr(bias), ptr(bias1)
    if residual is not None:
        assert residual.dtype == out.dtype and residual.shape == out.shape
        g.residual, g.ldr = ptr(residual), _ld(residual)
    g.C, g.ldc = ptr(out), _ld(out)
    g.M, g.N, g.K = M, N, Kd
    g.counts_dev = ptr(counts)
    g.split, g.act = -1, hip.ACT_NONE
    g.out_dtype = dtype_code(out.dtype)
    g.drop_p, g.drop_seed = drop_p, drop_seed & 0xFFFFFFFFFFFFFFFF
    g.alpha, g.ksplit = 1.0, 0
    if workspace is not None:
        g.workspace, g.workspace_bytes = ptr(workspace), workspace.numel()
    hip.call('vm_gemm_fp8', C.addressof(g), ptr(sa), ptr(sw), ptr(sw1), stream())
    return out


def ubench_mfma_bf16(seconds: float = 2.0) -> float:
    """sustained TFLOP/s of bare bf16 MFMAs on the current device (vm_ubench_mfma_bf16: random operands, ~`seconds`, synchronous)"""
    out = C.c_float(0.0)
    hip.call('vm_ubench_mfma_bf16', float(seconds), C.addressof(out), stream())
    return float(out.value)


_F32_MODE = [3]          # mirror of the library's process default (gemm.hip f32_mode(): six products); f32_mode() below changes both


def gemm_f32_mode(mode: int):
    """arithmetic of vm_gemm_f32 (vm_gemm_f32_mode): 0 exact f32 MFMA, 2 split-bf16 in registers with 3 products, 3 with 6 (default)"""
    hip.call('vm_gemm_f32_mode', mode)
    _F32_MODE[0] = mode


def f32_mode_resolved(f32_split: int = 0) -> int:
    """the arithmetic a call with `f32_split` gets: 0 exact, 2 / 3 split-bf16 (f32_split 0 = the process default, 1 = exact)"""
    return _F32_MODE[0] if f32_split == 0 else (0 if f32_split == 1 else f32_split)



def transpose(x: torch.Tensor, *, pad_to: int = 1, nrows: torch.Tensor | None = None, colsum_out: torch.Tensor | None = None) -> torch.Tensor:
    """out[cols, rows_padded] = x^T, zero beyond the true row count (K-contiguous operand of a wgrad GEMM). `colsum_out`: fp32 [cols]
    accumulator that additionally receives the column sums of x (atomically added)"""
    assert x.dim() == 2 and x.stride(1) == 1
    rows, cols = x.shape
    rp = (rows + pad_to - 1) // pad_to * pad_to
    out = torch.empty(cols, rp, dtype=x.dtype, device=x.device)
    if rp > (rows + 63) // 64 * 64:      # (up to the next multiple of 64 the kernel zero-fills the pad columns itself)
        out[:, rows:].zero_()
    if colsum_out is not None:
        assert nrows is None and colsum_out.dtype == torch.float32 and colsum_out.is_contiguous() and colsum_out.numel() == cols
        hip.call('vm_transpose_colsum', ptr(x), _ld(x), ptr(out), rp, rows, cols, dtype_code(x.dtype), ptr(colsum_out), stream())
    else:
        hip.call('vm_transpose', ptr(x), _ld(x), ptr(out), rp, rows, cols, dtype_code(x.dtype), ptr(nrows), stream())
    return out


def transpose_batched(desc: torch.Tensor, n: int, tiles_per_entry: int, dtype: torch.dtype):
    """desc: int64 [n, 6] on the device — {src, dst, rows, cols, ld_src, ld_dst}; one launch for the whole table"""
    assert desc.dtype == torch.int64 and desc.is_contiguous() and desc.numel() >= n * 6
    hip.call('vm_transpose_batched', ptr(desc), n, tiles_per_entry, dtype_code(dtype), stream())


def accum_f32_table(desc: torch.Tensor, n: int, blocks_per_entry: int = 4):
    """desc: int64 [n, 3] on the device — {dst bf16 ptr, src fp32 ptr, count}: dst += bf16(src), src = 0, one launch"""
    assert desc.dtype == torch.int64 and desc.is_contiguous() and desc.numel() >= n * 3
    hip.call('vm_accum_f32_table', ptr(desc), n, blocks_per_entry, stream())


def transpose_segment(x: torch.Tensor, counts: torch.Tensor, segment: int, pad_to: int = 64) -> torch.Tensor:
    """transpose of row segment `segment` (0: [0,counts[0]), 1: [counts[0],counts[1])) -> [cols, rows_padded]"""
    rows, cols = x.shape
    rp = (rows + pad_to - 1) // pad_to * pad_to
    out = torch.empty(cols, rp, dtype=x.dtype, device=x.device)
    if rp > (rows + 63) // 64 * 64:
        out[:, rows:].zero_()
    hip.call('vm_transpose_segment', ptr(x), _ld(x), ptr(out), rp, rows, cols, dtype_code(x.dtype), ptr(counts), segment, stream())
    return out


def colsum(x: torch.Tensor, nrows: torch.Tensor | None = None, out: torch.Tensor | None = None) -> torch.Tensor:
    """out[c] += sum_r x[r, c] (fp32 atomics); `out`: an existing fp32 accumulator (e.g. a bias gradient in its bucket)"""
    assert x.dim() == 2 and x.stride(1) == 1
    if out is None:
        out = torch.zeros(x.shape[1], dtype=torch.float32, device=x.device)
    assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == x.shape[1]
    hip.call('vm_colsum', ptr(x), _ld(x), ptr(out), x.shape[0], x.shape[1], dtype_code(x.dtype), ptr(nrows), stream())
    return out


LORA_DOWN_TARGET_WGS = 512          # workgroup slots a lora_down launch may fill (vm_lora_down_target_); A/B: bench.py --set kernels.LORA_DOWN_TARGET_WGS=...
_lora_down_target_applied = [512]


def lora_down(x: torch.Tensor, A0: torch.Tensor, A1: torch.Tensor | None = None, *, counts: torch.Tensor | None = None,
              split: int = -1, drop_p: float = 0.0, drop_seed: int = 0) -> torch.Tensor:
    """t[M,64] = drop(x) @ A^T (bf16; rank 64; K % 8 == 0)"""
    if LORA_DOWN_TARGET_WGS != _lora_down_target_applied[0]:
        hip.call('vm_lora_down_target_', int(LORA_DOWN_TARGET_WGS))
        _lora_down_target_applied[0] = LORA_DOWN_TARGET_WGS
        _lora_ws_bytes.cache_clear()
    M, Kd = x.shape
    t = torch.empty(M, A0.shape[0], dtype=x.dtype, device=x.device)
    segmented = counts is not None or split >= 0
    nbytes = _lora_ws_bytes(M, Kd, segmented)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device) if nbytes else None
    hip.call('vm_lora_down', ptr(x), _ld(x), ptr(A0), ptr(A1), _ld(A0), ptr(t), _ld(t), M, Kd, A0.shape[0], ptr(counts), split,
             drop_p, drop_seed & 0xFFFFFFFFFFFFFFFF, ptr(ws), nbytes, stream())
    return t


@functools.lru_cache(maxsize=256)
def _lora_ws_bytes(M: int, Kd: int, segmented: bool) -> int:
    n = C.c_int64(0)
    hip.call('vm_lora_down_workspace', M, Kd, int(segmented), C.addressof(n))
    return n.value


def lora_down_supported(x: torch.Tensor, A: torch.Tensor) -> bool:
    return x.dtype == torch.bfloat16 and A.shape[0] == 64 and x.shape[1] % 8 == 0 and x.shape[1] >= 8


def gemm_tn(X: torch.Tensor, Y: torch.Tensor, *, counts: torch.Tensor | None = None, segment: int = -1,
            nrows: torch.Tensor | None = None, alpha: float = 1.0, drop_p: float = 0.0, drop_seed: int = 0,
            out_dtype: torch.dtype | None = None) -> torch.Tensor:
    """C[P,Q] = alpha * X[M,P]^T @ drop(Y)[M,Q], contraction over rows (all rows, or one routed segment)"""
    assert X.dtype == torch.bfloat16 and Y.dtype == torch.bfloat16 and X.shape[0] == Y.shape[0]
    M, P = X.shape
    Q = Y.shape[1]
    out_dtype = out_dtype or X.dtype
    tiles = ((P + 127) // 128) * ((Q + 127) // 128)
    splits = 1
    if tiles < 512 and M >= 256:
        # row-contraction of a skinny product streams X and Y once: it wants >= 2-4 workgroups per CU in flight
        splits = max(1, min(64, 1024 // tiles, M // 64))
    if splits > 1:
        C32 = torch.zeros(P, Q, dtype=torch.float32, device=X.device)
        hip.call('vm_gemm_tn_bf16', ptr(X), _ld(X), P, ptr(Y), _ld(Y), Q, ptr(C32), Q, VM_F32_, M, ptr(counts), segment, ptr(nrows),
                 splits, alpha, drop_p, drop_seed & 0xFFFFFFFFFFFFFFFF, Y.shape[1], stream())
        return C32 if out_dtype == torch.float32 else cast(C32, out_dtype)
    Cc = torch.empty(P, Q, dtype=out_dtype, device=X.device)
    hip.call('vm_gemm_tn_bf16', ptr(X), _ld(X), P, ptr(Y), _ld(Y), Q, ptr(Cc), Q, dtype_code(out_dtype), M, ptr(counts), segment,
             ptr(nrows), 1, alpha, drop_p, drop_seed & 0xFFFFFFFFFFFFFFFF, Y.shape[1], stream())
    return Cc


VM_F32_ = hip.VM_F32


def gemm_tn_f32_supported(X: torch.Tensor, Y: torch.Tensor, out: torch.Tensor, f32_split: int = 0) -> bool:
    """(the TN kernel exists in the split-bf16 forms only: under the exact arithmetic — VM_F32_SPLIT=0, gemm_f32_mode(0), f32_split=1 —
    the caller takes the transpose + NT GEMM route)"""
    return (f32_mode_resolved(f32_split) in (2, 3) and X.dtype == torch.float32 and Y.dtype == torch.float32 and out.dtype == torch.float32 and X.dim() == 2 and Y.dim() == 2
            and X.shape[0] == Y.shape[0] and X.stride(1) == 1 and Y.stride(1) == 1 and out.stride(1) == 1
            and X.shape[1] % 8 == 0 and Y.shape[1] % 8 == 0 and X.stride(0) % 4 == 0 and Y.stride(0) % 4 == 0
            and out.shape == (X.shape[1], Y.shape[1]))


def gemm_tn_f32(X: torch.Tensor, Y: torch.Tensor, out: torch.Tensor, *, colsum_out: torch.Tensor | None = None, f32_split: int = 0):
    """out[P, Q] += X[M, P]^T @ Y[M, Q] (fp32 operands as stored: the weight gradient dy^T x without transposed copies; split-bf16
    products, f32_split 2 / 3 / 0 = default); `colsum_out` fp32 [P] += column sums of X (the bias gradient)"""
    assert gemm_tn_f32_supported(X, Y, out, f32_split)
    M, P = X.shape
    Q = Y.shape[1]
    if colsum_out is not None:
        assert colsum_out.dtype == torch.float32 and colsum_out.is_contiguous() and colsum_out.numel() == P
    hip.call('vm_gemm_tn_f32', ptr(X), _ld(X), P, ptr(Y), _ld(Y), Q, ptr(out), _ld(out), M, ptr(colsum_out), f32_split, stream())
    return out


@functools.lru_cache(maxsize=256)
def _tn_skinny_ws_bytes(M: int, Cw: int) -> int:
    n = C.c_int64(0)
    hip.call('vm_tn_skinny_workspace', M, Cw, C.addressof(n))
    return n.value


def tn_skinny_supported(W: torch.Tensor, S: torch.Tensor) -> bool:
    return (W.dtype == torch.bfloat16 and S.dtype == torch.bfloat16 and S.shape[1] == 64 and W.shape[1] % 8 == 0
            and W.stride(1) == 1 and S.stride(1) == 1 and W.stride(0) % 8 == 0 and S.stride(0) % 8 == 0)


def tn_skinny(W: torch.Tensor, S: torch.Tensor, *, transpose_out: bool, out=None, accumulate: bool = False,
              counts: torch.Tensor | None = None, segment: int = -1, nrows: torch.Tensor | None = None, alpha: float = 1.0,
              drop_p: float = 0.0, drop_seed: int = 0, out_dtype: torch.dtype | None = None):
    """LoRA factor gradient: out[c][n] (or out[n][c] when transpose_out) (+)= alpha * sum_m drop(W)[m][c] * S[m][n].
    W [M, C] is the wide streamed operand, S [M, 64]; deterministic (workspace partials reduced in a fixed order).
    segment == 2 (with `counts`): both routed row segments in one launch; `out` is then a pair (segment 0, segment 1)."""
    M, Cw = W.shape
    assert S.shape == (M, 64)
    shape = (64, Cw) if transpose_out else (Cw, 64)
    both = segment == 2
    if out is None:
        assert not accumulate
        mk = lambda: torch.empty(shape, dtype=out_dtype or W.dtype, device=W.device)
        out = (mk(), mk()) if both else mk()
    o0, o1 = out if both else (out, None)
    for o in (o0, o1):
        assert o is None or (o.shape == shape and o.stride(1) == 1)
    assert o1 is None or (o1.dtype == o0.dtype and o1.stride(0) == o0.stride(0))
    nbytes = _tn_skinny_ws_bytes(M, Cw)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=W.device)
    hip.call('vm_tn_skinny_bf16', ptr(W), _ld(W), Cw, ptr(S), _ld(S), ptr(o0), ptr(o1), _ld(o0), dtype_code(o0.dtype), int(transpose_out),
             int(accumulate), M, ptr(counts), segment, ptr(nrows), alpha, drop_p, drop_seed & 0xFFFFFFFFFFFFFFFF, ptr(ws), nbytes, stream())
    return out


def tn_skinny_group(items: list) -> None:
    """A batch of LoRA factor gradients in one launch (vm_tn_skinny_group_bf16). Each item is a tuple
    (W [M, C] bf16, S [M, 64] bf16, out, transpose_out, counts | None, segment, alpha, drop_p, drop_seed): `out` ([C, 64], or [64, C]
    when transpose_out; bf16 or fp32) is ACCUMULATED into. Deterministic (one workgroup per 64 output columns walks all rows)."""
    for i in range(0, len(items), hip.TN_GROUP_MAX):
        chunk = items[i:i + hip.TN_GROUP_MAX]
        arr = (hip.TnGroupItem * len(chunk))()
        for q, (W, S, out, transpose_out, counts, segment, alpha, drop_p, seed) in zip(arr, chunk):
            M, Cw = W.shape
            assert S.shape == (M, 64) and W.dtype == torch.bfloat16 and S.dtype == torch.bfloat16 and W.stride(1) == 1 and S.stride(1) == 1
            assert out.shape == ((64, Cw) if transpose_out else (Cw, 64)) and out.stride(1) == 1 and out.dtype in (torch.bfloat16, torch.float32)
            q.W, q.ldw, q.C, q.M = ptr(W), _ld(W), Cw, M
            q.S, q.lds = ptr(S), _ld(S)
            q.out, q.ldo, q.out_f32, q.transpose_out = ptr(out), _ld(out), int(out.dtype == torch.float32), int(transpose_out)
            q.counts_dev, q.segment = ptr(counts), segment if counts is not None else -1
            q.alpha, q.drop_p, q.seed = alpha, drop_p, seed & 0xFFFFFFFFFFFFFFFF
        hip.call('vm_tn_skinny_group_bf16', C.addressof(arr), len(chunk), stream())


# ------------------------------------------------------------------ norms
def rmsnorm_fwd(x: torch.Tensor, w: torch.Tensor, eps: float, nrows: torch.Tensor | None = None):
    x = _c(x)
    rows, cols = x.shape
    y = torch.empty_like(x)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    hip.call('vm_rmsnorm_fwd', ptr(x), ptr(w), ptr(y), ptr(rstd), rows, cols, eps, dtype_code(x.dtype), ptr(nrows), stream())
    return y, rstd


def rmsnorm_bwd(x, w, dy, rstd, nrows: torch.Tensor | None = None, need_dw: bool = True, need_dx: bool = True, dw_out=None, dx_add=None):
    """`dw_out`: an fp32 [cols] accumulator to add into (e.g. the parameter's gradient slot) instead of a zeroed scratch;
    `dx_add`: gradient of the residual branch that forked off x, summed into dx by the kernel"""
    x, dy = _c(x), _c(dy)
    rows, cols = x.shape
    dx = torch.empty_like(x) if need_dx else None
    dw = (dw_out if dw_out is not None else torch.zeros(cols, dtype=torch.float32, device=x.device)) if need_dw else None
    if dx_add is not None:
        dx_add = _c(dx_add)
        assert need_dx and dx_add.shape == x.shape and dx_add.dtype == x.dtype
    hip.call('vm_rmsnorm_bwd_res', ptr(x), ptr(w), ptr(dy), ptr(rstd), ptr(dx_add), ptr(dx), ptr(dw), rows, cols, dtype_code(x.dtype),
             ptr(nrows), stream())
    return dx, dw


def layernorm_fwd(x, w, b, eps: float, residual: torch.Tensor | None = None):
    x = _c(x)
    rows, cols = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    hip.call('vm_layernorm_fwd', ptr(x), ptr(w), ptr(b), ptr(residual), ptr(y), ptr(mean), ptr(rstd), rows, cols, eps,
             dtype_code(x.dtype), stream())
    return y, mean, rstd


def layernorm_bwd(x, w, dy, mean, rstd, need_dw: bool = True, need_dx: bool = True, dw_out=None, db_out=None, dx_add=None):
    x, dy = _c(x), _c(dy)
    rows, cols = x.shape
    dx = torch.empty_like(x) if need_dx else None
    dw = (dw_out if dw_out is not None else torch.zeros(cols, dtype=torch.float32, device=x.device)) if need_dw else None
    db = (db_out if db_out is not None else torch.zeros(cols, dtype=torch.float32, device=x.device)) if need_dw else None
    if dx_add is not None:
        dx_add = _c(dx_add)
        assert need_dx and dx_add.shape == x.shape and dx_add.dtype == x.dtype
    hip.call('vm_layernorm_bwd_res', ptr(x), ptr(w), ptr(dy), ptr(mean), ptr(rstd), ptr(dx_add), ptr(dx), ptr(dw), ptr(db), rows, cols,
             dtype_code(x.dtype), stream())
    return dx, dw, db


# ------------------------------------------------------------------ rope / activations / misc
def rope_(qkv: torch.Tensor, row_pos: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, n_heads: int, head_dim: int,
          inverse: bool = False, nrows: torch.Tensor | None = None):
    assert qkv.dim() == 2 and qkv.stride(1) == 1 and cos.dtype == torch.float32 and cos.is_contiguous()
    assert row_pos.dtype == torch.int32
    hip.call('vm_rope_inplace', ptr(qkv), _ld(qkv), ptr(row_pos), ptr(cos), ptr(sin), cos.shape[0], qkv.shape[0],
             n_heads, head_dim, dtype_code(qkv.dtype), int(inverse), ptr(nrows), stream())
    return qkv


def silu_mul(gate, up):
    gate, up = _c(gate), _c(up)
    out = torch.empty_like(gate)
    hip.call('vm_silu_mul_fwd', ptr(gate), ptr(up), ptr(out), gate.numel(), dtype_code(gate.dtype), stream())
    return out


def silu_mul_bwd(gate, up, dout):
    gate, up, dout = _c(gate), _c(up), _c(dout)
    dg, du = torch.empty_like(gate), torch.empty_like(up)
    hip.call('vm_silu_mul_bwd', ptr(gate), ptr(up), ptr(dout), ptr(dg), ptr(du), gate.numel(), dtype_code(gate.dtype), stream())
    return dg, du


def gelu(x):
    x = _c(x)
    y = torch.empty_like(x)
    hip.call('vm_gelu_fwd', ptr(x), ptr(y), x.numel(), dtype_code(x.dtype), stream())
    return y


def gelu_bwd(x, dy):
    x, dy = _c(x), _c(dy)
    dx = torch.empty_like(x)
    hip.call('vm_gelu_bwd', ptr(x), ptr(dy), ptr(dx), x.numel(), dtype_code(x.dtype), stream())
    return dx


def relu_bwd(y, dy):
    y, dy = _c(y), _c(dy)
    dx = torch.empty_like(y)
    hip.call('vm_relu_bwd', ptr(y), ptr(dy), ptr(dx), y.numel(), dtype_code(y.dtype), stream())
    return dx


def dropout(x, p: float, seed: int):
    x = _c(x)
    y = torch.empty_like(x)
    hip.call('vm_dropout', ptr(x), ptr(y), x.numel(), p, seed & 0xFFFFFFFFFFFFFFFF, dtype_code(x.dtype), stream())
    return y


def add(a, b):
    a, b = _c(a), _c(b)
    y = torch.empty_like(a)
    hip.call('vm_add', ptr(a), ptr(b), ptr(y), a.numel(), dtype_code(a.dtype), stream())
    return y


def cast(x, dtype: torch.dtype):
    x = _c(x)
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    hip.call('vm_cast', ptr(x), dtype_code(x.dtype), ptr(y), dtype_code(dtype), x.numel(), stream())
    return y


def gather_rows(src, idx, rows: int | None = None, nrows: torch.Tensor | None = None, out: torch.Tensor | None = None):
    assert src.dim() == 2 and src.stride(1) == 1 and idx.dtype == torch.int32
    rows = idx.numel() if rows is None else rows
    if out is None:
        out = torch.empty(rows, src.shape[1], dtype=src.dtype, device=src.device)
    hip.call('vm_gather_rows', ptr(src), _ld(src), ptr(idx), ptr(out), _ld(out), rows, src.shape[1], dtype_code(src.dtype),
             ptr(nrows), stream())
    return out


def scatter_rows(src, idx, out, rows: int | None = None, nrows: torch.Tensor | None = None):
    assert src.dim() == 2 and src.stride(1) == 1 and idx.dtype == torch.int32
    rows = min(idx.numel(), src.shape[0]) if rows is None else rows
    hip.call('vm_scatter_rows', ptr(src), _ld(src), ptr(idx), ptr(out), _ld(out), rows, src.shape[1], dtype_code(src.dtype),
             ptr(nrows), stream())
    return out


def embedding_bwd(dout, ids: torch.Tensor, rows: torch.Tensor, dweight: torch.Tensor):
    """dweight[id] = sum of dout[row] over entries with that id (ids < 0 ignored). dweight must be pre-zeroed."""
    order = torch.argsort(ids.to(torch.int64), stable=True)
    sorted_ids = ids[order].to(torch.int32).contiguous()
    sorted_rows = rows[order].to(torch.int32).contiguous()
    hip.call('vm_embedding_bwd', ptr(dout), _ld(dout), ptr(sorted_ids), ptr(sorted_rows), sorted_ids.numel(), ptr(dweight),
             _ld(dweight), dout.shape[1], dtype_code(dout.dtype), stream())
    return dweight


def ce_fwd(logits, labels, vocab: int, nrows: torch.Tensor | None = None):
    rows = logits.shape[0]
    row_loss = torch.zeros(rows, dtype=torch.float32, device=logits.device)   # rows >= n_rows must read 0 in the dot product
    lse = torch.empty(rows, dtype=torch.float32, device=logits.device)
    hip.call('vm_ce_fwd', ptr(logits), _ld(logits), ptr(labels), ptr(row_loss), ptr(lse), rows, vocab, dtype_code(logits.dtype),
             ptr(nrows), stream())
    return row_loss, lse


def ce_bwd(logits, labels, lse, row_scale, vocab: int, nrows: torch.Tensor | None = None, out: torch.Tensor | None = None):
    rows = logits.shape[0]
    if out is None:
        out = torch.empty_like(logits)
    hip.call('vm_ce_bwd', ptr(logits), _ld(logits), ptr(labels), ptr(lse), ptr(row_scale), ptr(out), _ld(out), rows, vocab,
             dtype_code(logits.dtype), ptr(nrows), stream())
    return out


def im2col3d(image: torch.Tensor, patch: tuple[int, int, int], pad_k_to: int = 1):
    image = _c(image)
    Cc, D, H, W = image.shape
    pz, py, px = patch
    n = (D // pz) * (H // py) * (W // px)
    K = Cc * pz * py * px
    Kp = (K + pad_k_to - 1) // pad_k_to * pad_k_to
    cols = torch.empty(n, Kp, dtype=image.dtype, device=image.device)
    if Kp > K:
        cols[:, K:].zero_()
    hip.call('vm_im2col3d', ptr(image), Cc, D, H, W, pz, py, px, ptr(cols), Kp, dtype_code(image.dtype), stream())
    return cols


def expert_index_build(token_type_ids: torch.Tensor, attention_mask: torch.Tensor):
    B, L = token_type_ids.shape
    dev = token_type_ids.device
    tt = _c(token_type_ids.to(torch.int64))
    am = _c(attention_mask.to(torch.int64))
    counts = torch.empty(4, dtype=torch.int32, device=dev)
    row_of_tok = torch.empty(B * L, dtype=torch.int32, device=dev)
    tok_of_row = torch.empty(B * L, dtype=torch.int32, device=dev)
    cu = torch.empty(B + 1, dtype=torch.int32, device=dev)
    row_of_pos = torch.empty(B * L, dtype=torch.int32, device=dev)
    mask = torch.empty(B * L, dtype=torch.uint8, device=dev)
    hip.call('vm_expert_index_build', ptr(tt), ptr(am), B, L, ptr(counts), ptr(row_of_tok), ptr(tok_of_row), ptr(cu),
             ptr(row_of_pos), ptr(mask), stream())
    return dict(counts=counts, row_of_tok=row_of_tok, tok_of_row=tok_of_row, cu_seqlens=cu, row_of_pos=row_of_pos,
                expert_mask=mask.view(B, L))


# ------------------------------------------------------------------ attention (bf16, var-len)
# The backward's dS^T scratch (dK / dV leaves dS^T behind, dQ is a plain product over it instead of a second recomputation of S and dP):
# round 4 measured it faster at 785 / 456 tokens, equal at 2049 and slower at 4609 and switched on the sequence length (1536). Re-measured
# inside the step at the end of round 5 (A B A B, profiles/r5_attn_ds_threshold_ab.txt) it wins at every length the workloads have:
# phase-grg-3d (8 x 2049) 660.1 -> 656.2 ms, model-hr-2d (4 x 4097) 594.8 -> 585.9, model-hr-3d (4 x 4609) 827.7 -> 801.6, phase-vlm-mixed
# 465.1 -> 463.4. The sequence limit is now only the kernels' 32-bit offset range; the byte cap bounds the transient allocation (2.8 GB at
# 4 x 4609 x 16 heads; 0: always recompute).
ATTN_DS_MAX_BYTES = 8192 << 20
ATTN_DS_MAX_SEQLEN = 16384
def _attn_args(q, k, v, out, lse, cu_seqlens, max_seqlen, n_heads, head_dim, scale, causal, row_of_pos, total_pos_max):
    a = hip.AttnArgs()
    a.q, a.k, a.v, a.out = ptr(q), ptr(k), ptr(v), ptr(out)
    a.ldq, a.ldk, a.ldv, a.ldo = q.stride(0), k.stride(0), v.stride(0), out.stride(0)
    a.lse = ptr(lse)
    a.cu_seqlens, a.n_seq = ptr(cu_seqlens), cu_seqlens.numel() - 1
    a.row_of_pos = ptr(row_of_pos)
    a.total_pos_max = total_pos_max
    a.max_seqlen = max_seqlen
    a.n_heads, a.head_dim = n_heads, head_dim
    a.scale = scale
    a.causal = int(causal)
    return a


def attn_fwd(q, k, v, cu_seqlens, max_seqlen: int, n_heads: int, head_dim: int, scale: float, causal: bool,
             row_of_pos: torch.Tensor | None = None, total_pos_max: int | None = None):
    """q,k,v: [rows, n_heads*head_dim] views (unit inner stride; may alias one qkv buffer)."""
    assert q.dtype == torch.bfloat16 and q.stride(1) == 1 and k.stride(1) == 1 and v.stride(1) == 1
    rows = q.shape[0]
    total_pos_max = rows if total_pos_max is None else total_pos_max
    # rows that belong to no sequence (packed-layout padding) are never read downstream; lse likewise
    out = torch.empty(rows, n_heads * head_dim, dtype=q.dtype, device=q.device)
    lse = torch.empty(n_heads, total_pos_max, dtype=torch.float32, device=q.device)
    a = _attn_args(q, k, v, out, lse, cu_seqlens, max_seqlen, n_heads, head_dim, scale, causal, row_of_pos, total_pos_max)
    hip.call('vm_attn_fwd_bf16', C.addressof(a), stream())
    return out, lse


def attn_bwd(q, k, v, out, lse, dout, cu_seqlens, max_seqlen, n_heads, head_dim, scale, causal,
             row_of_pos: torch.Tensor | None = None, total_pos_max: int | None = None):
    rows = q.shape[0]
    total_pos_max = rows if total_pos_max is None else total_pos_max
    dout = _c(dout)
    dqkv = torch.empty(rows, 3, n_heads * head_dim, dtype=q.dtype, device=q.device)   # every valid row is written by the dq / dkv kernels
    dq, dk, dv = dqkv[:, 0], dqkv[:, 1], dqkv[:, 2]
    delta = torch.empty(n_heads, total_pos_max, dtype=torch.float32, device=q.device)
    a = _attn_args(q, k, v, out, lse, cu_seqlens, max_seqlen, n_heads, head_dim, scale, causal, row_of_pos, total_pos_max)
    a.dout, a.lddo = ptr(dout), dout.stride(0)
    a.dq, a.dk, a.dv = ptr(dq), ptr(dk), ptr(dv)
    a.lddq, a.lddk, a.lddv = dq.stride(0), dk.stride(0), dv.stride(0)
    a.delta = ptr(delta)
    # dS^T scratch (vm_attn_bwd_workspace_bytes): with it dQ is a product over what dK / dV left behind instead of a second recomputation.
    # Freed on return: the caching allocator hands it out again in stream order.
    need = C.c_int64(0)
    hip.call('vm_attn_bwd_workspace_bytes', C.addressof(a), C.addressof(need))
    ws = None
    if 0 < need.value <= ATTN_DS_MAX_BYTES and max_seqlen <= ATTN_DS_MAX_SEQLEN:
        ws = torch.empty(need.value, dtype=torch.uint8, device=q.device)
        a.workspace, a.workspace_bytes = ptr(ws), need.value
    hip.call('vm_attn_bwd_bf16', C.addressof(a), stream())
    return dqkv


def sumsq_partials(x: torch.Tensor, out: torch.Tensor) -> None:
    """out[b] = sum of squares of workgroup b's share of the flat tensor x (vm_sumsq_partials: fp32, deterministic)"""
    assert x.is_contiguous() and out.dtype == torch.float32 and out.is_contiguous()
    hip.call('vm_sumsq_partials', ptr(x), x.numel(), dtype_code(x.dtype), ptr(out), out.numel(), stream())


def adamw_(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, *, lr: float, betas=(0.9, 0.999), eps: float = 1e-8,
           weight_decay: float = 1e-2, step: int, clip_coef: torch.Tensor | None = None):
    """in-place fused clip + AdamW over flat 1-D buffers of one dtype"""
    assert p.dim() == 1 and p.is_contiguous() and g.shape == p.shape and m.shape == p.shape and v.shape == p.shape
    assert g.dtype == p.dtype and m.dtype == p.dtype and v.dtype == p.dtype
    assert clip_coef is None or (clip_coef.dtype == torch.float32 and clip_coef.numel() == 1)
    hip.call('vm_adamw', ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, betas[0], betas[1], eps, weight_decay, step, ptr(clip_coef),
             dtype_code(p.dtype), stream())


# ------------------------------------------------------------------ trilinear up-sampling
def upsample_trilinear3d(x: torch.Tensor, size) -> torch.Tensor:
    """x fp32 [n, d, h, w] -> [n, *size]: F.interpolate(mode='trilinear', align_corners=False)"""
    assert x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
    n, d, h, w = x.shape
    D, H, W = (int(v) for v in size)
    y = torch.empty(n, D, H, W, dtype=x.dtype, device=x.device)
    hip.call('vm_upsample_trilinear3d_fwd', ptr(x), ptr(y), n, d, h, w, D, H, W, stream())
    return y


def upsample_trilinear3d_bwd(gy: torch.Tensor, in_shape) -> torch.Tensor:
    assert gy.dtype == torch.float32 and gy.dim() == 4 and gy.is_contiguous()
    n, D, H, W = gy.shape
    d, h, w = (int(v) for v in in_shape)
    gx = torch.empty(n, d, h, w, dtype=gy.dtype, device=gy.device)
    hip.call('vm_upsample_trilinear3d_bwd', ptr(gy), ptr(gx), n, d, h, w, D, H, W, stream())
    return gx


# ------------------------------------------------------------------ Dice + focal loss
def dice_focal_fwd(x: torch.Tensor, target: torch.Tensor | None, gamma: float, alpha: float | None):
    """x fp32 [R, n], target uint8 [R, n] | None -> (sums [R, 4], out [R, 2] = (dice, focal sum)); mmmm/models/loss.py:32-56"""
    assert x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous()
    assert target is None or (target.dtype == torch.uint8 and target.shape == x.shape and target.is_contiguous())
    R, n = x.shape
    sums = torch.empty(R, 4, dtype=torch.float32, device=x.device)
    out = torch.empty(R, 2, dtype=torch.float32, device=x.device)
    nb = C.c_int64(0)
    hip.call('vm_dice_focal_workspace', R, n, C.addressof(nb))
    ws = torch.empty(max(nb.value // 4, 1), dtype=torch.float32, device=x.device)
    hip.call('vm_dice_focal_fwd', ptr(x), ptr(target), R, n, float(gamma), -1.0 if alpha is None else float(alpha), ptr(sums), ptr(out),
             ptr(ws), nb.value, stream())
    return sums, out


def dice_focal_bwd(x, target, gamma: float, alpha: float | None, sums, g_dice, g_focal):
    dx = torch.empty_like(x)
    hip.call('vm_dice_focal_bwd', ptr(x), ptr(target), x.shape[0], x.shape[1], float(gamma), -1.0 if alpha is None else float(alpha),
             ptr(sums), ptr(g_dice), ptr(g_focal), ptr(dx), stream())
    return dx


# ------------------------------------------------------------------ Hungarian matching
LSAP_MAX_COLS = 64


def lsap(cost: torch.Tensor, dims: torch.Tensor, max_cols: int) -> torch.Tensor:
    """cost fp32 [P, R, C] (padded), dims int32 [P, 2] = (rows, cols) of each problem (rows <= cols <= 64; rows 0 = padding
    entry) -> int32 [P, R]: column assigned to each row, identical to scipy.optimize.linear_sum_assignment"""
    assert cost.dtype == torch.float32 and cost.dim() == 3 and cost.stride(2) == 1 and dims.dtype == torch.int32
    P, R, _ = cost.shape
    out = torch.full((P, R), -1, dtype=torch.int32, device=cost.device)
    hip.call('vm_lsap_f32', ptr(cost), cost.stride(0), cost.stride(1), ptr(dims), ptr(out), out.stride(0), P, int(max_cols), stream())
    return out


def box_match_cost(desc: torch.Tensor, n_problems: int, rows: int, width: int, l1_weight: float, giou_weight: float, disc_weight: float,
                   match_ce: bool, gamma: float, alpha: float | None) -> torch.Tensor:
    """desc int64 [P, 6] on the device = {reg ptr, logit ptr, label ptr, n_pos, n_col, nq} per target -> cost fp32 [P, rows, width]"""
    assert desc.dtype == torch.int64 and desc.is_contiguous() and desc.numel() >= n_problems * 6
    cost = torch.empty(n_problems, rows, width, dtype=torch.float32, device=desc.device)
    hip.call('vm_box_match_cost', ptr(desc), n_problems, ptr(cost), rows, width, l1_weight, giou_weight, disc_weight, int(match_ce),
             gamma, -1.0 if alpha is None else alpha, stream())
    return cost


def _instance_loss_args(logit, reg, label, match):
    nt, nq = logit.shape
    assert logit.dtype == torch.float32 and reg.dtype == torch.float32 and label.dtype == torch.float32 and match.dtype == torch.int64
    assert reg.shape == (nt, nq + 1, 6) and label.dim() == 2 and label.shape[1] == 6 and match.shape == (nt, nq)
    assert logit.is_contiguous() and reg.is_contiguous() and label.is_contiguous() and match.is_contiguous()
    return nt, nq


def instance_loss_fwd(logit: torch.Tensor, reg: torch.Tensor, label: torch.Tensor, match: torch.Tensor, gamma: float,
                      alpha: float | None) -> torch.Tensor:
    """-> fp32 [6]: focal mean (all), focal mean of matched vs 1, of unmatched vs 0, l1 mean, 1 - mean giou, matched count"""
    nt, nq = _instance_loss_args(logit, reg, label, match)
    out = torch.empty(6, dtype=torch.float32, device=logit.device)
    hip.call('vm_instance_loss_fwd', ptr(logit), ptr(reg), ptr(label), ptr(match), nt, nq, gamma, -1.0 if alpha is None else alpha,
             ptr(out), stream())
    return out


def instance_loss_bwd(logit, reg, label, match, gamma: float, alpha: float | None, out6: torch.Tensor, grad_out: torch.Tensor):
    """-> (d_logit like logit, d_reg like reg)"""
    nt, nq = _instance_loss_args(logit, reg, label, match)
    assert grad_out.dtype == torch.float32 and grad_out.is_contiguous() and grad_out.numel() == 6
    d_logit, d_reg = torch.empty_like(logit), torch.empty_like(reg)
    hip.call('vm_instance_loss_bwd', ptr(logit), ptr(reg), ptr(label), ptr(match), nt, nq, gamma, -1.0 if alpha is None else alpha,
             ptr(out6), ptr(grad_out), ptr(d_logit), ptr(d_reg), stream())
    return d_logit, d_reg


# ------------------------------------------------------------------ generation path
@functools.lru_cache(maxsize=64)
def _attn_decode_ws_bytes(batch: int, n_heads: int, head_dim: int, max_len: int) -> int:
    n = C.c_int64(0)
    hip.call('vm_attn_decode_workspace', batch, n_heads, head_dim, max_len, C.addressof(n))
    return n.value


def attn_decode(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor, kv_lens: torch.Tensor, n_heads: int, head_dim: int,
                scale: float, max_len: int, out: torch.Tensor | None = None) -> torch.Tensor:
    """single-query attention of one new token per sample against a KV cache (modeling_cogvlm.py:129-141).
    q [B, H*hd] (row stride free), k_cache / v_cache [B, Lmax, H*hd] bf16, kv_lens int32[B] (new token included),
    max_len: host upper bound of kv_lens -> [B, H*hd]"""
    B = q.shape[0]
    assert q.dtype == torch.bfloat16 and k_cache.dtype == torch.bfloat16 and v_cache.dtype == torch.bfloat16
    assert q.stride(1) == 1 and k_cache.dim() == 3 and k_cache.stride(2) == 1 and k_cache.stride() == v_cache.stride()
    assert kv_lens.dtype == torch.int32 and kv_lens.numel() == B and 0 < max_len <= k_cache.shape[1]
    if out is None:
        out = torch.empty(B, n_heads * head_dim, dtype=q.dtype, device=q.device)
    nbytes = _attn_decode_ws_bytes(B, n_heads, head_dim, max_len)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=q.device)
    hip.call('vm_attn_decode_bf16', ptr(q), _ld(q), ptr(k_cache), ptr(v_cache), k_cache.stride(1), k_cache.stride(0), ptr(kv_lens),
             ptr(out), _ld(out), B, n_heads, head_dim, max_len, scale, ptr(ws), nbytes, stream())
    return out


def gemv_supported(x: torch.Tensor, w: torch.Tensor, a2: torch.Tensor | None = None) -> bool:
    return (x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.dim() == 2 and x.shape[0] <= 16 and x.shape[1] % 64 == 0
            and (a2 is None or a2.shape[1] % 64 == 0))


def gemv(x: torch.Tensor, w: torch.Tensor, *, a2: torch.Tensor | None = None, b2: torch.Tensor | None = None, alpha2: float = 1.0,
         bias: torch.Tensor | None = None, residual: torch.Tensor | None = None, out: torch.Tensor | None = None) -> torch.Tensor:
    """out[M<=16, N] = x w^T + alpha2 * a2 b2^T + bias (+ residual after rounding), bf16, W streamed once (decode step)"""
    M, Kd = x.shape
    N = w.shape[0]
    assert gemv_supported(x, w, a2) and w.shape[1] == Kd and (a2 is None) == (b2 is None)
    if out is None:
        out = torch.empty(M, N, dtype=x.dtype, device=x.device)
    K2 = 0 if a2 is None else a2.shape[1]
    hip.call('vm_gemv_bf16', ptr(x), _ld(x), ptr(w), _ld(w), ptr(a2), _ld(a2) if a2 is not None else 0, ptr(b2),
             _ld(b2) if b2 is not None else 0, alpha2, ptr(bias), ptr(residual), _ld(residual) if residual is not None else 0,
             ptr(out), _ld(out), M, N, Kd, K2, stream())
    return out


# ------------------------------------------------------------------ profiling helpers
def prof_enable(kinds=True):
    """kinds: True (every kind), False / () (off) or an iterable of hip.PROF_* kinds"""
    mask = 0xF if kinds is True else 0 if not kinds else sum(1 << int(k) for k in kinds)
    hip.call('vm_prof_enable', mask)


def prof_stride(every: int):
    hip.call('vm_prof_stride', int(every))


def prof_reset():
    hip.call('vm_prof_reset')


def prof_last_bytes() -> float:
    b = C.c_double()
    hip.call('vm_prof_last_bytes', C.addressof(b))
    return b.value


def prof_collect(kind: int):
    ms, fl, n = C.c_double(), C.c_double(), C.c_int64()
    hip.call('vm_prof_collect', kind, C.addressof(ms), C.addressof(fl), C.addressof(n))
    return ms.value, fl.value, n.value


# ------------------------------------------------------------------ attention (fp32 islands)
def _attn_f32_args(q, k, v, out, lse, n_heads, head_dim, scale, cu_seqlens):
    """q [Bn, Lq, H*hd] / k,v [Bn, Lk, H*hd] (strided views allowed, inner dim contiguous); or packed
    [T, H*hd] with cu_seqlens (self-attention)."""
    a = hip.AttnF32Args()
    a.q, a.k, a.v, a.out = ptr(q), ptr(k), ptr(v), ptr(out)
    if cu_seqlens is not None:
        a.q_ls, a.k_ls, a.v_ls, a.o_ls = q.stride(0), k.stride(0), v.stride(0), out.stride(0)
        a.Bn, a.Lq, a.Lk = 1, q.shape[0], k.shape[0]
        a.cu_seqlens, a.n_seq = ptr(cu_seqlens), cu_seqlens.numel() - 1
    else:
        a.q_bs, a.q_ls, a.k_bs, a.k_ls = q.stride(0), q.stride(1), k.stride(0), k.stride(1)
        a.v_bs, a.v_ls, a.o_bs, a.o_ls = v.stride(0), v.stride(1), out.stride(0), out.stride(1)
        a.Bn, a.Lq, a.Lk = q.shape[0], q.shape[1], k.shape[1]
    a.lse = ptr(lse)
    a.n_heads, a.head_dim, a.scale = n_heads, head_dim, scale
    return a


def attn_f32_fwd(q, k, v, n_heads: int, head_dim: int, scale: float, cu_seqlens=None, max_seqlen: int | None = None, f32_split: int = 0,
                 causal: bool = False, row_of_pos: torch.Tensor | None = None):
    """`causal` / `row_of_pos` (packed self-attention, exact arithmetic): the towers' fp32 mode — see vm_attn_f32_args"""
    assert q.dtype == torch.float32 and q.stride(-1) == 1 and k.stride(-1) == 1 and v.stride(-1) == 1
    # (indirect layout: rows that belong to no sequence are never written — zero them, the caller's row-wise kernels may still pass over them)
    out = (torch.zeros if row_of_pos is not None else torch.empty)(*q.shape[:-1], n_heads * head_dim, dtype=q.dtype, device=q.device)
    n_q = q.shape[0] if cu_seqlens is not None else q.shape[0] * q.shape[1]
    lse = torch.empty(n_heads, n_q, dtype=torch.float32, device=q.device)
    a = _attn_f32_args(q, k, v, out, lse, n_heads, head_dim, scale, cu_seqlens)
    if cu_seqlens is not None and max_seqlen is not None:
        a.Lq = a.Lk = max_seqlen
    a.f32_split = f32_split          # head_dim 64 only: 2 / 3 = split-bf16 products (3 / 6 MFMAs), else the exact f32 MFMA chain
    a.causal, a.row_of_pos = int(causal), ptr(row_of_pos)
    hip.call('vm_attn_fwd_f32', C.addressof(a), stream())
    return out, lse


def attn_f32_bwd(q, k, v, out, lse, dout, n_heads: int, head_dim: int, scale: float, cu_seqlens=None, max_seqlen: int | None = None,
                 grads=None, f32_split: int = 0, causal: bool = False, row_of_pos: torch.Tensor | None = None):
    """`grads`: optional (dq, dk, dv) destinations with the SAME strides as q / k / v (the kernel addresses them with the
    operands' strides), e.g. the thirds of one packed dqkv"""
    dout = _c(dout)
    if grads is not None:
        dq, dk, dv = grads
        assert all(g.stride() == t.stride() for g, t in zip(grads, (q, k, v)))
    else:
        # (every row of a dense [Bn, L, C] operand is written by the dQ / dKV kernels: no zero fill; packed rows past cu[-1] would not be)
        mk = torch.empty_like if cu_seqlens is None else torch.zeros_like
        dq, dk, dv = mk(q, memory_format=torch.contiguous_format), mk(k, memory_format=torch.contiguous_format), \
            mk(v, memory_format=torch.contiguous_format)
        q, k, v = _c(q), _c(k), _c(v)
    a = _attn_f32_args(q, k, v, out, lse, n_heads, head_dim, scale, cu_seqlens)
    if cu_seqlens is not None and max_seqlen is not None:
        a.Lq = a.Lk = max_seqlen
    if cu_seqlens is not None:
        a.do_ls = dout.stride(0)
    else:
        a.do_bs, a.do_ls = dout.stride(0), dout.stride(1)
    delta = torch.empty_like(lse)
    a.dout, a.dq, a.dk, a.dv, a.delta = ptr(dout), ptr(dq), ptr(dk), ptr(dv), ptr(delta)
    a.f32_split = f32_split
    a.causal, a.row_of_pos = int(causal), ptr(row_of_pos)
    hip.call('vm_attn_bwd_f32', C.addressof(a), stream())
    return dq, dk, dv
