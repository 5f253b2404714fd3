"""Autograd operators of the VividMed step, each a torch.autograd.Function over the C-ABI kernels.

torch is used here for what the task calls plumbing: device memory, streams, and the autograd tape that
strings the hand-written forward/backward kernels together. Every FLOP of the listed ops runs in
libvividmed_hip.so; there is no eager/PyTorch fallback (a CPU tensor raises in mmmm_amd.hip.ptr).
"""
from __future__ import annotations

import weakref
from dataclasses import dataclass

import torch
from torch.autograd import Function

from . import hip
from . import kernels as K


def once_differentiable(fn):
    """The kernels' backward passes are not differentiable again. torch's decorator of this name wraps EVERY backward call in a no_grad
    context and an output check (~3 us x ~1 200 nodes per step on autograd's thread); the engine already runs backward with grad mode
    off unless create_graph=True — so the only case to catch is that one, and it is an error here."""
    def backward(ctx, *grads):
        if torch.is_grad_enabled():
            raise RuntimeError('the VividMed HIP operators are once differentiable: backward under create_graph=True is not supported')
        return fn(ctx, *grads)
    backward.__name__ = getattr(fn, '__name__', 'backward')
    backward.__doc__ = fn.__doc__
    return backward


def _direct(cls):
    """`cls.call` = the C++ `apply` of the autograd Function without `Function.apply`'s per-call functorch pass over the arguments
    (`unwrap_dead_wrappers` + the setup_context probe: 8.2 -> 4.5 us per node, ~1 200 nodes per forward). No functorch transform is
    ever active around these operators (they would have to trace raw device pointers)."""
    cls.call = staticmethod(super(Function, cls).apply)
    return cls


_CU_CACHE: dict = {}


_UPLOAD_STREAMS: dict = {}


def cu_seqlens_tensor(lens, device) -> torch.Tensor:
    """int32 prefix sums of `lens` on `device`, cached per (lens, device). A miss uploads through pinned memory on a dedicated
    copy stream and waits for THAT copy only: a fresh torch.tensor(..., device=...) is a blocking pageable host->device copy queued
    behind everything on the current stream, i.e. a full host/GPU synchronisation in the middle of the forward pass; a non-blocking
    copy on the current stream would be invisible to the OTHER streams that read the cached table (the two grounding heads run on two
    streams: the first step after a miss read a half-written table, found by the full-size replay tests)."""
    key = (tuple(int(n) for n in lens), str(device))
    t = _CU_CACHE.get(key)
    if t is None:
        cu = [0]
        for n in key[0]:
            cu.append(cu[-1] + n)
        if len(_CU_CACHE) > 256:
            if torch.device(device).type == 'cuda':
                torch.cuda.synchronize(device)          # kernels in flight may still read the tables about to be freed
            _CU_CACHE.clear()
        host = torch.tensor(cu, dtype=torch.int32)
        dev = torch.device(device)
        if dev.type == 'cuda':
            st = _UPLOAD_STREAMS.get(dev)
            if st is None:
                st = _UPLOAD_STREAMS[dev] = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                t = host.pin_memory().to(dev, non_blocking=True)
            st.synchronize()                            # the table is resident before any stream can be handed it
        else:
            t = host.to(dev)
        _CU_CACHE[key] = t
    return t


# ----------------------------------------------------------------------------- linear (plain / gated / LoRA)
@dataclass
class LinearMeta:
    """non-tensor arguments of `linear`"""
    gated: bool = False          # two row segments (vision expert | language expert), boundary from `counts`
    lora_scale: float = 0.0      # alpha/sqrt(r) (rsLoRA) — 0 disables the LoRA path
    drop_p: float = 0.0          # lora_dropout (training only)
    drop_seed: int = 0
    act: int = hip.ACT_NONE      # fused epilogue activation (only when the input needs no pre-activation later)
    out_dtype: torch.dtype | None = None
    # resident K-contiguous copies of the LoRA factors (models/lora.py LoraTransposes); None -> transposed per use
    At0: torch.Tensor | None = None
    Bt0: torch.Tensor | None = None
    At1: torch.Tensor | None = None
    Bt1: torch.Tensor | None = None
    # fp8 mode of the frozen base weight (models/lora.py Fp8Weights per expert): e4m3 copies of W and W^T with their scales
    f8_0: object = None
    f8_1: object = None
    # fp32 operands: arithmetic of this layer's three GEMMs (kernels.gemm `f32_split`; 0 = the process default)
    f32_split: int = 0
    # also return x itself (a view): a post-norm block's residual takes THAT output, so the residual's gradient arrives in this
    # node's backward and rides in the dgrad GEMM's epilogue instead of a separate element-wise add over the activations
    fork: bool = False


def _t(w: torch.Tensor) -> torch.Tensor:
    """[N,K] -> contiguous [K,N] through the transpose kernel (small tensors: LoRA factors)"""
    return K.transpose(w.detach())


def _ktile(dtype: torch.dtype) -> int:
    return 64 if dtype == torch.bfloat16 else 32


def _padk(*ts):
    """zero-pad the contraction (last) dimension up to the GEMM K-tile. Only small odd-shaped layers of the fp32
    grounding heads (box / disc heads, hyper-network outputs) and tiny test models ever take this path."""
    k = ts[0].shape[-1]
    kt = _ktile(ts[0].dtype)
    if k % kt == 0:
        return ts if len(ts) > 1 else ts[0]
    pad = kt - k % kt
    out = tuple(None if t is None else torch.nn.functional.pad(t, (0, pad)) for t in ts)
    return out if len(out) > 1 else out[0]


def _lora_project(x, A0, A1, gated, counts, drop_p=0.0, seed=0):
    """t = drop(x) @ A^T : fused skinny kernel when the shape allows, else dropout + padded NT GEMM"""
    if K.lora_down_supported(x, A0):
        return K.lora_down(x, A0, A1 if gated else None, counts=counts if gated else None, drop_p=drop_p, drop_seed=seed)
    xd = K.dropout(x, drop_p, seed) if drop_p > 0 else x
    xdp, A0p, A1p = _padk(xd, A0, A1)
    return K.gemm(xdp, A0p, w1=A1p if gated else None, counts=counts if gated else None)


_WGRAD_STREAMS: dict = {}
_WGRAD_EVENTS: dict = {}
SIDE_LAG = 24          # forked calls the side stream may trail the main stream by (~2 transformer layers)


_HELD_BY_TASK: dict = {}     # graph-task id -> tensors held until that backward pass ends


def _hold_until_backward_ends(t: torch.Tensor):
    """Keep a second reference to `t` until autograd finishes the current backward pass.
    The output gradient `dy` of a linear with a fused residual is handed back to autograd as the residual's gradient, and the
    engine accumulates further gradients into such a buffer IN PLACE when it holds the last reference (input_buffer.cpp:
    can_accumulate_inplace) — on the main stream, while the factor-gradient kernels forked to the side stream may still be
    reading it. `record_stream` does not help (it only stops the allocator from recycling FREED memory). A second reference
    makes the engine add out of place; once the backward pass is over nothing accumulates any more and the reference goes.
    (Found by replaying a full-size step, tests/test_fullsize_gpu.py: the LoRA-B gradients of the decoder's dense / down_proj
    linears came out different on every run.)
    The release callback is registered once per BACKWARD PASS, keyed on the engine's graph-task id: the engine drops its final
    callbacks when a backward pass raises (an out-of-memory error in the calibration steps of bench.py), and a registration keyed on
    "the list is empty" would then never happen again — every later step would append its dy tensors and free none."""
    task = torch._C._current_graph_task_id()
    lst = _HELD_BY_TASK.get(task)
    if lst is None:
        # one list per backward pass: a nested pass (reentrant checkpoint, double backward) must not drop the outer pass's tensors.
        # The list of a pass that RAISED is never released by its callback (the engine drops it): `_drop_stale_held` (the next
        # forward pass) or ddp.abort_step removes it.
        lst = _HELD_BY_TASK[task] = []
        torch.autograd.Variable._execution_engine.queue_callback(lambda task=task: _HELD_BY_TASK.pop(task, None))
    lst.append(t)


def _release_held():
    _HELD_BY_TASK.clear()


def _drop_stale_held():
    """called from forward ops: outside any backward pass (graph-task id < 0 — a checkpoint recompute runs INSIDE one) nothing can be
    legitimately held, so whatever is left belongs to passes that raised"""
    if _HELD_BY_TASK and torch._C._current_graph_task_id() < 0:
        _HELD_BY_TASK.clear()


def abort_backward_state():
    """forget what an aborted forward / backward pass left behind in this module (ddp.BucketedGradAllReduce.abort_step)"""
    _release_held()
    for ring in _WGRAD_EVENTS.values():
        ring.clear()
    _WGRAD_QUEUE.clear()
    _WGRAD_QUEUE_IDS.clear()
    _WGRAD_QUEUE_WGS[0] = 0
    _WGRAD_QUEUE_STATE[0] = _WGRAD_QUEUE_STATE[1] = None
    if '_U_STASH' in globals():
        _U_STASH.clear()
        _U_STASH_TASK[0] = None


# [r3] OFF by default: with the chip saturated by the main stream the side stream no longer hides anything (round 2 already measured
# 334.3 vs 334.9 ms) and, now that the LoRA factor gradients go out as grouped launches on the main stream, it costs time:
# 329.9 (on) vs 325.8 / 325.3 ms (off), A B A in one call. It also withholds memory from the caching allocator.
WGRAD_SIDE_STREAM = False      # (module constants, not environment switches: tests and `bench.py --set functional.NAME=value` flip them)
# 1: dgrad reads the weight as stored (`b_nn` form of the 256-column GEMM) and no transposed copies of the frozen weights are kept
# (-35 GB). Bit-identical to the NT form but currently 0.6-0.85x its rate (192-row tiles only, 64-byte DMA segments, twice the LDS
# read instructions): the step loses more than the freed memory buys (model-hr-3d: every layer kept, yet 3.47 vs 3.59 images/s;
# phase-vg-448 445 vs 359 ms), so the default stays the NT dgrad on resident transposes.
NN_DGRAD = False
# fp32 weight gradients of the unfrozen heads through the TN kernel (vm_gemm_tn_f32: no transposed copies of dy and x); 0: round 2's
# two transposes + NT GEMM (A/B measurements)
F32_TN_WGRAD = True
# 0: ops asked to `fork` (hand their input back for the block's residual) return the input itself, i.e. autograd sums the two gradients of
# the input with its own element-wise add (A/B measurements)
FORK = True
# the LINEAR form of the fork (post-norm ViT-E blocks: the residual's gradient rides in the dgrad GEMM's epilogue): removes 126 element-wise
# adds per step and makes 126 dgrad GEMMs read one more [tokens, hidden] operand. A tie in rounds 2-4 (342.4 vs 342.8 ms) and at the start of
# round 5 (299.8 vs 299.5); with the rest of the step tightened it is a small, repeatable gain — 299.15 -> 298.53 ms over three A B pairs in
# one call (profiles/r5_fork_linear_ab.txt), the residual costs the epilogue 0.6-3 us where the add kernel takes 10.5 — so it is ON.
FORK_LINEAR = True


def _off_critical_path(fn, device, keep_alive):
    """Run `fn` (LoRA factor-gradient kernels that accumulate straight into the gradient buckets: nothing downstream in
    backward reads their result) on a side stream. They are HBM-bound and their workgroups need little of a CU, so they fill
    the tails of the MFMA-bound dgrad GEMMs on the main stream instead of serialising with them. The gradient bucket joins
    the side stream before it is reduced / clipped (ddp._launch); `keep_alive` tensors are recorded on the side stream so
    the caching allocator cannot hand their memory out while the kernels still read it."""
    if not (WGRAD_SIDE_STREAM and device.type == 'cuda'):
        fn()
        return
    side = _WGRAD_STREAMS.get(device)
    if side is None:
        side = _WGRAD_STREAMS[device] = torch.cuda.Stream(device=device)
    # (set_stream directly: the `torch.cuda.stream` context manager costs ~20 us of host time per use and this runs ~1.4k
    # times per step on the autograd thread; measured neutral for the step time — the step is GPU-bound — but it keeps the
    # host's launch lead comfortable)
    cur = torch.cuda.current_stream(device)
    side.wait_stream(cur)
    torch.cuda.set_stream(side)
    try:
        fn()
    finally:
        torch.cuda.set_stream(cur)
    for t in keep_alive:
        if t is not None:
            t.record_stream(side)
    # Bound how far the side stream may fall behind. Every tensor recorded on it (whole activation / gradient matrices) is withheld
    # from the caching allocator until the side stream has passed this point, and a side stream that runs "in the shadows" of the
    # dgrad GEMMs can lag by dozens of layers: the allocator then grows its pool instead of recycling — 170 GB allocated but
    # 200-260 GB reserved on the high-resolution workloads, different from run to run. The main stream therefore waits for the work
    # forked SIDE_LAG calls ago (normally long finished: no stall), which caps the withheld memory at a few layers' worth.
    ring = _WGRAD_EVENTS.setdefault(device, [])
    ev = torch.cuda.Event()
    ev.record(side)
    ring.append(ev)
    if len(ring) > SIDE_LAG:
        cur.wait_event(ring.pop(0))


# LoRA factor gradients are not launched one by one: they are queued and go out as ONE grouped launch per ~layer
# (kernels.tn_skinny_group / vm_tn_skinny_group_bf16): a transformer layer's backward produces 8 (ViT-E) or 20 (decoder, two experts)
# of them, each too small for the chip alone. 0: one launch (+ its reduce launch) per factor as in round 2 (A/B measurements).
WGRAD_GROUP = True
_WGRAD_QUEUE: list = []           # [(item for kernels.tn_skinny_group, param, ready callback)]
_WGRAD_QUEUE_IDS: set = set()     # id() of the queued parameters (the queued entries keep them alive)
_WGRAD_QUEUE_WGS = [0]            # workgroups the queued items will launch
WGRAD_GROUP_SLOTS = 768           # 3 workgroups of tn_group_k per CU x 256 CUs
_WGRAD_QUEUE_STATE = [None, None]     # graph-task id of the backward pass the queue belongs to, stream its operands were produced on


def _queue_wgrad(param, ready, W, S, transpose_out, counts, seg, alpha, drop_p, seed):
    task = torch._C._current_graph_task_id()
    st = hip.current_stream_obj(W.device)            # (one cached object per stream: `is` instead of Stream.__eq__)
    if task != _WGRAD_QUEUE_STATE[0] or st is not _WGRAD_QUEUE_STATE[1]:
        flush_wgrad_queue()
        if task != _WGRAD_QUEUE_STATE[0] and task >= 0:
            # whatever is still queued when this backward pass ends goes out then (callers may read .grad right after backward())
            torch.autograd.Variable._execution_engine.queue_callback(flush_wgrad_queue)
        _WGRAD_QUEUE_STATE[0], _WGRAD_QUEUE_STATE[1] = task, st
    if id(param) in _WGRAD_QUEUE_IDS:
        # the same slot twice in one grouped launch (a LoRA linear applied twice within 24 queued factors: shared modules, depth-1 models):
        # two workgroups would read-modify-write one tile unsynchronised — the earlier items go out first
        flush_wgrad_queue()
    # A grouped launch is ONE round of workgroups (256 columns of one factor each, every one walking all rows at its own pace; three fit a
    # CU): it should fill the 3 x CUs slots and never exceed them. ViT-E: 176 workgroups per layer -> four layers per launch (704 of 768;
    # three layers = 528 ran at 4.0 TB/s where the path gives 5.3-5.5); decoder: 546 per layer -> a layer and a third.
    wgs = (W.shape[1] + 255) // 256
    if _WGRAD_QUEUE and _WGRAD_QUEUE_WGS[0] + wgs > WGRAD_GROUP_SLOTS:
        flush_wgrad_queue()
    _WGRAD_QUEUE_WGS[0] += wgs
    _WGRAD_QUEUE_IDS.add(id(param))
    _WGRAD_QUEUE.append(((W, S, param.grad, transpose_out, counts, seg, alpha, drop_p, seed), param, ready))
    if len(_WGRAD_QUEUE) >= hip.TN_GROUP_MAX:
        flush_wgrad_queue()


def flush_wgrad_queue():
    """launch the queued factor gradients (on the stream their operands were produced on) and tell the gradient buckets. Called
    when the queue is full, before a gradient bucket is reduced (ddp._launch / finish) and at the end of every backward pass."""
    if not _WGRAD_QUEUE:
        return
    items = _WGRAD_QUEUE[:]
    _WGRAD_QUEUE.clear()
    _WGRAD_QUEUE_IDS.clear()
    _WGRAD_QUEUE_WGS[0] = 0
    st = _WGRAD_QUEUE_STATE[1]
    cur = hip.current_stream_obj(st.device)
    if cur is not st:
        torch.cuda.set_stream(st)
    try:
        K.tn_skinny_group([it for it, _, _ in items])
        for _, p, ready in items:
            ready(p)
    finally:
        if cur is not st:
            torch.cuda.set_stream(cur)


# Parameters whose gradients live in flat reduction buckets, by storage address (filled by ddp.BucketedGradAllReduce). Under activation
# checkpointing (torch.utils.checkpoint, non-reentrant) a backward node gets its saved tensors back as DETACHED aliases of what the
# recomputed forward saved: a saved parameter then is a plain tensor without `.grad` and without the bucket's tags, and the kernels
# would fall back to returning gradient tensors for AccumulateGrad — every layer of the reference's own checkpoint-everything mode
# took that path (found when the grouped launch made the two paths differ in the last bit: tools/debug_group_inputs.py).
# Weak references: a model that goes away takes its entries (and through `.grad` its flat bucket buffers) with it; a parameter whose
# storage has moved since registration (`.to()`, a load replacing `.data`) is no longer found under its old address.
PARAM_BY_PTR = weakref.WeakValueDictionary()


def _real_param(p):
    """the registered Parameter behind a (possibly detached) alias of it, else `p` itself"""
    if p is None or getattr(p, '_vm_grad_ready', None) is not None:
        return p
    q = PARAM_BY_PTR.get(p.data_ptr())
    return q if (q is not None and q.data_ptr() == p.data_ptr() and q.shape == p.shape and q.dtype == p.dtype) else p


def _direct_slot(param):
    """the parameter's gradient lives in a flat reduction bucket the kernels may accumulate into (ddp.BucketedGradAllReduce)"""
    return (getattr(param, '_vm_grad_ready', None) is not None and param.grad is not None
            and param.grad.dtype in (torch.bfloat16, torch.float32) and param.grad.stride(-1) == 1)


def _lora_wgrad(param, W, S, transpose_out, counts, seg, scale, drop_p, seed):
    """gradient of one LoRA factor through the skinny row-contraction kernel. When the parameter's gradient lives in a
    flat reduction bucket (ddp.BucketedGradAllReduce tags it with `_vm_grad_ready`) the kernel accumulates straight into
    that view and autograd gets None: no temporary, no AccumulateGrad add, and the bucket is told the slot is ready."""
    if not K.tn_skinny_supported(W, S):
        if transpose_out:
            return K.gemm_tn(S, W, counts=counts, segment=seg, alpha=scale, drop_p=drop_p, drop_seed=seed)
        return K.gemm_tn(W, S, counts=counts, segment=seg, alpha=scale)
    param = _real_param(param)
    if _direct_slot(param):
        ready = param._vm_grad_ready
        if WGRAD_GROUP:
            _queue_wgrad(param, ready, W, S, transpose_out, counts, seg, scale, drop_p, seed)
            return None

        def run():
            K.tn_skinny(W, S, transpose_out=transpose_out, out=param.grad, accumulate=True, counts=counts, segment=seg, alpha=scale,
                        drop_p=drop_p, drop_seed=seed)
            ready(param)
        _off_critical_path(run, W.device, (W, S, counts))
        return None
    return K.tn_skinny(W, S, transpose_out=transpose_out, counts=counts, segment=seg, alpha=scale, drop_p=drop_p, drop_seed=seed,
                       out_dtype=param.dtype)


def _lora_wgrad_pair(params, W, S, transpose_out, counts, scale, drop_p, seed):
    """both experts of a gated linear in ONE launch (segment 0 -> params[0], segment 1 -> params[1]); returns the pair of
    gradients for autograd (None where the kernel accumulated straight into the bucket view)."""
    p0, p1 = (_real_param(q) for q in params)
    direct = _direct_slot(p0) and _direct_slot(p1) and p0.grad.dtype == p1.grad.dtype and p0.grad.stride() == p1.grad.stride()
    if direct and WGRAD_GROUP:
        _queue_wgrad(p0, p0._vm_grad_ready, W, S, transpose_out, counts, 0, scale, drop_p, seed)
        _queue_wgrad(p1, p1._vm_grad_ready, W, S, transpose_out, counts, 1, scale, drop_p, seed)
        return None, None
    if direct:
        r0, r1 = p0._vm_grad_ready, p1._vm_grad_ready

        def run():
            K.tn_skinny(W, S, transpose_out=transpose_out, out=(p0.grad, p1.grad), accumulate=True, counts=counts, segment=2,
                        alpha=scale, drop_p=drop_p, drop_seed=seed)
            r0(p0)
            r1(p1)
        _off_critical_path(run, W.device, (W, S, counts))
        return None, None
    return K.tn_skinny(W, S, transpose_out=transpose_out, counts=counts, segment=2, alpha=scale, drop_p=drop_p, drop_seed=seed,
                       out_dtype=p0.dtype)


@_direct
class _Linear(Function):
    """y = act(x W^T + s·(drop(x) A^T) B^T + b) + residual, optionally per row segment.

    Tensor arguments: x, residual, counts, then per expert e in (0, 1): W_e, Wt_e (frozen transposed copy or
    None), b_e, A_e, B_e. Gradients: x, residual, and every tensor that requires grad."""

    @staticmethod
    def forward(ctx, meta: LinearMeta, x, residual, counts, W0, Wt0, b0, A0, B0, W1, Wt1, b1, A1, B1):
        x_in = x
        x = x if x.is_contiguous() else x.contiguous()
        lora = meta.lora_scale != 0.0 and A0 is not None
        t = None
        if lora:
            t = _lora_project(x, A0, A1, meta.gated, counts, meta.drop_p, meta.drop_seed)
        if meta.f8_0 is not None and meta.act == hip.ACT_NONE:
            # frozen base weight in e4m3 (BASELINE configs[4]): quantise the activation rows, main product on the fp8 MFMA, the LoRA
            # extension in bf16 with its operands pre-divided by the scales the epilogue multiplies back
            f0, f1 = meta.f8_0, meta.f8_1
            x8, sx, inv_x = K.quant_rows_fp8(x, counts[1:2] if meta.gated else None)
            y = K.gemm_fp8(
                x8, sx, f0.w8, f0.sw, w1_8=f1.w8 if meta.gated else None, sw1=f1.sw if meta.gated else None,
                a2=K.scale_rows(t, inv_x) if lora else None, b2=K.scale_rows(B0, f0.inv_sw) if lora else None,
                b2_1=K.scale_rows(B1, f1.inv_sw) if (lora and meta.gated) else None, alpha2=meta.lora_scale if lora else 1.0,
                bias=b0, bias1=b1 if meta.gated else None, residual=residual, counts=counts if meta.gated else None,
                out_dtype=meta.out_dtype or x.dtype)
        else:
            xp, W0p, W1p = _padk(x, W0, W1)
            y = K.gemm(
                xp, W0p, w1=W1p if meta.gated else None,
                a2=t, b2=B0 if lora else None, b2_1=B1 if (lora and meta.gated) else None, alpha2=meta.lora_scale if lora else 1.0,
                bias=b0, bias1=b1 if meta.gated else None, residual=residual,
                counts=counts if meta.gated else None, act=meta.act, out_dtype=meta.out_dtype, f32_split=meta.f32_split,
            )
        ctx.meta, ctx.lora = meta, lora
        ctx.save_for_backward(x, t, counts, W0, Wt0, b0, A0, B0, W1, Wt1, b1, A1, B1)
        ctx.has_residual = residual is not None
        if meta.fork:
            ctx.set_materialize_grads(False)
            return y, x_in.view_as(x_in)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy, dpass=None):
        meta: LinearMeta = ctx.meta
        x, t, counts, W0, Wt0, b0, A0, B0, W1, Wt1, b1, A1, B1 = ctx.saved_tensors
        if dy is None:                 # only the passed-through input was used downstream
            return (None, dpass) + (None,) * 12
        if dpass is not None:
            dpass = dpass if dpass.dtype == x.dtype else dpass.to(x.dtype)
            dpass = (dpass if dpass.is_contiguous() else dpass.contiguous()).view(x.shape)
        assert meta.act == hip.ACT_NONE, 'fused activations are forward-only; use the separate activation op when training'
        dy = dy if dy.is_contiguous() else dy.contiguous()
        if dy.dtype != x.dtype:
            dy = K.cast(dy, x.dtype)
        gated, lora, s = meta.gated, ctx.lora, meta.lora_scale
        cnt = counts if gated else None
        need = ctx.needs_input_grad  # (meta, x, residual, counts, W0, Wt0, b0, A0, B0, W1, Wt1, b1, A1, B1)
        g = [None] * 14
        u = None
        if lora and (need[1] or need[7] or need[12]):
            Bt0 = meta.Bt0 if meta.Bt0 is not None else _t(B0)
            Bt1 = (meta.Bt1 if meta.Bt1 is not None else _t(B1)) if gated else None
            u = _lora_project(dy, Bt0, Bt1, gated, counts)                            # [M, r] = dy · B
        if need[1] and meta.f8_0 is not None:
            # fp8 dgrad: dy rows quantised per token, W^T in e4m3 per input channel
            f0, f1 = meta.f8_0, meta.f8_1
            dy8, sdy, inv_dy = K.quant_rows_fp8(dy, counts[1:2] if gated else None)
            At0 = (meta.At0 if meta.At0 is not None else _t(A0)) if lora else None
            At1 = (meta.At1 if meta.At1 is not None else _t(A1)) if (lora and gated) else None
            g[1] = K.gemm_fp8(dy8, sdy, f0.wt8, f0.swt, w1_8=f1.wt8 if gated else None, sw1=f1.swt if gated else None,
                              a2=K.scale_rows(u, inv_dy) if lora else None, b2=K.scale_rows(At0, f0.inv_swt) if lora else None,
                              b2_1=K.scale_rows(At1, f1.inv_swt) if (lora and gated) else None, alpha2=s if lora else 1.0, counts=cnt,
                              drop_p=meta.drop_p if lora else 0.0, drop_seed=meta.drop_seed, out_dtype=x.dtype, residual=dpass)
        elif need[1] and Wt0 is None and NN_DGRAD and x.dtype == torch.bfloat16 and dy.shape[1] % 64 == 0 and x.shape[1] % 8 == 0:
            # dx = dy W with W as it sits in HBM (the 256-column kernel reads the weight through transposed LDS reads): no transposed
            # copy of the weight, resident or per use
            At0 = (meta.At0 if meta.At0 is not None else _t(A0)) if lora else None
            At1 = (meta.At1 if meta.At1 is not None else _t(A1)) if (lora and gated) else None
            g[1] = K.gemm(dy, W0.detach(), w1=W1.detach() if gated else None, b_nn=True, a2=u, b2=At0, b2_1=At1,
                          alpha2=s if lora else 1.0, counts=cnt, drop_p=meta.drop_p if lora else 0.0, drop_seed=meta.drop_seed,
                          residual=dpass)
        elif need[1]:
            wt0 = Wt0 if Wt0 is not None else K.transpose(W0.detach())
            wt1 = (Wt1 if Wt1 is not None else K.transpose(W1.detach())) if gated else None
            dyp, wt0, wt1 = _padk(dy, wt0, wt1)
            At0 = (meta.At0 if meta.At0 is not None else _t(A0)) if lora else None
            At1 = (meta.At1 if meta.At1 is not None else _t(A1)) if (lora and gated) else None
            g[1] = K.gemm(dyp, wt0, w1=wt1, a2=u, b2=At0, b2_1=At1,
                          alpha2=s if lora else 1.0, counts=cnt, drop_p=meta.drop_p if lora else 0.0, drop_seed=meta.drop_seed,
                          f32_split=meta.f32_split, residual=dpass)
        elif dpass is not None:
            g[1] = dpass
        if ctx.has_residual and need[2]:
            g[2] = dy
            if WGRAD_SIDE_STREAM and dy.is_cuda:
                _hold_until_backward_ends(dy)
        # parameter gradients contract over token rows
        experts = ((0, W0, b0, A0, B0, 4), (1, W1, b1, A1, B1, 9)) if gated else ((0, W0, b0, A0, B0, 4),)
        tn_ok = x.dtype == torch.bfloat16 and dy.shape[1] % 8 == 0 and x.shape[1] % 8 == 0
        # gated + LoRA on both experts: the two experts' factor gradients come out of one launch each (dB pair, dA pair)
        paired = (gated and lora and tn_ok and need[8] and need[13] and need[7] and need[12]
                  and K.tn_skinny_supported(dy, t) and K.tn_skinny_supported(x, u))
        if paired:
            g[8], g[13] = _lora_wgrad_pair((B0, B1), dy, t, False, cnt, s, 0.0, 0)
            g[7], g[12] = _lora_wgrad_pair((A0, A1), x, u, True, cnt, s, meta.drop_p, meta.drop_seed)
        for e, W, b, A, B, base in experts:
            seg = e if gated else -1
            if tn_ok:
                # row-contraction MFMA kernel on the row-major activations as they are: no transposed copies
                if need[base]:
                    g[base] = K.gemm_tn(dy, x, counts=cnt, segment=seg)
                if lora and need[base + 4] and not paired:                    # dB [N, r] = s · dy^T · t
                    g[base + 4] = _lora_wgrad(B, dy, t, False, cnt, seg, s, 0.0, 0)
                if lora and need[base + 3] and not paired:                    # dA [r, K] = s · u^T · drop(x)
                    g[base + 3] = _lora_wgrad(A, x, u, True, cnt, seg, s, meta.drop_p, meta.drop_seed)
            else:
                # fp32 islands / odd shapes: K-contiguous transposes feed the NT kernel
                def tr(z):
                    return K.transpose_segment(z, counts, e) if gated else K.transpose(z, pad_to=64)
                # full fp32 weight gradient of an ungated, LoRA-free linear whose gradient lives in a reduction bucket (the
                # unfrozen SAM / iSAM / vg_proj linears): W.grad += dy^T x through the GEMM's residual path, transposes
                # included, on the side stream — off the critical path, no temporary, no AccumulateGrad add
                W, b = _real_param(W), _real_param(b)
                wready = getattr(W, '_vm_grad_ready', None)
                bias_done = False
                if (need[base] and not gated and not lora and wready is not None and W.grad is not None and W.grad.dtype == torch.float32
                        and dy.dtype == torch.float32 and W.grad.is_contiguous() and W.grad.shape[1] % 4 == 0):
                    # the bias gradient (column sums of dy) rides along with the transpose of dy when its slot is an fp32 bucket view too
                    bready = getattr(b, '_vm_grad_ready', None) if (b is not None and need[base + 2]) else None
                    fuse_b = bready is not None and b.grad is not None and b.grad.dtype == torch.float32 and b.grad.is_contiguous()

                    def run(W=W, wready=wready, b=b, bready=bready, fuse_b=fuse_b):
                        if F32_TN_WGRAD and K.gemm_tn_f32_supported(dy, x, W.grad, meta.f32_split):
                            # TN form: dy and x as they are, the bias gradient from the tiles the kernel stages anyway
                            K.gemm_tn_f32(dy, x, W.grad, colsum_out=b.grad if fuse_b else None, f32_split=meta.f32_split)
                        else:
                            dyT = K.transpose(dy, pad_to=64, colsum_out=b.grad if fuse_b else None)
                            K.gemm(dyT, tr(x), out=W.grad, accumulate=True, f32_split=meta.f32_split)
                        wready(W)
                        if fuse_b:
                            bready(b)
                    _off_critical_path(run, dy.device, (dy, x))
                    need_w = False
                    bias_done = fuse_b
                else:
                    need_w = need[base]
                dyT = tr(dy) if (need_w or (lora and need[base + 4])) else None
                if need_w:
                    g[base] = K.gemm(dyT, tr(x), f32_split=meta.f32_split)
                if lora and need[base + 4]:
                    dB = K.gemm(dyT, tr(t))
                    g[base + 4] = dB if s == 1.0 else dB * s
                if lora and need[base + 3]:
                    xd = K.dropout(x, meta.drop_p, meta.drop_seed) if meta.drop_p > 0 else x
                    dA = K.gemm(tr(u), tr(xd))
                    g[base + 3] = dA if s == 1.0 else dA * s
            if b is not None and need[base + 2] and not (not tn_ok and bias_done):
                if gated:
                    dyT = K.transpose_segment(dy, counts, e)
                    ones = torch.ones(8, dyT.shape[1], dtype=dyT.dtype, device=dyT.device)
                    g[base + 2] = K.gemm(dyT, ones)[:, 0].to(b.dtype)
                else:
                    bready = getattr(b, '_vm_grad_ready', None)
                    if bready is not None and b.grad is not None and b.grad.dtype == torch.float32 and b.grad.is_contiguous():
                        def run_b(b=b, bready=bready):
                            K.colsum(dy, out=b.grad)          # atomically accumulated into the bucket slot
                            bready(b)
                        _off_critical_path(run_b, dy.device, (dy,))
                    else:
                        g[base + 2] = K.colsum(dy).to(b.dtype)
        return tuple(g)


def linear(x, W0, *, meta: LinearMeta | None = None, Wt0=None, b0=None, A0=None, B0=None,
           W1=None, Wt1=None, b1=None, A1=None, B1=None, residual=None, counts=None):
    meta = meta or LinearMeta()
    if _HELD_BY_TASK:
        _drop_stale_held()
    return _Linear.call(meta, x, residual, counts, W0, Wt0, b0, A0, B0, W1, Wt1, b1, A1, B1)


# ----------------------------------------------------------------------------- norms
def _norm_params_off_path(params, run_kernel, keep_alive):
    """Weight / bias gradients of a norm layer, when their slots live in a gradient bucket: computed on the side stream and
    added straight into the slots (fp32 slots: the kernel's atomics land there; bf16 slots: fp32 scratch, rounded, added —
    the same rounding as AccumulateGrad would apply). Returns True when it took care of them."""
    ps = [_real_param(p) for p in params if p is not None]
    if not ps or any(getattr(p, '_vm_grad_ready', None) is None or p.grad is None or not p.grad.is_contiguous() for p in ps):
        return False

    def run():
        if all(p.grad.dtype == torch.float32 for p in ps):
            run_kernel(*[p.grad for p in ps])
        elif all(getattr(p, '_vm_f32_acc', None) is not None for p in ps):
            # bf16 slots: the kernel's atomics land in fp32 side accumulators that the bucket folds into the slots with one
            # launch when it is reduced (ddp.f32_accumulator) — no zeros(), cast and add per parameter
            run_kernel(*[p._vm_f32_acc(p) for p in ps])
        else:
            outs = run_kernel(*[None for _ in ps])
            for p, g in zip(ps, outs):
                p.grad.add_(g.to(p.grad.dtype))
        for p in ps:
            p._vm_grad_ready(p)
    _off_critical_path(run, ps[0].device, keep_alive)
    return True


@_direct
class _RMSNorm(Function):
    """`fork`: also return x itself (a view) — the residual branch of a pre-norm block takes THAT, so both gradients of x arrive
    here and the kernel sums them (`dx_add`) instead of autograd adding them with one more pass over the activations"""

    @staticmethod
    def forward(ctx, x, w, eps, nrows, fork=False):
        y, rstd = K.rmsnorm_fwd(x, w, eps, nrows)
        ctx.save_for_backward(x, w, rstd, nrows)
        ctx.set_materialize_grads(False)
        return (y, x.view_as(x)) if fork else y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy, dpass=None):
        if dy is None:
            return dpass, None, None, None, None
        x, w, rstd, nrows = ctx.saved_tensors
        if dpass is not None and dpass.dtype != x.dtype:
            dpass = dpass.to(x.dtype)
        if ctx.needs_input_grad[1] and _norm_params_off_path(
                (w,), lambda dw_out: (K.rmsnorm_bwd(x, w, dy, rstd, nrows, need_dx=False, dw_out=dw_out)[1],), (x, dy, rstd, nrows)):
            return K.rmsnorm_bwd(x, w, dy, rstd, nrows, need_dw=False, dx_add=dpass)[0], None, None, None, None
        dx, dw = K.rmsnorm_bwd(x, w, dy, rstd, nrows, need_dw=ctx.needs_input_grad[1], dx_add=dpass)
        return dx, (dw.to(w.dtype) if dw is not None else None), None, None, None


def rms_norm(x, w, eps: float, nrows=None, fork: bool = False):
    """`fork`: -> (y, x passed through) — use the second output as the block's residual"""
    if fork and not FORK:
        return _RMSNorm.call(x, w, eps, nrows, False), x
    return _RMSNorm.call(x, w, eps, nrows, fork)


@_direct
class _LayerNorm(Function):
    @staticmethod
    def forward(ctx, x, w, b, eps, residual, fork=False):
        y, mean, rstd = K.layernorm_fwd(x, w, b, eps, residual)
        ctx.save_for_backward(x, w, mean, rstd)
        ctx.has_res = residual is not None
        ctx.bias = b          # only its gradient slot is touched in backward (no value needed): not saved as a tensor
        ctx.set_materialize_grads(False)
        return (y, x.view_as(x)) if fork else y      # (fork: see _RMSNorm)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy, dpass=None):
        if dy is None:
            return dpass, None, None, None, None, None
        x, w, mean, rstd = ctx.saved_tensors
        if dpass is not None and dpass.dtype != x.dtype:
            dpass = dpass.to(x.dtype)
        need_dw = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        b = ctx.bias
        if ctx.needs_input_grad[1] and ctx.needs_input_grad[2] and b is not None and _norm_params_off_path(
                (w, b), lambda dw_out, db_out: K.layernorm_bwd(x, w, dy, mean, rstd, need_dx=False, dw_out=dw_out, db_out=db_out)[1:],
                (x, dy, mean, rstd)):
            if ctx.has_res and dy.is_cuda and WGRAD_SIDE_STREAM:
                _hold_until_backward_ends(dy)       # dy doubles as the residual's gradient while the side stream still reads it
            return (K.layernorm_bwd(x, w, dy, mean, rstd, need_dw=False, dx_add=dpass)[0], None, None, None,
                    (dy if ctx.has_res else None), None)
        dx, dw, db = K.layernorm_bwd(x, w, dy, mean, rstd, need_dw=need_dw, dx_add=dpass)
        return (dx, dw.to(w.dtype) if dw is not None else None, db.to(w.dtype) if db is not None else None, None,
                dy if ctx.has_res else None, None)


def layer_norm(x, w, b, eps: float = 1e-5, residual=None, fork: bool = False):
    """residual + LayerNorm(x) (residual optional); `fork`: -> (y, x passed through) for a pre-norm block's residual"""
    if fork and not FORK:
        return _LayerNorm.call(x, w, b, eps, residual, False), x
    return _LayerNorm.call(x, w, b, eps, residual, fork)


# ----------------------------------------------------------------------------- activations
@_direct
class _SiluMul(Function):
    @staticmethod
    def forward(ctx, gate, up):
        ctx.save_for_backward(gate, up)
        return K.silu_mul(gate, up)

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        gate, up = ctx.saved_tensors
        return K.silu_mul_bwd(gate, up, dout)


def silu_mul(gate, up):
    return _SiluMul.call(gate, up)


@_direct
class _Gelu(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return K.gelu(x)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return K.gelu_bwd(x, dy)


def gelu(x):
    return _Gelu.call(x)


@_direct
class _Relu(Function):
    """relu applied by the GEMM epilogue is not differentiable there; this is the standalone op"""
    @staticmethod
    def forward(ctx, x):
        y = torch.clamp_min(x, 0)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return K.relu_bwd(y, dy)


def relu(x):
    return _Relu.call(x)


# ----------------------------------------------------------------------------- RoPE (in place on the packed qkv buffer)
@_direct
class _Rope(Function):
    @staticmethod
    def forward(ctx, qkv, row_pos, cos, sin, n_heads, head_dim, nrows):
        ctx.mark_dirty(qkv)
        K.rope_(qkv, row_pos, cos, sin, n_heads, head_dim, False, nrows)
        ctx.save_for_backward(row_pos, cos, sin, nrows)
        ctx.dims = (n_heads, head_dim)
        return qkv

    @staticmethod
    @once_differentiable
    def backward(ctx, d):
        row_pos, cos, sin, nrows = ctx.saved_tensors
        d = d.clone() if not d.is_contiguous() else d.clone()
        K.rope_(d, row_pos, cos, sin, *ctx.dims, True, nrows)
        return d, None, None, None, None, None, None


def rope_(qkv, row_pos, cos, sin, n_heads: int, head_dim: int, nrows=None):
    return _Rope.call(qkv, row_pos, cos, sin, n_heads, head_dim, nrows)


# ----------------------------------------------------------------------------- var-len attention (bf16)
@_direct
class _Attention(Function):
    """`rope` = (row_pos, cos, sin, nrows): q and k are rotated IN PLACE on the packed qkv buffer before the attention (the standalone
    `_Rope` node's job) and the inverse rotation is applied in place to this node's own dqkv in the backward — no copy of the incoming
    gradient (the standalone node must clone it: 100 MB per decoder layer) and one autograd node less per layer."""

    @staticmethod
    def forward(ctx, qkv, cu_seqlens, row_of_pos, max_seqlen, n_heads, head_dim, scale, causal, total_pos_max, rope=None):
        hdim = n_heads * head_dim
        if rope is not None:
            row_pos, cos, sin, nrows = rope
            ctx.mark_dirty(qkv)
            K.rope_(qkv, row_pos, cos, sin, n_heads, head_dim, False, nrows)
        q, k, v = qkv[:, :hdim], qkv[:, hdim:2 * hdim], qkv[:, 2 * hdim:]
        out, lse = K.attn_fwd(q, k, v, cu_seqlens, max_seqlen, n_heads, head_dim, scale, causal, row_of_pos, total_pos_max)
        ctx.save_for_backward(qkv, out, lse, cu_seqlens, row_of_pos, *(rope if rope is not None else ()))
        ctx.cfg = (max_seqlen, n_heads, head_dim, scale, causal, total_pos_max, rope is not None)
        if rope is not None:
            ctx.set_materialize_grads(False)
            return out, qkv                 # (the rotated buffer is an output because it was modified in place)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout, dqkv_direct=None):
        max_seqlen, n_heads, head_dim, scale, causal, total_pos_max, roped = ctx.cfg
        saved = ctx.saved_tensors               # (once: checkpointing unpacks on access)
        qkv, out, lse, cu_seqlens, row_of_pos = saved[:5]
        hdim = n_heads * head_dim
        q, k, v = qkv[:, :hdim], qkv[:, hdim:2 * hdim], qkv[:, 2 * hdim:]
        dqkv = K.attn_bwd(q, k, v, out, lse, dout, cu_seqlens, max_seqlen, n_heads, head_dim, scale, causal, row_of_pos,
                          total_pos_max).view(qkv.shape)
        if roped:
            if dqkv_direct is not None:      # somebody differentiated through the rotated buffer itself (the training step never does)
                dqkv = dqkv + dqkv_direct
            row_pos, cos, sin, nrows = saved[5:]
            K.rope_(dqkv, row_pos, cos, sin, n_heads, head_dim, True, nrows)
        return dqkv, None, None, None, None, None, None, None, None, None


def attention(qkv, cu_seqlens, max_seqlen: int, n_heads: int, head_dim: int, scale: float, causal: bool,
              row_of_pos=None, total_pos_max: int | None = None, rope=None):
    """qkv: [rows, 3*H*hd] packed (q | k | v) -> [rows, H*hd]. `rope` = (row_pos, cos, sin, nrows): rotate q / k in place first (and
    un-rotate the gradient in the backward) — `rope_` + `attention` in one autograd node.
    fp32 qkv (the towers' "32-true" mode, reference mmmm.py:468-492 without MyPrecision): the exact-f32 kernels of the fp32 islands,
    head widths 112 / 128, causal mask and the packed layout's position table included (vm_attn_*_f32)."""
    if qkv.dtype == torch.float32:
        if rope is not None:
            qkv = rope_(qkv, *rope[:3], n_heads, head_dim, rope[3])
        return self_attention_f32(qkv, n_heads, head_dim, scale, cu_seqlens, max_seqlen, 1, causal, row_of_pos)
    if rope is not None:
        return _Attention.call(qkv, cu_seqlens, row_of_pos, max_seqlen, n_heads, head_dim, scale, causal, total_pos_max, tuple(rope))[0]
    return _Attention.call(qkv, cu_seqlens, row_of_pos, max_seqlen, n_heads, head_dim, scale, causal, total_pos_max)


# ----------------------------------------------------------------------------- fp32 attention (SAM islands)
@_direct
class _AttentionF32(Function):
    @staticmethod
    def forward(ctx, q, k, v, n_heads, head_dim, scale, cu_seqlens, max_seqlen):
        q, k, v = (t if t.stride(-1) == 1 else t.contiguous() for t in (q, k, v))
        out, lse = K.attn_f32_fwd(q, k, v, n_heads, head_dim, scale, cu_seqlens, max_seqlen)
        ctx.save_for_backward(q, k, v, out, lse, cu_seqlens)
        ctx.cfg = (n_heads, head_dim, scale, max_seqlen)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        q, k, v, out, lse, cu = ctx.saved_tensors
        n_heads, head_dim, scale, max_seqlen = ctx.cfg
        dq, dk, dv = K.attn_f32_bwd(q, k, v, out, lse, dout, n_heads, head_dim, scale, cu, max_seqlen)
        return dq, dk, dv, None, None, None, None, None


@_direct
class _SelfAttentionF32(Function):
    """fp32 self-attention on a packed qkv [T, 3*H*hd] (SAM ViT-B, image_encoder.py:126-136): q/k/v are strided views in
    both directions and the gradient is ONE dqkv tensor written in place by the kernels — no contiguous copies of the
    thirds, no zero-filled slice gradients, no accumulation adds."""
    @staticmethod
    def forward(ctx, qkv, n_heads, head_dim, scale, cu_seqlens, max_seqlen, f32_split=0, causal=False, row_of_pos=None):
        Cw = n_heads * head_dim
        qkv = qkv if qkv.stride(-1) == 1 else qkv.contiguous()
        out, lse = K.attn_f32_fwd(qkv[:, :Cw], qkv[:, Cw:2 * Cw], qkv[:, 2 * Cw:], n_heads, head_dim, scale, cu_seqlens, max_seqlen,
                                  f32_split=f32_split, causal=causal, row_of_pos=row_of_pos)
        ctx.save_for_backward(qkv, out, lse, cu_seqlens, row_of_pos)
        ctx.cfg = (n_heads, head_dim, scale, max_seqlen, f32_split, causal)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        qkv, out, lse, cu, rop = ctx.saved_tensors
        n_heads, head_dim, scale, max_seqlen, f32_split, causal = ctx.cfg
        Cw = n_heads * head_dim
        # every row belongs to a sequence: the dq / dkv kernels write all of it (not so in the indirect layout: rows past the valid ones)
        dqkv = torch.empty_like(qkv) if rop is None else torch.zeros_like(qkv)
        K.attn_f32_bwd(qkv[:, :Cw], qkv[:, Cw:2 * Cw], qkv[:, 2 * Cw:], out, lse, dout, n_heads, head_dim, scale, cu, max_seqlen,
                       grads=(dqkv[:, :Cw], dqkv[:, Cw:2 * Cw], dqkv[:, 2 * Cw:]), f32_split=f32_split, causal=causal, row_of_pos=rop)
        return dqkv, None, None, None, None, None, None, None, None


def self_attention_f32(qkv, n_heads: int, head_dim: int, scale: float, cu_seqlens, max_seqlen, f32_split: int = 0,
                       causal: bool = False, row_of_pos=None):
    """`f32_split` (head_dim 64): arithmetic of the attention products — 0 exact f32 MFMA, 2 / 3 split-bf16 with 3 / 6 products
    (kernels.attn_f32_fwd); the image encoders pass their blocks' `f32_split` (image_encoder.ENCODER_F32_SPLIT).
    `causal` / `row_of_pos`: the towers' fp32 mode (exact arithmetic)"""
    return _SelfAttentionF32.call(qkv, n_heads, head_dim, scale, cu_seqlens, max_seqlen, f32_split, causal, row_of_pos)


_F32_HEAD_DIMS = (8, 16, 32, 48, 64, 96, 112, 128)       # instantiations of attn_f32_*_k (csrc/attn_f32.hip)


def attention_f32(q, k, v, n_heads: int, head_dim: int, scale: float, cu_seqlens=None, max_seqlen=None):
    """dense batched [Bn, L, H*hd] or packed var-len [T, H*hd] (+cu_seqlens) fp32 attention"""
    if head_dim not in _F32_HEAD_DIMS and head_dim < _F32_HEAD_DIMS[-1]:
        # odd head widths (only tiny test models: the reference fixtures' SAM has 8 decoder heads over 32 / 16 channels): every head
        # is zero-padded to the next width the kernels are instantiated for — scores and outputs are unchanged, `scale` stays that
        # of the true width
        hp = next(h for h in _F32_HEAD_DIMS if h >= head_dim)

        def pad(t):
            return torch.nn.functional.pad(t.reshape(*t.shape[:-1], n_heads, head_dim), (0, hp - head_dim)).reshape(*t.shape[:-1], n_heads * hp)
        out = _AttentionF32.call(pad(q), pad(k), pad(v), n_heads, hp, scale, cu_seqlens, max_seqlen)
        return out.reshape(*out.shape[:-1], n_heads, hp)[..., :head_dim].reshape(*out.shape[:-1], n_heads * head_dim)
    return _AttentionF32.call(q, k, v, n_heads, head_dim, scale, cu_seqlens, max_seqlen)


# ----------------------------------------------------------------------------- rows: embedding / gather / scatter
@_direct
class _EmbeddingRows(Function):
    """out[r] = weight[ids[r]] (ids < 0 -> zeros); weight gradient by the segmented-sum kernel"""
    @staticmethod
    def forward(ctx, weight, ids):
        ctx.save_for_backward(ids)
        ctx.wshape, ctx.wdtype = weight.shape, weight.dtype
        return K.gather_rows(weight, ids)

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        (ids,) = ctx.saved_tensors
        dw = torch.zeros(ctx.wshape, dtype=ctx.wdtype, device=dout.device)
        rows = torch.arange(ids.numel(), dtype=torch.int32, device=dout.device)
        K.embedding_bwd(dout.contiguous(), ids, rows, dw)
        return dw, None


def embedding_rows(weight, ids):
    return _EmbeddingRows.call(weight, ids)


@_direct
class _GatherRows(Function):
    @staticmethod
    def forward(ctx, src, idx, nrows_out):
        ctx.save_for_backward(idx)
        ctx.n_src = src.shape[0]
        return K.gather_rows(src, idx, rows=nrows_out)

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        d = torch.zeros(ctx.n_src, dout.shape[1], dtype=dout.dtype, device=dout.device)
        K.scatter_rows(dout.contiguous(), idx, d)   # idx must be injective where >= 0
        return d, None, None


def gather_rows(src, idx, nrows_out: int | None = None):
    """out[r] = src[idx[r]]; idx must not repeat a source row (permutation-like maps)"""
    return _GatherRows.call(src, idx, nrows_out)


@_direct
class _OverwriteRows(Function):
    """base[idx[r]] = src[r] in place (image features into the embedded sequence, modeling_cogvlm.py:451-453)"""
    @staticmethod
    def forward(ctx, base, src, idx):
        ctx.mark_dirty(base)
        K.scatter_rows(src.contiguous(), idx, base)
        ctx.save_for_backward(idx)
        return base

    @staticmethod
    @once_differentiable
    def backward(ctx, d):
        (idx,) = ctx.saved_tensors
        dsrc = K.gather_rows(d.contiguous(), idx)
        # rows that were overwritten carry no gradient to `base`; the caller guarantees base rows there came
        # from ids == -1 (zero rows) so the embedding backward ignores them
        return d, dsrc, None


def overwrite_rows_(base, src, idx):
    return _OverwriteRows.call(base, src, idx)


# ----------------------------------------------------------------------------- trilinear up-sampling of mask logits


@_direct
class _UpsampleTrilinear(Function):
    @staticmethod
    def forward(ctx, x, size):
        ctx.in_shape = tuple(x.shape[1:])
        return K.upsample_trilinear3d(x, size)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        return K.upsample_trilinear3d_bwd(gy.contiguous(), ctx.in_shape), None


def upsample_trilinear(x: torch.Tensor, size) -> torch.Tensor:
    """F.interpolate(x, size, mode='trilinear') for fp32 device volumes [..., d, h, w] (segvol/modeling/sam.py:57-87) through the HIP
    kernels (gather-form, deterministic backward). No other path: anything else raises, like every wrapper of this module."""
    size = tuple(int(v) for v in size)
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() >= 4):
        raise TypeError(f'upsample_trilinear: fp32 device tensor [..., d, h, w] expected, got {x.dtype} {tuple(x.shape)} on {x.device}')
    lead = x.shape[:-3]
    if x.numel() == 0:                         # no prompts: nothing to interpolate, the result is empty too
        return x.new_zeros(*lead, *size) + 0 * x.sum()
    y = _UpsampleTrilinear.call(x.reshape(-1, *x.shape[-3:]).contiguous(), size)
    return y.view(*lead, *size)


# ----------------------------------------------------------------------------- Dice + focal loss of mask logits
@_direct
class _DiceFocal(Function):
    """per-row Dice loss and per-row SUM of the sigmoid-focal loss of fp32 mask logits (mmmm/models/loss.py:32-56) in one
    streaming pass each way instead of ~40 element-wise / reduction launches"""
    @staticmethod
    def forward(ctx, x, target, gamma, alpha):
        sums, out = K.dice_focal_fwd(x, target, gamma, alpha)
        ctx.save_for_backward(x, target, sums)
        ctx.cfg = (gamma, alpha)
        return out[:, 0], out[:, 1]

    @staticmethod
    @once_differentiable
    def backward(ctx, g_dice, g_focal):
        x, target, sums = ctx.saved_tensors
        gamma, alpha = ctx.cfg
        gd = g_dice.float().contiguous() if g_dice is not None else None
        gf = g_focal.float().contiguous() if g_focal is not None else None
        return K.dice_focal_bwd(x, target, gamma, alpha, sums, gd, gf), None, None, None


def dice_focal(x, target, gamma: float, alpha: float | None):
    """x fp32 [R, n] logits, target uint8 [R, n] | None -> (dice [R], focal_sum [R])"""
    return _DiceFocal.call(x, target, gamma, alpha)


# ----------------------------------------------------------------------------- hyper-network mask product (SAM mask decoder)
@_direct
class _InstanceLoss(Function):
    """the instance losses of one sample in one launch each way (kernels.instance_loss_fwd / _bwd)"""

    @staticmethod
    def forward(ctx, logit, reg, label, match, gamma, alpha):
        out = K.instance_loss_fwd(logit, reg, label, match, gamma, alpha)
        ctx.save_for_backward(logit, reg, label, match, out)
        ctx.gamma, ctx.alpha = gamma, alpha
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        logit, reg, label, match, out = ctx.saved_tensors
        d_logit, d_reg = K.instance_loss_bwd(logit, reg, label, match, ctx.gamma, ctx.alpha, out, g.contiguous())
        return d_logit, d_reg, None, None, None, None


def instance_loss(logit, reg, label, match, gamma: float, alpha: float | None):
    """logit fp32 [nt, nq], reg fp32 [nt, 1 + nq, 6], label fp32 [n_boxes, 6], match int64 [nt, nq] (matched label box | < 0)
    -> fp32 [6] = (focal mean over all entries, focal mean matched vs 1 [log], focal mean unmatched vs 0 [log], l1 mean of matched
    pairs, 1 - mean GIoU of matched pairs, matched count); gradients flow to logit and reg through entries 0, 3, 4"""
    return _InstanceLoss.call(logit, reg, label, match, gamma, alpha)


@_direct
class _HyperProduct(Function):
    """y[p] = up[p] @ w[p]^T for every prompt p: up [P, V, C] fp32 (up-scaled image embedding per prompt, channel-last), w [P, M, C]
    (per-prompt hyper-network weights) -> y [P, V, M]. Reference mask_decoder.py:139-147 (`einsum('n m c, n c ... -> n m ...')`).
    One autograd node for all prompts: the per-prompt GEMMs write into slices of ONE output / gradient buffer, so no prompt-wise
    indexing reaches autograd (whose slice backward zero-fills and adds a full `up`-sized tensor per prompt: 16 x 77 MB per head and
    step at 448 x 448)."""
    @staticmethod
    def forward(ctx, up, w):
        P, V, C = up.shape
        M = w.shape[1]
        Mp = (M + 3) // 4 * 4
        y = torch.empty(P, V, Mp, dtype=up.dtype, device=up.device)
        upp, wp = _padk(up.reshape(P * V, C), w.reshape(P * M, C))
        upp, wp = upp.view(P, V, -1), wp.view(P, M, -1)
        for n in range(P):
            K.gemm(upp[n], wp[n], out=y[n][:, :M])
        ctx.save_for_backward(up, w)
        return y[..., :M]

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        up, w = ctx.saved_tensors
        P, V, C = up.shape
        M = w.shape[1]
        dy = dy if dy.is_contiguous() else dy.contiguous()
        d_up = d_w = None
        if ctx.needs_input_grad[0]:
            d_up = torch.empty_like(up)
            wt = w.transpose(1, 2).contiguous()                    # [P, C, M]: tiny
            for n in range(P):
                dyp, wtp = _padk(dy[n], wt[n])
                K.gemm(dyp, wtp, out=d_up[n])
        if ctx.needs_input_grad[1]:
            d_w = torch.zeros_like(w)              # (tiny; the split-K product of a V-long contraction accumulates into it)
            for n in range(P):
                K.gemm(K.transpose(dy[n], pad_to=64), K.transpose(up[n], pad_to=64), out=d_w[n], out_is_zero=True)
        return d_up, d_w


def hyper_product(up, w):
    return _HyperProduct.call(up, w)


# ----------------------------------------------------------------------------- weighted CE over the vocabulary
@_direct
class _WeightedCE(Function):
    """loss = sum_r ce_r * w_r / n_valid  (modeling_cogvlm.py:610-627); also returns the per-row CE (no grad)."""
    @staticmethod
    def forward(ctx, logits, labels, weight, vocab, nrows):
        row_loss, lse = K.ce_fwd(logits, labels, vocab, nrows)
        valid = labels >= 0
        n_valid = valid.sum().clamp_min(1).to(torch.float32)
        w = torch.where(valid, weight.to(torch.float32), torch.zeros((), device=logits.device))
        loss = torch.dot(row_loss, w) / n_valid
        ctx.save_for_backward(logits, labels, lse, w / n_valid, nrows)
        ctx.vocab = vocab
        ctx.mark_non_differentiable(row_loss)
        return loss, row_loss

    @staticmethod
    @once_differentiable
    def backward(ctx, dloss, _):
        logits, labels, lse, scale, nrows = ctx.saved_tensors
        d = K.ce_bwd(logits, labels, lse, scale * dloss.to(torch.float32), ctx.vocab, nrows)
        return d, None, None, None, None


def weighted_ce(logits, labels, weight, vocab: int, nrows=None):
    return _WeightedCE.call(logits, labels, weight, vocab, nrows)


# ----------------------------------------------------------------------------- patch embedding
@_direct
class _Im2Col(Function):
    """[C,D,H,W] image -> [n_patch, C*pz*py*px] (no gradient to the image: inputs are data)"""
    @staticmethod
    def forward(ctx, image, patch):
        return K.im2col3d(image, patch)

    @staticmethod
    def backward(ctx, d):
        return None, None


def im2col3d(image, patch):
    return _Im2Col.call(image, tuple(patch))
