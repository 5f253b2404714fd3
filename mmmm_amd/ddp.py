"""Data-parallel gradient exchange for the VividMed step: bucketed all-reduce over RCCL/xGMI, overlapped with backward.

Replaces what the reference gets from Lightning's DDPStrategy(gradient_as_bucket_view=True,
broadcast_buffers=False) (conf/phase-*/fit.yaml:11-15). One process per GPU; `torch.distributed`'s "nccl"
backend is RCCL on ROCm. Only the trainable set is exchanged (LoRA factors + modules_to_save: ~791 M
parameters = 1.58 GB bf16); the frozen 17.6 B base weights are replicated.

Design for MI355X xGMI (fully connected, 7 links x ~153 GB/s per GPU): few LARGE buckets (default 256 MiB) so
each collective runs at link bandwidth, filled in the order in which BACKWARD PRODUCES the gradients so that a bucket is
complete — and its all-reduce is enqueued on RCCL's stream — as soon as the backward of the layers it covers has finished.
That order is NOT the reverse of the registration order for this model: `CogVLMModel` registers `embed_tokens, layers, norm,
vision` and `MMMMForCausalLM` adds `lm_head, sam, isam_model, vg_proj`, while the data flows vision -> embed -> layers -> norm
-> lm_head -> heads. `grad_production_order(model)` states the order from the module structure (heads, lm_head, norm, decoder
31..0, embed_tokens, GLU adapter, ViT-E 62..0, patch embedding: SURVEY §8e) and is what bench.py / the Lightning shell pass
(`order='given'`); `order='reverse'` (reverse registration, torch DDP's first-iteration heuristic) stays the default for plain
networks. Every step records the order in which slots became final (`ready_log`) and when each bucket was launched
(`launch_log`), so `exposed_report()` can say how many bucket bytes were issued after backward's last gradient — the quantity
the overlap is judged on — without any hardware (tests/test_ddp_cpu.py). The last bucket of a key is kept small (`tail_bytes`):
whatever completes with the step's LAST gradient (ViT-E layer 0 / the patch embedding) cannot overlap with anything.
Gradients live directly inside the flat bucket buffers (param.grad is a view), so there is no pack/unpack copy.
The 1/world averaging is not a pass of its own: `finish()` leaves the SUM in the buckets and `grad_scale` = 1/world pending;
`optim.FlatAdamW` folds it into the clip coefficient its kernel applies anyway (`clip_grad_norm_` folds it likewise).

Unused-parameter hazard (reference mmmm.py:263-278 runs dummy SAM forwards so that every rank produces every
gradient): here every bucket is reduced every step in a fixed order; a parameter that received no gradient
contributes its zero-filled slot, so ranks can never disagree on the collective sequence.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import torch
import torch.distributed as dist

from . import hip
from .param import no_weight_decay


def production_stages(model) -> list:
    """Trainable parameters of an `MMMMForCausalLM` (or any module tree with the same child names) grouped into the stages whose
    gradients backward produces together, in production order: stages in REVERSE forward order, inside a stage reverse registration
    order (~ reverse use order). Forward order (mmmm.py:296-352 -> modeling_cogvlm.py:659-710, visual.py:193-210):
    vision.patch_embedding, vision.transformer.layers 0..62, the rest of vision (GLU adapter, boi / eoi), embed_tokens, layers 0..31,
    norm, lm_head, then the grounding heads (vg_proj feeds isam_model, then sam: mmmm.py `visual_grounding`). Parameters outside
    these stages keep reverse registration order at the end."""
    stages = []                                    # forward order

    def sub(root, path):
        m = root
        for name in path.split('.'):
            m = getattr(m, name, None)
            if m is None:
                return None
        return m

    core = sub(model, 'model') or model
    vis = sub(core, 'vision')
    if vis is not None:
        pe = sub(vis, 'patch_embedding')
        if pe is not None:
            stages.append(list(pe.parameters()))
        tl = sub(vis, 'transformer.layers')
        if tl is not None:
            stages.extend(list(l.parameters()) for l in tl)
        stages.append(list(vis.parameters()))      # what is left of the tower (deduplicated below)
    m = sub(core, 'embed_tokens')
    if m is not None:
        stages.append(list(m.parameters()))
    dl = sub(core, 'layers')
    if dl is not None:
        stages.extend(list(l.parameters()) for l in dl)
    for root, name in ((core, 'norm'), (model, 'lm_head'), (model, 'vg_proj'), (model, 'isam_model'), (model, 'sam')):
        m = sub(root, name)
        if m is not None:
            stages.append(list(m.parameters()))
    seen, fwd = set(), []
    for st in stages:
        st = [p for p in st if id(p) not in seen]
        seen.update(id(p) for p in st)
        fwd.append(st)
    rest = [p for p in model.parameters() if id(p) not in seen]
    out = [list(reversed(st)) for st in reversed(fwd)] + [list(reversed(rest))]
    return [[p for p in st if p.requires_grad] for st in out]


def grad_production_order(model) -> list:
    """`production_stages` flattened: the `params` argument of `BucketedGradAllReduce(..., order='given')`"""
    return [p for st in production_stages(model) for p in st]


def plan_buckets(seq, bucket_bytes: int = 256 << 20, tail_bytes: int | None = None) -> list:
    """[(key, [params])] for parameters `seq` GIVEN IN GRADIENT-PRODUCTION ORDER; works on meta tensors (no allocation).
    key = (dtype, device, decayed?): decayed and undecayed parameters (param.NoWeightDecayParameter) never share a bucket, so the
    fused AdamW launch of a bucket has ONE decay value. A bucket is closed when the next parameter of its key would overflow it;
    buckets are ordered by the production index of their LAST parameter — the moment they complete — so launching them strictly
    in list order never holds a finished bucket behind an unfinished one (the fp32 heads' remainder bucket completes early in
    backward although it is closed last). The final bucket of every key is split so that its trailing part is <= tail_bytes."""
    if tail_bytes is None:
        tail_bytes = max(bucket_bytes // 4, 1)
    nbytes = lambda p: (p.numel() + 7) // 8 * 8 * p.element_size()
    cur: dict = {}
    closed: list = []                                # (completion index, key, params)
    for i, p in enumerate(seq):
        key = (p.dtype, p.device, not no_weight_decay(p))
        lst, size = cur.get(key, ([], 0))
        if lst and size + nbytes(p) > bucket_bytes:
            closed.append((lst[-1][0], key, [q for _, q in lst]))
            lst, size = [], 0
        lst.append((i, p))
        cur[key] = (lst, size + nbytes(p))
    for key, (lst, size) in cur.items():
        if not lst:
            continue
        if size > tail_bytes and len(lst) > 1:      # split: [head][tail <= tail_bytes] (at least the last parameter)
            acc, cut = 0, len(lst) - 1
            for j in range(len(lst) - 1, 0, -1):
                if acc + nbytes(lst[j][1]) > tail_bytes and j < len(lst) - 1:
                    break
                acc += nbytes(lst[j][1])
                cut = j
            closed.append((lst[cut - 1][0], key, [q for _, q in lst[:cut]]))
            lst = lst[cut:]
        closed.append((lst[-1][0], key, [q for _, q in lst]))
    closed.sort(key=lambda t: t[0])
    return [(key, lst) for _, key, lst in closed]


@dataclass(eq=False)
class _Bucket:
    buffer: torch.Tensor
    params: list = field(default_factory=list)
    decay: bool = True                            # False: every slot is a NoWeightDecayParameter (optim.FlatAdamW)
    pending: int = 0
    launched: bool = False
    work: object = None
    streams: set = field(default_factory=set)     # device streams that produced gradients of this bucket in this step
    last_stream: object = None                    # the stream added last (hip.current_stream_obj: one object per stream, compared with `is`)


_POST_HOOK_ON_NONE: list = []


def _post_hook_fires_on_none() -> bool:
    """Does this torch run a parameter's post-accumulate-grad hook when every use handed back None (the fused kernels accumulate
    straight into the bucket view)? A three-element CPU experiment, once per process."""
    if not _POST_HOOK_ON_NONE:
        class _Probe(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x, w):
                return x * 2

            @staticmethod
            def backward(ctx, g):
                return g * 2, None

        fired = []
        with torch.enable_grad():
            w = torch.nn.Parameter(torch.ones(3))
            x = torch.ones(3, requires_grad=True)
            h = w.register_post_accumulate_grad_hook(lambda p: fired.append(1))
            _Probe.apply(x, w).sum().backward()
            h.remove()
        _POST_HOOK_ON_NONE.append(bool(fired))
    return _POST_HOOK_ON_NONE[0]


class BucketedGradAllReduce:
    """Readiness protocol. A slot is READY when autograd has finished the parameter's AccumulateGrad node, which the engine
    runs exactly once per backward pass, after EVERY use of the parameter has run its backward — also when all of them
    handed back `None` because their kernels accumulated straight into the bucket view (functional._lora_wgrad, the fp32
    weight / bias / norm gradients). So a parameter that is used several times per step (iSAM's box / discriminator heads run
    once per sample, the mask decoder once per grid group) is counted once, after its last use. The fused kernels' own callback
    `p._vm_grad_ready(p)` only REGISTERS the stream the kernel ran on; it never counts."""

    def __init__(self, params, process_group=None, bucket_bytes: int = 256 << 20, world_size: int | None = None,
                 force_collectives: bool = False, sync_params: bool = True, order: str = 'reverse', tail_bytes: int | None = None,
                 defer_average: bool = False, registration=None):
        """`order`: 'reverse' = `params` are in registration order, gradients are expected in the reverse of it; 'given' = `params`
        are already in gradient-production order (`grad_production_order(model)`). `defer_average`: finish() leaves the SUM over
        ranks in the buckets and `grad_scale` = 1/world for the consumer to fold into its own pass (FlatAdamW sets this).
        `registration` (with order='given'): the same parameters in the order `model.parameters()` yields them — the numbering of
        optimizer state dicts (`FlatAdamW.state_dict` / `load_state_dict` number by it, so that a checkpoint of `torch.optim.AdamW(
        model.parameters())`, or of a run with another bucket order, lands on the right parameters: LoRA factors of different layers
        have identical shapes, a shape check cannot catch a permutation). Defaults to `params` as given."""
        assert order in ('reverse', 'given')
        self.params = [p for p in params if p.requires_grad]
        self.registration = self.params if registration is None else [p for p in registration if p.requires_grad]
        if registration is not None and ({id(p) for p in self.registration} != {id(p) for p in self.params} or len(self.registration) != len(self.params)):
            raise ValueError('`registration` must list exactly the trainable parameters of `params`')
        self._seq = list(reversed(self.params)) if order == 'reverse' else list(self.params)
        self.tail_bytes = tail_bytes
        self.defer_average = defer_average
        self.grad_scale = 1.0                     # pending factor on the bucket contents (1/world after a deferred finish())
        self.ready_log: list = []                 # ids of the parameters in the order their slots became final (this step)
        self.launch_log: list = []                # (bucket index, len(ready_log) at launch, launched by finish()?)
        self.pg = process_group
        if world_size is None:
            world_size = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.world_size = world_size
        self.collectives = world_size > 1 or force_collectives      # force: run the all-reduce even on one rank (tests)
        self.buckets: list[_Bucket] = []
        self._bucket_of: dict[int, int] = {}
        self._next = 0
        self._ready: set[int] = set()             # ids of the parameters whose slot is final in this step
        self._step_streams: set = set()           # every stream that produced or joined gradients in this step (finish() joins them)
        self._build(bucket_bytes)
        self._f32_acc: dict[int, torch.Tensor] = {}                  # id(param) -> fp32 side accumulator (f32_accumulator)
        self._f32_chunks: list = []                                   # [(buffer, used)] the accumulators are carved from
        self._f32_entries: dict[int, list] = {}                       # bucket index -> [(dst ptr, src ptr, count)]
        self._f32_tables: dict[int, torch.Tensor] = {}                # bucket index -> device table (rebuilt when entries are added)
        self._hooks = []
        self._acc_nodes = []                      # the hooks live on the AccumulateGrad nodes: keep the nodes alive
        assert torch.is_grad_enabled(), 'BucketedGradAllReduce must be built with autograd enabled (it hooks the AccumulateGrad nodes)'
        if self.world_size > 1 and sync_params:
            self.sync_parameters()
        need_prehook = not _post_hook_fires_on_none()
        for p in self.params:
            acc = p.view_as(p).grad_fn.next_functions[0][0]
            self._acc_nodes.append(acc)
            if need_prehook:
                # every use returned None (direct accumulation): older engines skip the post-accumulate hook then. (Probed once per
                # process: where the post hook fires anyway the prehook would only double the ~1 800 Python calls per backward pass.)
                self._hooks.append(acc.register_prehook(lambda grads, p=p: self._mark_ready(p) if grads[0] is None else None))
            self._hooks.append(p.register_post_accumulate_grad_hook(self._mark_ready))
            p._vm_grad_ready = self._note_stream
            if p.dtype == torch.bfloat16 and p.is_cuda:
                p._vm_f32_acc = self.f32_accumulator
        self.register_addresses()

    def register_addresses(self):
        """(re-)register every parameter under its CURRENT storage address: checkpointed backward nodes see detached aliases of their
        parameters and find the real ones by address (functional.PARAM_BY_PTR). Whoever re-homes the parameters afterwards
        (`FlatAdamW` moves them into its flat buffers) must call this again — the old addresses no longer resolve."""
        from . import functional as Fh
        for p in self.params:
            old = getattr(p, '_vm_reg_ptr', None)
            if old is not None and old != p.data_ptr() and Fh.PARAM_BY_PTR.get(old) is p:
                del Fh.PARAM_BY_PTR[old]
            Fh.PARAM_BY_PTR[p.data_ptr()] = p
            p._vm_reg_ptr = p.data_ptr()

    # -- layout ---------------------------------------------------------------------------------
    def _build(self, bucket_bytes: int):
        for (dtype, device, decay), lst in plan_buckets(self._seq, bucket_bytes, self.tail_bytes):
            n = sum((p.numel() + 7) // 8 * 8 for p in lst)       # 16-byte aligned slots
            buf = torch.zeros(n, dtype=dtype, device=device)
            b = _Bucket(buffer=buf, params=lst, decay=decay)
            off = 0
            for p in lst:
                p.grad = buf[off:off + p.numel()].view_as(p)
                self._bucket_of[id(p)] = len(self.buckets)
                off += (p.numel() + 7) // 8 * 8
            b.pending = len(lst)
            self.buckets.append(b)

    # -- overlap bookkeeping ---------------------------------------------------------------------
    def exposed_report(self) -> dict:
        """From the last step's logs: per bucket the position in `ready_log` at which it was launched; `exposed_bytes` = bytes of the
        buckets launched when the step's LAST gradient was already final (by that gradient's own hook or by finish()): nothing of
        backward is left to hide their all-reduce. `inversions` = buckets that were complete before an earlier-indexed bucket was
        (they waited behind it): 0 when the bucket order matches the production order."""
        n = len(self.ready_log)
        pos = {pid: i for i, pid in enumerate(self.ready_log)}
        complete_at = [max((pos.get(id(p), n) for p in b.params), default=0) for b in self.buckets]
        rows, exposed, inv = [], 0, 0
        for bi, at, by_finish in self.launch_log:
            nb = self.buckets[bi].buffer.numel() * self.buckets[bi].buffer.element_size()
            late = by_finish or at >= n
            exposed += nb if late else 0
            waited = at - 1 - complete_at[bi] if not by_finish else n - 1 - complete_at[bi]
            inv += 1 if waited > 0 else 0
            rows.append({'bucket': bi, 'bytes': nb, 'launched_at': at, 'complete_at': complete_at[bi], 'by_finish': by_finish})
        return {'n_ready': n, 'buckets': rows, 'exposed_bytes': exposed, 'inversions': inv, 'total_bytes': self.total_bytes}

    # -- replica consistency -------------------------------------------------------------------
    @torch.no_grad()
    def sync_parameters(self, extra: tuple = ()):
        """What DDPStrategy does when it wraps the module (conf/phase-vg/fit.yaml:11-15; torch DDP's `_sync_module_states`): rank 0's
        trainable parameters (and `extra` tensors: optimizer state on resume) are broadcast to every rank, packed per dtype into a few
        large messages (xGMI is per-link bound: few big transfers), then every rank checks that it holds rank 0's bytes — an
        order-independent integer checksum, all-reduced with MIN and MAX, must agree."""
        if not (dist.is_available() and dist.is_initialized()) or self.world_size <= 1:
            return
        tensors = [p.data for p in self.params] + [t for t in extra if t is not None]
        groups: dict = {}
        for t in tensors:
            groups.setdefault((t.dtype, t.device), []).append(t)
        for (dtype, device), ts in groups.items():
            for i in range(0, len(ts), 512):
                chunk = ts[i:i + 512]
                flat = torch.cat([t.reshape(-1) for t in chunk])
                dist.broadcast(flat, 0, group=self.pg)
                off = 0
                for t in chunk:
                    t.copy_(flat[off:off + t.numel()].view_as(t))
                    off += t.numel()
        self.assert_replicas_equal(tensors)

    @torch.no_grad()
    def assert_replicas_equal(self, tensors=None, what: str = 'trainable parameters'):
        """raise unless every rank holds bit-identical `tensors` (default: the trainable parameters)"""
        if not (dist.is_available() and dist.is_initialized()) or self.world_size <= 1:
            return
        tensors = [p.data for p in self.params] if tensors is None else list(tensors)
        dev = tensors[0].device
        acc = torch.zeros(2, dtype=torch.int64, device=dev)
        for t in tensors:
            b = t.detach().contiguous().view(torch.uint8) if t.element_size() == 1 else t.detach().contiguous().view(
                {2: torch.int16, 4: torch.int32, 8: torch.int64}[t.element_size()])
            v = b.reshape(-1).to(torch.int64)
            acc[0] += v.sum()
            acc[1] += (v * (torch.arange(v.numel(), device=dev, dtype=torch.int64) % 65521 + 1)).sum()      # position-sensitive
        lo, hi = acc.clone(), acc.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.pg)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.pg)
        if not torch.equal(lo, hi):
            raise RuntimeError(f'data-parallel replicas diverged: {what} differ between ranks (checksum min {lo.tolist()} max {hi.tolist()})')

    @property
    def total_bytes(self) -> int:
        return sum(b.buffer.numel() * b.buffer.element_size() for b in self.buckets)

    # -- fp32 side accumulators for bf16 slots ---------------------------------------------------
    def f32_accumulator(self, p: torch.Tensor) -> torch.Tensor:
        """A zeroed fp32 vector of p.numel() elements that kernels may accumulate into atomically during backward (the column-sum
        gradients of the bf16 norm layers). When p's bucket is launched, ONE kernel per bucket rounds every such accumulator into
        its bf16 gradient slot — `grad += acc.to(bf16)`, AccumulateGrad's rounding — and zeroes it again. Replaces, per norm
        parameter and step, a zeros() + a cast + an add launch."""
        acc = self._f32_acc.get(id(p))
        if acc is None:
            n = (p.numel() + 3) // 4 * 4
            if not self._f32_chunks or self._f32_chunks[-1][1] + n > self._f32_chunks[-1][0].numel():
                self._f32_chunks.append([torch.zeros(max(1 << 20, n), dtype=torch.float32, device=p.device), 0])
            buf, used = self._f32_chunks[-1]
            acc = buf[used:used + p.numel()]
            self._f32_chunks[-1][1] = used + n
            self._f32_acc[id(p)] = acc
            bi = self._bucket_of[id(p)]
            self._f32_entries.setdefault(bi, []).append((p.grad.data_ptr(), acc.data_ptr(), p.numel(), p))
            self._f32_tables.pop(bi, None)
        return acc

    def _fold_f32(self, bi: int):
        ent = self._f32_entries.get(bi)
        if not ent:
            return
        from . import kernels as K
        t = self._f32_tables.get(bi)
        if t is None or any(e[0] != e[3].grad.data_ptr() for e in ent):      # (grad views re-pointed: rebuild)
            ent[:] = [(e[3].grad.data_ptr(), e[1], e[2], e[3]) for e in ent]
            rows = torch.tensor([[e[0], e[1], e[2]] for e in ent], dtype=torch.int64)
            t = self._f32_tables[bi] = rows.pin_memory().to(self.buckets[bi].buffer.device, non_blocking=True)
        K.accum_f32_table(t, len(ent))

    # -- backward hooks -------------------------------------------------------------------------
    def _note_stream(self, p: torch.Tensor):
        """a fused kernel accumulated into p.grad on the current stream (possibly one of several uses of p in this step)"""
        if p.is_cuda:
            st = hip.current_stream_obj(p.device)
            b = self.buckets[self._bucket_of[id(p)]]
            if st is not b.last_stream:            # (one cached object per stream: the common case — same stream as last time — adds nothing)
                b.last_stream = st
                b.streams.add(st)
                self._step_streams.add(st)

    def _mark_ready(self, p: torch.Tensor):
        """autograd has finished p's AccumulateGrad node: every use of p has reported, the slot is final"""
        if id(p) in self._ready:
            if self.collectives and self.buckets[self._bucket_of[id(p)]].launched:
                # a second backward pass before finish(): this slot's bucket has already been all-reduced, the new contribution
                # would be added to the SUM of all ranks and never exchanged — silently wrong gradients
                raise RuntimeError('BucketedGradAllReduce: a parameter received a gradient after its bucket was reduced — with '
                                   'collectives on, run exactly one backward pass per finish() (no local gradient accumulation)')
            return
        self._ready.add(id(p))
        self.ready_log.append(id(p))
        b = self.buckets[self._bucket_of[id(p)]]
        b.pending -= 1
        if p.is_cuda:       # backward nodes run on the stream of their forward (the grounding heads use a side stream)
            st = hip.current_stream_obj(p.device)
            if st is not b.last_stream:
                b.last_stream = st
                b.streams.add(st)
                self._step_streams.add(st)
        self._launch_ready()

    def _launch(self, b: _Bucket, by_finish: bool = False):
        self.launch_log.append((self._next, len(self.ready_log), by_finish))
        if b.buffer.is_cuda:
            from . import functional as Fh
            Fh.flush_wgrad_queue()          # factor gradients still queued for a grouped launch belong to this (or an earlier) bucket
        if b.buffer.is_cuda:       # the launching stream must see every producer stream's gradient writes
            cur = torch.cuda.current_stream(b.buffer.device)
            for st in b.streams:
                if st != cur:
                    cur.wait_stream(st)
            b.streams.clear()
            b.last_stream = None
            self._step_streams.add(cur)
            self._fold_f32(self.buckets.index(b))
        if self.collectives:
            # SUM now, divide once after the collective (finish): dividing a bf16 bucket before the sum would round every
            # rank's contribution separately (the reference's DDP averages after the reduction as well)
            b.work = dist.all_reduce(b.buffer, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        b.launched = True

    def _launch_ready(self):
        # fixed launch order: bucket i+1 never overtakes bucket i, so every rank issues the same sequence
        while self._next < len(self.buckets) and self.buckets[self._next].pending <= 0:
            self._launch(self.buckets[self._next])
            self._next += 1

    # -- step boundary --------------------------------------------------------------------------
    def finish(self):
        """call after backward, on the stream that will read the gradients (clip / optimizer): reduces whatever is left
        (zero-filled slots for unused parameters), waits for the collectives, averages, and joins every stream that wrote a
        gradient in this step — also when there is no collective (one rank), where nothing else orders the side-stream
        weight-gradient kernels before the optimizer."""
        if self.buckets and self.buckets[0].buffer.is_cuda:
            from . import functional as Fh
            Fh.flush_wgrad_queue()
        while self._next < len(self.buckets):
            self._launch(self.buckets[self._next], by_finish=True)
            self._next += 1
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
                b.work = None
        if not self.collectives:
            # several backward passes before one finish() (local gradient accumulation): side accumulators filled after their bucket
            # was launched are folded now (a fold of zeroed accumulators is a no-op). With collectives ONE backward pass per finish()
            # is assumed, as for torch's DDP without no_sync(): a bucket is reduced the first time it completes.
            for st in self._step_streams:
                cur = torch.cuda.current_stream(st.device)
                if st != cur:
                    cur.wait_stream(st)
            for bi in self._f32_entries:
                self._fold_f32(bi)
        consumer = None
        for st in self._step_streams:
            if consumer is None:
                consumer = torch.cuda.current_stream(st.device)
            if st != consumer:
                consumer.wait_stream(st)
        self._step_streams.clear()
        for b in self.buckets:
            b.last_stream = None
        if self.collectives and self.world_size > 1:
            if self.defer_average:
                # the buckets hold the SUM over ranks; whoever reads them next folds 1/world into a pass it makes anyway
                # (optim.FlatAdamW: into the clip coefficient of vm_adamw; clip_grad_norm_: into its one scale per bucket)
                self.grad_scale = 1.0 / self.world_size
            else:
                inv = 1.0 / self.world_size
                for b in self.buckets:
                    b.buffer.mul_(inv)

    def abort_step(self):
        """Bring the exchange back to a clean state after a forward / backward pass that RAISED (bench.py's calibration retries after
        an out-of-memory error): wait for the device, drop the references the backward pass parked (functional._HELD_BY_TASK — the engine
        discards its final callbacks when a pass raises), forget the side stream's lag ring, zero the buckets AND the fp32 side
        accumulators (partial norm-gradient sums of the aborted pass: `zero_grad` alone leaves them to be folded into the next
        step's gradients), reset the readiness counters."""
        from . import functional as Fh
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
                b.work = None
        if self.buckets and self.buckets[0].buffer.is_cuda:
            torch.cuda.synchronize(self.buckets[0].buffer.device)
        Fh.abort_backward_state()
        for buf, _ in self._f32_chunks:
            buf.zero_()
        self.zero_grad()

    def zero_grad(self):
        off_fix = False
        for b in self.buckets:
            b.buffer.zero_()
            b.pending = len(b.params)
            b.launched = False
            b.streams.clear()
            b.last_stream = None
            for p in b.params:
                if p.grad is None or p.grad.data_ptr() < b.buffer.data_ptr() or \
                        p.grad.data_ptr() >= b.buffer.data_ptr() + b.buffer.numel() * b.buffer.element_size():
                    off_fix = True
        if off_fix:                      # somebody set grads to None: restore the views
            for b in self.buckets:
                off = 0
                for p in b.params:
                    p.grad = b.buffer[off:off + p.numel()].view_as(p)
                    off += (p.numel() + 7) // 8 * 8
        self._next = 0
        self._ready.clear()
        self._step_streams.clear()
        self.ready_log = []
        self.launch_log = []
        self.grad_scale = 1.0

    @torch.no_grad()
    def clip_grad_norm_(self, max_norm: float, eps: float = 1e-6) -> torch.Tensor:
        """torch.nn.utils.clip_grad_norm_ (L2) evaluated on the flat bucket buffers: a few reductions and one scale
        per bucket instead of one tiny kernel per parameter (the 8-byte alignment padding between slots is zero)."""
        sq = None
        for b in self.buckets:
            v = torch.linalg.vector_norm(b.buffer, 2, dtype=torch.float32)
            sq = v * v if sq is None else sq + v * v
        total = sq.sqrt() * self.grad_scale          # norm of the AVERAGED gradient (a deferred 1/world is still pending)
        coef = torch.clamp(max_norm / (total + eps), max=1.0) * self.grad_scale
        for b in self.buckets:
            b.buffer.mul_(coef.to(b.buffer.dtype))
        self.grad_scale = 1.0
        return total

    def remove(self):
        from . import functional as Fh
        for p in self.params:
            if Fh.PARAM_BY_PTR.get(p.data_ptr()) is p:
                del Fh.PARAM_BY_PTR[p.data_ptr()]
        for p in self.params:
            if getattr(p, '_vm_grad_ready', None) is not None:
                del p._vm_grad_ready
            if getattr(p, '_vm_f32_acc', None) is not None:
                del p._vm_f32_acc
        for h in self._hooks:
            h.remove()
