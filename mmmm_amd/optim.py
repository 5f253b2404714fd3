"""Optimizer side of the training step: gradient-norm clipping + AdamW fused into ONE kernel per flat bucket,
the reference's parameter groups and its learning-rate schedule.

Reference: `conf/phase-vg/fit.yaml:8-9,25-42` — `gradient_clip_val: 1` (norm), `torch.optim.AdamW(lr=5e-5, weight_decay=0.01)`,
`timm.scheduler.CosineLRScheduler(t_initial=max_steps, t_in_epochs=False, warmup_t=2000, warmup_prefix=True)` stepped with
`interval: step, frequency: 250`; parameters of type `NoWeightDecayParameter` (RMSNorm gains, cls / position tables, boi / eoi:
modeling_cogvlm.py:33, visual.py:32-35,189-190, image_encoder.py:56) are not decayed. Lightning executes this as
`clip_grad_norm_` + `AdamW.step()` — two passes over every gradient and ~1.9k parameter tensors.

Here the parameters of a `BucketedGradAllReduce` bucket are re-homed into one flat buffer with the same slot layout as the
gradients (decayed and undecayed parameters in separate buckets); the step is: a few reductions for the global norm (device
scalar), then `vm_adamw` once per bucket, which scales the gradient by the clip coefficient on the fly. No host
synchronisation, no per-tensor launches.
"""
from __future__ import annotations

import math

import torch

from . import kernels as K
from .ddp import BucketedGradAllReduce


class CosineLRSchedule:
    """`timm.scheduler.CosineLRScheduler` for one base value, as Lightning drives it from `fit.yaml:33-42`.

    timm is not in this image and is not vendored by the reference: the formula below restates timm 0.9's
    `CosineLRScheduler._get_lr` with its defaults (lr_min 0, warmup_lr_init 0, cycle_mul 1, cycle_decay 1, cycle_limit 1,
    k_decay 1) — **parity unpinned**, checked only against this closed form in tests/test_optim_cpu.py.
      t <  warmup_t : lr = warmup_lr_init + t (base − warmup_lr_init) / warmup_t
      t >= warmup_t : t' = t − warmup_t (warmup_prefix); lr = lr_min + ½ (base − lr_min)(1 + cos(π t'/t_initial)) while t' < t_initial,
                      lr_min afterwards
    `frequency`: Lightning calls the scheduler only when the optimizer-step count is a multiple of it (`frequency: 250`), so the
    rate is piecewise constant: lr(step) = f(frequency · ⌊step / frequency⌋), and the constructor's value f(0) = warmup_lr_init
    holds for the first `frequency` steps."""

    def __init__(self, base_lr: float, t_initial: int, warmup_t: int = 0, warmup_prefix: bool = False, lr_min: float = 0.0,
                 warmup_lr_init: float = 0.0, frequency: int = 1):
        self.base_lr, self.t_initial, self.warmup_t, self.warmup_prefix = base_lr, t_initial, warmup_t, warmup_prefix
        self.lr_min, self.warmup_lr_init, self.frequency = lr_min, warmup_lr_init, max(1, frequency)

    def value_at(self, t: int) -> float:
        if t < self.warmup_t:
            return self.warmup_lr_init + t * (self.base_lr - self.warmup_lr_init) / self.warmup_t
        if self.warmup_prefix:
            t = t - self.warmup_t
        if t >= self.t_initial:         # cycle_limit 1: past the single cycle
            return self.lr_min
        return self.lr_min + 0.5 * (self.base_lr - self.lr_min) * (1 + math.cos(math.pi * t / self.t_initial))

    def __call__(self, steps_done: int) -> float:
        """learning rate of the optimizer step that follows `steps_done` completed steps"""
        return self.value_at(steps_done // self.frequency * self.frequency)


class FlatAdamW:
    """`lr` may be a float or a callable steps_done -> float (`CosineLRSchedule`). The state dict has torch.optim.AdamW's layout
    (per-parameter `step` / `exp_avg` / `exp_avg_sq`, two `param_groups`: decayed and undecayed, parameters numbered in REGISTRATION
    order — `ddp.registration`, i.e. `model.parameters()`, not the production order the buckets are filled in), so a checkpoint written
    by either loads into the other."""

    def __init__(self, ddp: BucketedGradAllReduce, lr=1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 max_grad_norm: float | None = None):
        self.ddp, self.lr, self.betas, self.eps, self.weight_decay = ddp, lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self._sumsq = None               # [buckets, 1024] fp32 partial sums of squares (grad_norm_and_coef)
        ddp.defer_average = True          # finish() leaves the SUM over ranks; step() folds 1/world into the coefficient below
        self.step_count = 0
        self.last_lr = None
        self.flat_params, self.exp_avg, self.exp_avg_sq = [], [], []
        with torch.no_grad():
            for b in ddp.buckets:
                flat = torch.zeros_like(b.buffer)
                off = 0
                for p in b.params:
                    n = p.numel()
                    flat[off:off + n].copy_(p.detach().reshape(-1))
                    p.data = flat[off:off + n].view_as(p)          # the parameter now lives in the flat buffer
                    off += (n + 7) // 8 * 8
                self.flat_params.append(flat)
                self.exp_avg.append(torch.zeros_like(flat))
                self.exp_avg_sq.append(torch.zeros_like(flat))
        ddp.register_addresses()          # the parameters moved: checkpointed layers resolve their detached aliases by storage address

    def current_lr(self) -> float:
        return float(self.lr(self.step_count)) if callable(self.lr) else float(self.lr)

    @torch.no_grad()
    def grad_norm_and_coef(self):
        """global L2 norm of the bucketed gradients and the clip coefficient min(1, max_norm / (norm + 1e-6)) — device scalars"""
        # one streaming pass per bucket (vm_sumsq_partials: 1 024 per-workgroup partial sums each, a fixed element -> workgroup map), then ONE
        # reduction over all partials: ~1.7 GB of gradients at the HBM rate instead of 15 `linalg.vector_norm` calls at 1.2 TB/s (0.76 ms): 305.8 -> 304.8 ms per step, A B A B
        nb = len(self.ddp.buckets)
        if self._sumsq is None or self._sumsq.shape[0] != nb:
            self._sumsq = torch.empty(nb, 1024, dtype=torch.float32, device=self.ddp.buckets[0].buffer.device)
        for i, b in enumerate(self.ddp.buckets):
            K.sumsq_partials(b.buffer, self._sumsq[i])
        sq = self._sumsq.sum(dtype=torch.float32)
        gs = float(self.ddp.grad_scale)               # 1/world pending from a deferred finish(): the buckets hold the SUM
        total = sq.sqrt() * gs                          # norm of the averaged gradient
        coef = torch.clamp(self.max_grad_norm / (total + 1e-6), max=1.0).reshape(1) * gs if self.max_grad_norm is not None else None
        return total, coef

    @torch.no_grad()
    def step(self) -> torch.Tensor | None:
        """call after `ddp.finish()`; returns the (unclipped) gradient norm as a device scalar when clipping is on"""
        lr = self.last_lr = self.current_lr()
        self.step_count += 1
        total, coef = self.grad_norm_and_coef() if self.max_grad_norm is not None else (None, None)
        if coef is None and self.ddp.grad_scale != 1.0:
            coef = torch.full((1,), float(self.ddp.grad_scale), dtype=torch.float32, device=self.ddp.buckets[0].buffer.device)
        # (`ddp.grad_scale` stays pending: vm_adamw scales the gradients as it READS them, the buckets still hold the sum over ranks until
        # `ddp.zero_grad()` — a norm taken after step() must keep applying the factor)
        for b, p, m, v in zip(self.ddp.buckets, self.flat_params, self.exp_avg, self.exp_avg_sq):
            K.adamw_(p, b.buffer, m, v, lr=lr, betas=self.betas, eps=self.eps, weight_decay=self.weight_decay if b.decay else 0.0,
                     step=self.step_count, clip_coef=coef)
        # the kernel writes through raw pointers: tensor version counters do not move, so caches keyed on them (the resident
        # LoRA transposes, models/lora.py) are told explicitly that every parameter changed
        from .models.lora import ParamGeneration
        ParamGeneration.bump()
        return total

    # -- checkpoint / resume ---------------------------------------------------------------------
    def _slots(self):
        """(parameter, bucket index, offset) in REGISTRATION order (`ddp.registration`: what torch.optim numbers by), whatever order the
        gradient buckets were filled in"""
        where = {}
        for bi, b in enumerate(self.ddp.buckets):
            off = 0
            for p in b.params:
                where[id(p)] = (bi, off)
                off += (p.numel() + 7) // 8 * 8
        return [(p, *where[id(p)]) for p in self.ddp.registration]

    def state_dict(self) -> dict:
        slots = self._slots()
        state = {}
        for i, (p, bi, off) in enumerate(slots):
            n = p.numel()
            state[i] = {'step': torch.tensor(float(self.step_count)),
                        'exp_avg': self.exp_avg[bi][off:off + n].view_as(p).clone(),
                        'exp_avg_sq': self.exp_avg_sq[bi][off:off + n].view_as(p).clone()}
        groups = []
        for decay in (True, False):
            ids = [i for i, (p, bi, _) in enumerate(slots) if self.ddp.buckets[bi].decay == decay]
            if ids:
                groups.append({'lr': self.current_lr(), 'betas': tuple(self.betas), 'eps': self.eps,
                               'weight_decay': self.weight_decay if decay else 0.0, 'amsgrad': False, 'params': ids})
        return {'state': state if self.step_count else {}, 'param_groups': groups}

    @torch.no_grad()
    def load_state_dict(self, sd: dict):
        slots = self._slots()
        n_saved = sum(len(g['params']) for g in sd['param_groups'])
        if n_saved != len(slots):
            raise ValueError(f'optimizer state holds {n_saved} parameters, this optimizer {len(slots)}')
        # torch numbers parameters group by group; map saved index -> position in ddp.params
        order = [i for g in sd['param_groups'] for i in g['params']]
        mine = [i for decay in (True, False) for i, (p, bi, _) in enumerate(slots) if self.ddp.buckets[bi].decay == decay]
        if len(sd['param_groups']) == 1:        # a single-group checkpoint (torch.optim.AdamW(params)) lists them in the given order
            mine = list(range(len(slots)))
        steps = set()
        for saved_i, my_i in zip(order, mine):
            st = sd['state'].get(saved_i)
            p, bi, off = slots[my_i]
            n = p.numel()
            if st is None:
                self.exp_avg[bi][off:off + n].zero_()
                self.exp_avg_sq[bi][off:off + n].zero_()
                continue
            if tuple(st['exp_avg'].shape) != tuple(p.shape):
                raise ValueError(f'optimizer state {saved_i}: shape {tuple(st["exp_avg"].shape)} vs parameter {tuple(p.shape)}')
            self.exp_avg[bi][off:off + n].copy_(st['exp_avg'].reshape(-1))
            self.exp_avg_sq[bi][off:off + n].copy_(st['exp_avg_sq'].reshape(-1))
            steps.add(int(st['step']))
        if len(steps) > 1:
            raise ValueError(f'per-parameter step counts differ ({sorted(steps)}): one bias correction per bucket cannot represent that')
        self.step_count = steps.pop() if steps else 0
