"""Optimizer side of the training step: gradient-norm clipping + AdamW fused into ONE kernel per flat bucket.

Reference: `conf/phase-vg/fit.yaml` (gradient_clip_val 1.0, AdamW) executed by Lightning as
`torch.nn.utils.clip_grad_norm_` + `torch.optim.AdamW.step()` — two passes over every gradient and ~1.9k parameter tensors.
Here the parameters of a `BucketedGradAllReduce` bucket are re-homed into one flat buffer with the same slot layout as the
gradients; the step is: a few reductions for the global norm (device scalar), then `vm_adamw` once per bucket, which scales
the gradient by the clip coefficient on the fly. No host synchronisation, no per-tensor launches.
"""
from __future__ import annotations

import torch

from . import kernels as K
from .ddp import BucketedGradAllReduce


class FlatAdamW:
    def __init__(self, ddp: BucketedGradAllReduce, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 max_grad_norm: float | None = None):
        self.ddp, self.lr, self.betas, self.eps, self.weight_decay = ddp, lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.step_count = 0
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

    @torch.no_grad()
    def grad_norm_and_coef(self):
        """global L2 norm of the bucketed gradients and the clip coefficient min(1, max_norm / (norm + 1e-6)) — device scalars"""
        sq = None
        for b in self.ddp.buckets:
            v = torch.linalg.vector_norm(b.buffer, 2, dtype=torch.float32)
            sq = v * v if sq is None else sq + v * v
        total = sq.sqrt()
        coef = torch.clamp(self.max_grad_norm / (total + 1e-6), max=1.0).reshape(1) if self.max_grad_norm is not None else None
        return total, coef

    @torch.no_grad()
    def step(self) -> torch.Tensor | None:
        """call after `ddp.finish()`; returns the (unclipped) gradient norm as a device scalar when clipping is on"""
        self.step_count += 1
        total, coef = self.grad_norm_and_coef() if self.max_grad_norm is not None else (None, None)
        for b, p, m, v in zip(self.ddp.buckets, self.flat_params, self.exp_avg, self.exp_avg_sq):
            K.adamw_(p, b.buffer, m, v, lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=self.weight_decay,
                     step=self.step_count, clip_coef=coef)
        return total

    def state_dict(self) -> dict:
        return {'step': self.step_count, 'exp_avg': self.exp_avg, 'exp_avg_sq': self.exp_avg_sq}
