"""`NoWeightDecayParameter` — the marker the reference imports from `luolib.models.param` (an `nn.Parameter` subclass with no
behaviour of its own). The reference puts it on the RMSNorm gains (modeling_cogvlm.py:33), the ViT's
cls / cls-position / position tables (visual.py:32-35), `boi` / `eoi` (visual.py:189-190) and SAM's position table
(segvol/modeling/image_encoder.py:56); its optimizer builder (luolib, absent from /root/reference) gives parameters of this type
weight_decay = 0. `mmmm_amd.optim.FlatAdamW` and `mmmm_amd.ddp.BucketedGradAllReduce` honour the marker: marked and unmarked
parameters live in separate flat buckets, so one `vm_adamw` launch has one decay value."""
from __future__ import annotations

import torch
from torch import nn


class NoWeightDecayParameter(nn.Parameter):
    """nn.Parameter that optimizers must not decay"""

    def __new__(cls, data=None, requires_grad: bool = True):
        if data is None:
            data = torch.empty(0)
        return torch.Tensor._make_subclass(cls, data, requires_grad)

    def __deepcopy__(self, memo):
        if id(self) in memo:
            return memo[id(self)]
        out = type(self)(self.data.clone(memory_format=torch.preserve_format), self.requires_grad)
        memo[id(self)] = out
        return out

    def __reduce_ex__(self, proto):
        return _rebuild_no_decay, (self.data, self.requires_grad)


def _rebuild_no_decay(data, requires_grad):
    return NoWeightDecayParameter(data, requires_grad)


def no_weight_decay(p: torch.Tensor) -> bool:
    return isinstance(p, NoWeightDecayParameter)
