"""Linear layers of the VividMed model on the HIP GEMM, with the LoRA adapters fused in.

Mirrors what the reference gets from `nn.Linear` + `peft.lora.Linear` (scripts/cli.py:82-85,
conf/lora.yaml: r=64, lora_alpha=8, lora_dropout=0.05, use_rslora) — state-dict keys follow PEFT:
`<name>.weight` (base), `<name>.lora_A.default.weight` [r,in], `<name>.lora_B.default.weight` [out,r].
"""
from __future__ import annotations

from dataclasses import dataclass
import math

import torch
from torch import nn

from .. import functional as Fh
from .. import kernels as K


@dataclass
class LoraConfig:
    """the fields of peft.LoraConfig that conf/lora.yaml sets"""
    r: int = 64
    lora_alpha: float = 8.0
    lora_dropout: float = 0.05
    use_rslora: bool = True

    @property
    def scale(self) -> float:
        return self.lora_alpha / math.sqrt(self.r) if self.use_rslora else self.lora_alpha / self.r


class _Holder(nn.Module):
    def __init__(self, w: torch.Tensor):
        super().__init__()
        self.weight = nn.Parameter(w)


class StepState:
    """dropout seeds must be identical in forward, checkpoint recompute and backward: they are a pure
    function of (global step, site id). The LightningModule bumps `step` once per training_step."""
    step: int = 0
    seed: int = 0x5EED
    _sites: int = 0

    @classmethod
    def new_site(cls) -> int:
        cls._sites += 1
        return cls._sites

    @classmethod
    def seed_for(cls, site: int) -> int:
        return (cls.seed * 1000003 + cls.step * 7919 + site * 104729) & 0x7FFFFFFFFFFFFFFF


class Linear(nn.Module):
    """y = x W^T (+ b) (+ s·B A drop(x)); frozen weights keep a transposed copy for the dgrad GEMM."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, dtype=None, device=None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features, dtype=dtype, device=device))
        self.bias = nn.Parameter(torch.zeros(out_features, dtype=dtype, device=device)) if bias else None
        self.lora_A: nn.ModuleDict | None = None
        self.lora_B: nn.ModuleDict | None = None
        self.lora_cfg: LoraConfig | None = None
        self._wt: torch.Tensor | None = None
        self._site = StepState.new_site()
        nn.init.normal_(self.weight, std=0.02)

    # -- LoRA
    def add_lora(self, cfg: LoraConfig):
        w = self.weight
        a = torch.empty(cfg.r, self.in_features, dtype=w.dtype, device=w.device)
        nn.init.kaiming_uniform_(a, a=math.sqrt(5))          # peft's default init: A kaiming, B zero
        b = torch.zeros(self.out_features, cfg.r, dtype=w.dtype, device=w.device)
        self.lora_A = nn.ModuleDict({'default': _Holder(a)})
        self.lora_B = nn.ModuleDict({'default': _Holder(b)})
        self.lora_cfg = cfg
        self.weight.requires_grad_(False)
        if self.bias is not None:
            self.bias.requires_grad_(False)

    @property
    def A(self):
        return self.lora_A['default'].weight if self.lora_A is not None else None

    @property
    def B(self):
        return self.lora_B['default'].weight if self.lora_B is not None else None

    def wt(self) -> torch.Tensor | None:
        """transposed copy [in, out] of a FROZEN weight (kept resident: 288 GB HBM buys a plain NT dgrad)"""
        if self.weight.requires_grad:
            return None
        if self._wt is None or self._wt.device != self.weight.device or self._wt.dtype != self.weight.dtype:
            self._wt = K.transpose(self.weight.detach())
        return self._wt

    def meta(self, gated: bool = False) -> Fh.LinearMeta:
        m = Fh.LinearMeta(gated=gated)
        if self.lora_cfg is not None:
            m.lora_scale = self.lora_cfg.scale
            if self.training and self.lora_cfg.lora_dropout > 0:
                m.drop_p = self.lora_cfg.lora_dropout
                m.drop_seed = StepState.seed_for(self._site)
        return m

    def forward(self, x: torch.Tensor, residual: torch.Tensor | None = None) -> torch.Tensor:
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        r2 = residual.reshape(-1, self.out_features) if residual is not None else None
        need_dx = torch.is_grad_enabled() and x2.requires_grad
        y = Fh.linear(x2, self.weight, meta=self.meta(), Wt0=self.wt() if need_dx else None, b0=self.bias, A0=self.A, B0=self.B,
                      residual=r2)
        return y.view(*shape[:-1], self.out_features)


def gated_linear(x: torch.Tensor, vision: Linear, language: Linear, counts: torch.Tensor, residual: torch.Tensor | None = None):
    """token-type gated pair of linears on the expert-sorted row layout: rows [0,counts[0]) -> `vision`,
    rows [counts[0],counts[1]) -> `language` (reference modeling_cogvlm.py:243-245, 277-279, 95-97)."""
    need_dx = torch.is_grad_enabled() and x.requires_grad
    m = vision.meta(gated=True)
    if language.lora_cfg is not None and vision.lora_cfg is None:
        raise NotImplementedError('LoRA on the language expert only')
    lora_l = language.lora_cfg is not None
    return Fh.linear(
        x, vision.weight, meta=m, Wt0=vision.wt() if need_dx else None, b0=vision.bias, A0=vision.A, B0=vision.B,
        W1=language.weight, Wt1=language.wt() if need_dx else None, b1=language.bias,
        A1=language.A if lora_l else _zero_like(vision.A), B1=language.B if lora_l else _zero_like(vision.B),
        residual=residual, counts=counts,
    )


def _zero_like(t):
    return None if t is None else torch.zeros_like(t)
