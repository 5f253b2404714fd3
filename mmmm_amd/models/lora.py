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


class ParamGeneration:
    """Bumped by optimizers that update parameters through raw device pointers (optim.FlatAdamW: the fused kernel never moves
    a tensor's `_version`). Part of the freshness key of every cache derived from parameter values (`Linear.lora_t`), so a
    cached transpose can never outlive an optimizer step."""
    value: int = 0

    @classmethod
    def bump(cls):
        cls.value += 1


class ActivationBudget:
    """How many transformer layers may keep their activations instead of being recomputed in backward.

    The reference turns gradient checkpointing on for every layer (mmmm.py:232-233 via on_fit_start) because an 80 GB
    device leaves no choice; the recompute is a quarter of the step's matrix work. With 288 GB of HBM3E the saved
    activations of the whole VividMed step fit several times over, so checkpointing becomes a per-layer decision
    against a byte budget: `claim()` hands each transformer the number of layers it may run un-checkpointed. Results
    are identical either way (dropout masks are a pure function of (seed, step, site, element))."""
    limit: int | None = None      # bytes; None = reference behaviour (checkpoint every layer)
    remaining: int = 0
    safety: float = 1.25
    last_plan: list = []          # [(layers, kept)] of the current step, for logging

    @classmethod
    def reset(cls):
        cls.remaining = cls.limit or 0
        cls.last_plan = []

    @classmethod
    def claim(cls, n_layers: int, bytes_per_layer: int) -> int:
        if cls.limit is None or bytes_per_layer <= 0:
            cls.last_plan.append((n_layers, 0))
            return 0
        need = int(bytes_per_layer * cls.safety)
        n = int(min(n_layers, cls.remaining // need))
        cls.remaining -= n * need
        cls.last_plan.append((n_layers, n))
        return n


class Fp8Weights:
    """e4m3 copies of one FROZEN weight for the fp8 mode (BASELINE configs[4]): `w8` [out, in] with one scale per output channel for the
    forward, `wt8` [in, out] with one scale per input channel for the input gradient. 2 bytes per parameter in total — what the bf16
    transposed copy alone costs, which this mode no longer keeps."""

    def __init__(self, weight: torch.Tensor):
        w = weight.detach()
        self.w8, self.sw, self.inv_sw = K.quant_rows_fp8(w)
        wt = K.transpose(w)
        self.wt8, self.swt, self.inv_swt = K.quant_rows_fp8(wt)
        del wt


FLAT_LINEAR = True      # a 2-D input goes to the GEMM as it is and the result comes back as it is: without this every tower linear sat between a reshape and a
                        # view of the same shape (1 100 ViewBackward0 nodes per step, ~10 ms of host enqueue: tools/graph_nodes.py); A/B: bench.py --set models.lora.FLAT_LINEAR=False


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
        self._lora_t: tuple | None = None      # (At [in, r], Bt [r, out], version key) — see LoraTransposes
        self.f8: Fp8Weights | None = None      # set by enable_fp8(): e4m3 copies of the frozen weight
        self.f32_split = 0                     # fp32 layers: arithmetic of this layer's GEMMs (kernels.gemm `f32_split`; 0 = default)
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
        self._drop_factor_cache()
        if not getattr(self, '_factor_hook', False):
            self.register_load_state_dict_post_hook(lambda module, incompatible: module._drop_factor_cache())
            self._factor_hook = True
        self.lora_cfg = cfg
        self.weight.requires_grad_(False)
        if self.bias is not None:
            self.bias.requires_grad_(False)

    # (plain attribute reads: `self.lora_A['default'].weight` is three nn.Module.__getattr__ calls + a ModuleDict lookup, and the step reads
    # the factors ~3 700 times — 12 ms of host time per step under cProfile. The Parameter objects themselves never change: optimizers and
    # checkpoint loads write into them.)
    @property
    def A(self):
        a = self.__dict__.get('_A')
        if a is None and self.lora_A is not None:
            a = self.lora_A['default'].weight
            self.__dict__['_A'] = a
        return a

    @property
    def B(self):
        b = self.__dict__.get('_B')
        if b is None and self.lora_B is not None:
            b = self.lora_B['default'].weight
            self.__dict__['_B'] = b
        return b

    def _drop_factor_cache(self):
        self.__dict__.pop('_A', None), self.__dict__.pop('_B', None)

    # Everything that can REPLACE the Parameter objects (rather than write into them) goes through one of these two: `_apply` (to / to_empty /
    # cuda / float / ... incl. the swap-module-params-on-conversion switches) and `load_state_dict(assign=True)` (-> _load_from_state_dict of the
    # holder, whose parent gets the post hook below). The cached objects are dropped there, so forward and the factor-gradient queue can never
    # keep using tensors the module no longer owns.
    def _apply(self, fn, *args, **kwargs):
        self._drop_factor_cache()
        out = super()._apply(fn, *args, **kwargs)
        self._drop_factor_cache()
        return out

    def fp8_eligible(self) -> bool:
        w = self.weight
        return (not w.requires_grad and w.dtype == torch.bfloat16 and w.is_cuda and self.in_features % 128 == 0 and self.out_features % 128 == 0)

    def wt(self) -> torch.Tensor | None:
        """transposed copy [in, out] of a FROZEN weight (kept resident: 288 GB HBM buys a plain NT dgrad)"""
        if self.weight.requires_grad or self.f8 is not None:
            return None
        if Fh.NN_DGRAD and self.weight.dtype == torch.bfloat16 and self.out_features % 64 == 0 and self.in_features % 8 == 0:
            return None          # the dgrad GEMM reads W itself (functional._Linear.backward, `b_nn`): no resident transpose
        if self._wt is None or self._wt.device != self.weight.device or self._wt.dtype != self.weight.dtype:
            self._wt = K.transpose(self.weight.detach())
        return self._wt

    def lora_t(self):
        """(At, Bt) resident transposed LoRA factors if `LoraTransposes.refresh()` ran after the last in-place
        update of A/B (tensor version counters), else (None, None) and the backward transposes per use."""
        c = self._lora_t
        if c is not None and c[2] == (self.A._version, self.B._version, self.A.data_ptr(), self.B.data_ptr(), ParamGeneration.value):
            return c[0], c[1]
        return None, None

    def meta(self, gated: bool = False) -> Fh.LinearMeta:
        m = Fh.LinearMeta(gated=gated)
        m.f8_0 = self.f8
        m.f32_split = self.f32_split
        if self.lora_cfg is not None:
            m.lora_scale = self.lora_cfg.scale
            if torch.is_grad_enabled():
                m.At0, m.Bt0 = self.lora_t()
            if self.training and self.lora_cfg.lora_dropout > 0:
                m.drop_p = self.lora_cfg.lora_dropout
                m.drop_seed = StepState.seed_for(self._site)
        return m

    def forward(self, x: torch.Tensor, residual: torch.Tensor | None = None, fork: bool = False):
        """`fork`: -> (y, x passed through): hand the second output to the block's residual add and the residual's gradient is
        summed into dx by the dgrad GEMM's epilogue (functional.LinearMeta.fork)."""
        shape = x.shape
        flat = x.dim() == 2 and FLAT_LINEAR          # the towers hand over [rows, hidden]: no reshape / view nodes around the linear
        x2 = x if flat else x.reshape(-1, shape[-1])
        r2 = (residual if residual.dim() == 2 else residual.reshape(-1, self.out_features)) if residual is not None else None
        need_dx = torch.is_grad_enabled() and x2.requires_grad
        meta = self.meta()
        meta.fork = fork and Fh.FORK_LINEAR
        y = Fh.linear(x2, self.weight, meta=meta, Wt0=self.wt() if need_dx else None, b0=self.bias, A0=self.A, B0=self.B,
                      residual=r2)
        if fork and not meta.fork:
            return (y if flat else y.view(*shape[:-1], self.out_features)), x
        if fork:
            y, xp = y
            return (y, xp) if flat else (y.view(*shape[:-1], self.out_features), xp.view(shape))
        return y if flat else y.view(*shape[:-1], self.out_features)


@torch.no_grad()
def linear_decode(x: torch.Tensor, lin: Linear, residual: torch.Tensor | None = None) -> torch.Tensor:
    """one Linear (LoRA included, dropout off) on the few rows of a decode step: W is streamed once by the skinny-M kernel
    (`vm_gemv_bf16`) when the shape allows, else by the tiled GEMM"""
    t = None
    if lin.lora_cfg is not None:
        # t = x A^T: the rank-64 factor is itself a skinny linear (one launch; the split-K down-projection kernel of the
        # training path is two launches sized for thousands of rows)
        t = K.gemv(x, lin.A) if K.gemv_supported(x, lin.A) else Fh._lora_project(x, lin.A, None, False, None)
    scale = lin.lora_cfg.scale if lin.lora_cfg is not None else 1.0
    if K.gemv_supported(x, lin.weight, t):
        return K.gemv(x, lin.weight, a2=t, b2=lin.B, alpha2=scale, bias=lin.bias, residual=residual)
    return K.gemm(x, lin.weight, a2=t, b2=lin.B, alpha2=scale, bias=lin.bias, residual=residual)


def gated_linear(x: torch.Tensor, vision: Linear, language: Linear, counts: torch.Tensor, residual: torch.Tensor | None = None):
    """token-type gated pair of linears on the expert-sorted row layout: rows [0,counts[0]) -> `vision`,
    rows [counts[0],counts[1]) -> `language` (reference modeling_cogvlm.py:243-245, 277-279, 95-97)."""
    need_dx = torch.is_grad_enabled() and x.requires_grad
    m = vision.meta(gated=True)
    if language.lora_cfg is not None and vision.lora_cfg is None:
        raise NotImplementedError('LoRA on the language expert only')
    lora_l = language.lora_cfg is not None
    if lora_l and torch.is_grad_enabled():
        m.At1, m.Bt1 = language.lora_t()
    m.f8_1 = language.f8
    if (m.f8_0 is None) != (m.f8_1 is None):
        m.f8_0 = m.f8_1 = None
    return Fh.linear(
        x, vision.weight, meta=m, Wt0=vision.wt() if need_dx else None, b0=vision.bias, A0=vision.A, B0=vision.B,
        W1=language.weight, Wt1=language.wt() if need_dx else None, b1=language.bias,
        A1=language.A if lora_l else _zero_like(vision.A), B1=language.B if lora_l else _zero_like(vision.B),
        residual=residual, counts=counts,
    )


def _zero_like(t):
    return None if t is None else torch.zeros_like(t)


class LoraTransposes:
    """Resident K-contiguous copies (A^T [in, r], B^T [r, out]) of every LoRA factor below `root`, refreshed by ONE
    launch (`vm_transpose_batched`) after each optimizer step. The backward of a LoRA linear needs both transposes
    (dx += s·(dy B) A; functional._Linear.backward); doing them per use costs two tiny launches per linear per step
    (~1.3k launches on the full VividMed model). Staleness is detected through the parameters' version counters plus
    `ParamGeneration` (raw-pointer optimizers), so a forgotten `refresh()` only falls back to the per-use path, never to stale
    factors."""

    def __init__(self, root: nn.Module):
        self.linears = [m for m in root.modules() if isinstance(m, Linear) and m.lora_cfg is not None]
        self.desc = None
        self.n = 0
        self.tiles = 1
        self.dtype = None
        self._ptrs = None

    def _build(self):
        rows = []
        bufs = []
        self.dtype = self.linears[0].A.dtype
        for m in self.linears:
            A, B = m.A, m.B
            assert A.dtype == self.dtype and B.dtype == self.dtype, 'LoRA factors must share one dtype'
            At = torch.empty(A.shape[1], A.shape[0], dtype=A.dtype, device=A.device)
            Bt = torch.empty(B.shape[1], B.shape[0], dtype=B.dtype, device=B.device)
            bufs.append((At, Bt))
            for src, dst in ((A, At), (B, Bt)):
                rows.append([src.data_ptr(), dst.data_ptr(), src.shape[0], src.shape[1], src.stride(0), dst.stride(0)])
                self.tiles = max(self.tiles, min(64, -(-src.shape[0] // 64) * -(-src.shape[1] // 64)))
        self.bufs = bufs
        self.n = len(rows)
        self.desc = torch.tensor(rows, dtype=torch.int64).to(self.linears[0].A.device)
        self._ptrs = tuple(m.A.data_ptr() for m in self.linears) + tuple(m.B.data_ptr() for m in self.linears)

    @torch.no_grad()
    def refresh(self):
        if not self.linears:
            return
        ptrs = tuple(m.A.data_ptr() for m in self.linears) + tuple(m.B.data_ptr() for m in self.linears)
        if self.desc is None or ptrs != self._ptrs:
            self._build()
        for i in range(0, self.n, 65534):
            n = min(65534, self.n - i)
            K.transpose_batched(self.desc[i:i + n].contiguous() if i else self.desc, n, self.tiles, self.dtype)
        for m, (At, Bt) in zip(self.linears, self.bufs):
            m._lora_t = (At, Bt, (m.A._version, m.B._version, m.A.data_ptr(), m.B.data_ptr(), ParamGeneration.value))


@torch.no_grad()
def enable_fp8(root: nn.Module) -> int:
    """Switch every eligible frozen bf16 Linear below `root` to the fp8 mode (e4m3 main product, forward and input gradient; see
    functional._Linear): quantise W and W^T once and drop the resident bf16 transpose. Returns the number of linears converted."""
    n = 0
    for m in root.modules():
        if isinstance(m, Linear) and m.f8 is None and m.fp8_eligible():
            m.f8 = Fp8Weights(m.weight)
            m._wt = None
            n += 1
    if n and torch.cuda.is_available():
        torch.cuda.empty_cache()        # the bf16 transposes the quantiser went through were temporaries: hand their blocks back
    return n
