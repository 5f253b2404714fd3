"""TEST INFRASTRUCTURE — runs only in the development container.

Imports the *unmodified* reference model files from /root/reference (read-only) so that golden
vectors can be generated from the reference itself (oracle/make_golden.py). Nothing here is shipped,
nothing here travels to the GPU box, and no product code imports it.

The reference depends on packages that are absent from this image and whose source is not under
/root/reference: luolib (empty submodule third-party/LuoLib, unpinned), monai, xformers
0.0.27, peft, lightning, cytoolz, jsonargparse, torchvision. They are replaced by the minimal shims
below. Each shim restates *published* semantics; where the true semantics are unknown
(`luolib.models.spadop.resample` when the shape changes) parity is "unpinned" and fixtures avoid or
isolate that case (the choice is recorded in the fixture metadata).
"""
from __future__ import annotations

import importlib
import sys
import types
from dataclasses import dataclass
from pathlib import Path

import einops
import torch
from torch import nn
import torch.nn.functional as F

REF = Path('/root/reference')


def _mod(name: str, **attrs) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, child = name.rpartition('.')
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


# ----------------------------------------------------------------------------- luolib
class NoWeightDecayParameter(nn.Parameter):
    pass


def forward_gc(module, enabled, fn, *args, **kwargs):
    if enabled and module.training and fn is not None:
        return fn(module.__call__, *args, **kwargs)
    return module(*args, **kwargs)


def resample(x: torch.Tensor, shape, scale: bool = False):
    """luolib.models.spadop.resample — UNPINNED. Identity when the spatial shape already matches;
    otherwise linear interpolation (align_corners=False), with `scale=True` rescaling values by the
    ratio of element counts so that a convolution kernel keeps its response to constant input."""
    shape = tuple(int(s) for s in shape)
    nd = len(shape)
    if tuple(x.shape[-nd:]) == shape:
        return x
    mode = {1: 'linear', 2: 'bilinear', 3: 'trilinear'}[nd]
    lead = x.shape[:-nd]
    y = F.interpolate(x.reshape(1, -1, *x.shape[-nd:]).float(), size=shape, mode=mode, align_corners=False)
    y = y.reshape(*lead, *shape).to(x.dtype)
    if scale:
        src = 1
        for s in x.shape[-nd:]:
            src *= s
        dst = 1
        for s in shape:
            dst *= s
        y = y * (src / dst)
    return y


def flatten(x):  # 'n c ... -> n (...) c'
    return x.flatten(2).transpose(1, 2)


def spatialize(x, shape):  # 'n (d h w) c -> n c d h w'
    return x.transpose(1, 2).reshape(x.shape[0], x.shape[2], *shape)


def channel_last(x):
    return x.movedim(1, -1)


def channel_first(x):
    return x.movedim(-1, 1)


def pairwise_forward(fn, a, b, **kwargs):
    n, m = a.shape[0], b.shape[0]
    ai = a[:, None].expand(n, m, *a.shape[1:]).reshape(n * m, *a.shape[1:])
    bi = b[None].expand(n, m, *b.shape[1:]).reshape(n * m, *b.shape[1:])
    return fn(ai, bi, **kwargs).reshape(n, m)


def zero_loss(*tensors):
    out = None
    for t in tensors:
        if t is None:
            continue
        z = 0 * t.sum()
        out = z if out is None else out + z
    if out is None:
        out = torch.zeros(())
    return out


def sigmoid_focal_loss(input, target, gamma: float = 2.0, alpha: float | None = None):
    """torchvision.ops.sigmoid_focal_loss formula, reduction='none', alpha optional."""
    target = target.to(input.dtype)
    p = torch.sigmoid(input)
    ce = F.binary_cross_entropy_with_logits(input, target, reduction='none')
    p_t = p * target + (1 - p) * (1 - target)
    loss = ce * (1 - p_t) ** gamma
    if alpha is not None:
        loss = (alpha * target + (1 - alpha) * (1 - target)) * loss
    return loss


def bce_with_binary_label(input, target):
    if target is None:
        target = torch.zeros_like(input)
    return F.binary_cross_entropy_with_logits(input, target.to(input.dtype), reduction='none')


class _LightningModule(nn.Module):
    trainer = None

    def on_fit_start(self):
        pass

    def log(self, *a, **k):
        self.__dict__.setdefault('_logged', {})[a[0]] = a[1].detach() if torch.is_tensor(a[1]) else a[1]

    def log_dict(self, d, *a, **k):
        lg = self.__dict__.setdefault('_logged', {})
        for key, v in d.items():
            lg[key] = v.detach() if torch.is_tensor(v) else v

    @property
    def device(self):
        return next(self.parameters()).device


class _PeftMixin:
    peft_model = None

    def set_peft_model(self, m):
        self.peft_model = m


# ----------------------------------------------------------------------------- monai
class _SABlock(nn.Module):
    def __init__(self, hidden_size, num_heads, dropout_rate=0.0, qkv_bias=False, save_attn=False):
        super().__init__()
        self.num_heads = num_heads
        self.out_proj = nn.Linear(hidden_size, hidden_size)
        self.qkv = nn.Linear(hidden_size, hidden_size * 3, bias=qkv_bias)
        self.drop_output = nn.Dropout(dropout_rate)
        self.drop_weights = nn.Dropout(dropout_rate)
        self.head_dim = hidden_size // num_heads
        self.scale = self.head_dim ** -0.5


class _MLPBlock(nn.Module):
    def __init__(self, hidden_size, mlp_dim, dropout_rate=0.0):
        super().__init__()
        self.linear1 = nn.Linear(hidden_size, mlp_dim)
        self.linear2 = nn.Linear(mlp_dim, hidden_size)
        self.fn = nn.GELU()
        self.drop1 = nn.Dropout(dropout_rate)
        self.drop2 = nn.Dropout(dropout_rate)

    def forward(self, x):
        return self.drop2(self.linear2(self.drop1(self.fn(self.linear1(x)))))


class _TransformerBlock(nn.Module):
    def __init__(self, hidden_size, mlp_dim, num_heads, dropout_rate=0.0, qkv_bias=False, save_attn=False):
        super().__init__()
        self.mlp = _MLPBlock(hidden_size, mlp_dim, dropout_rate)
        self.norm1 = nn.LayerNorm(hidden_size)
        self.attn = _SABlock(hidden_size, num_heads, dropout_rate, qkv_bias, save_attn)
        self.norm2 = nn.LayerNorm(hidden_size)


class CenterSizeMode:
    pass


def convert_box_mode(boxes, src_mode=None, dst_mode=None):
    """CenterSize (c0,c1,c2,s0,s1,s2) -> CornerCorner (min..., max...) as monai does for src=CenterSizeMode"""
    assert src_mode is CenterSizeMode and dst_mode is None
    c, s = boxes[..., :3], boxes[..., 3:]
    return torch.cat([c - s / 2, c + s / 2], dim=-1)


def box_pair_giou(boxes1, boxes2):
    """monai.data.box_utils.box_pair_giou (3-D, CornerCorner), fp32, eps in the denominators."""
    eps = torch.finfo(torch.float32).eps
    b1, b2 = boxes1.float(), boxes2.float()
    a1 = (b1[..., 3:] - b1[..., :3]).prod(-1)
    a2 = (b2[..., 3:] - b2[..., :3]).prod(-1)
    lt = torch.max(b1[..., :3], b2[..., :3])
    rb = torch.min(b1[..., 3:], b2[..., 3:])
    inter = (rb - lt).clamp(min=0).prod(-1)
    union = a1 + a2 - inter
    iou = inter / (union + eps)
    lt2 = torch.min(b1[..., :3], b2[..., :3])
    rb2 = torch.max(b1[..., 3:], b2[..., 3:])
    enclosure = (rb2 - lt2).clamp(min=0).prod(-1)
    giou = iou - (enclosure - union) / (enclosure + eps)
    return giou.to(boxes1.dtype)


# ----------------------------------------------------------------------------- xformers
class _BlockDiag:
    causal = False

    def __init__(self, q_lens, k_lens=None):
        self.q_lens = list(q_lens)
        self.k_lens = list(k_lens) if k_lens is not None else list(q_lens)

    @classmethod
    def from_tensor_list(cls, tensors):
        lens = [t.shape[1] for t in tensors]
        return cls(lens), torch.cat(tensors, dim=1)

    @classmethod
    def from_tensor_lists_qkv(cls, ql, kl, vl):
        bias = cls([t.shape[1] for t in ql], [t.shape[1] for t in kl])
        return bias, torch.cat(ql, 1), torch.cat(kl, 1), torch.cat(vl, 1)

    def split(self, x):
        return x.split(self.q_lens, dim=1)

    def materialize(self, device):
        Q, Kk = sum(self.q_lens), sum(self.k_lens)
        m = torch.zeros(Q, Kk, dtype=torch.bool, device=device)
        qs = ks = 0
        for ql, kl in zip(self.q_lens, self.k_lens):
            blk = torch.ones(ql, kl, dtype=torch.bool, device=device)
            if self.causal:
                blk = blk.tril()
            m[qs:qs + ql, ks:ks + kl] = blk
            qs += ql
            ks += kl
        return m


class BlockDiagonalMask(_BlockDiag):
    pass


class BlockDiagonalCausalMask(_BlockDiag):
    causal = True


def memory_efficient_attention(q, k, v, attn_bias=None, p: float = 0.0, scale: float | None = None):
    """[B, M, H, K] layout, exact softmax attention (the reference's FMHA differs only in summation order)."""
    assert p == 0.0
    qh, kh, vh = (t.transpose(1, 2) for t in (q, k, v))
    mask = None
    if attn_bias is not None:
        mask = attn_bias.materialize(q.device)
    out = F.scaled_dot_product_attention(qh, kh, vh, attn_mask=mask, scale=scale)
    return out.transpose(1, 2)


# ----------------------------------------------------------------------------- install
_installed = False


def install():
    """Register shims and package shells; afterwards `import mmmm.models.mmmm` etc. load the reference files."""
    global _installed
    if _installed:
        return
    import transformers  # noqa: F401  (must be imported before the fake torchvision appears)
    from transformers import PreTrainedModel  # noqa: F401

    _mod('luolib')
    _mod('luolib.types', tuple2_t=tuple, tuple3_t=tuple, param3_t=tuple, PathLike=object)
    _mod('luolib.models')
    _mod('luolib.models.param', NoWeightDecayParameter=NoWeightDecayParameter)
    _mod('luolib.models.utils', forward_gc=forward_gc)
    _mod('luolib.models.spadop', resample=resample)
    _mod('luolib.utils', flatten=flatten, spatialize=spatialize, channel_first=channel_first, channel_last=channel_last,
         pairwise_forward=pairwise_forward, load_pt_zst=None)
    _mod('luolib.losses', zero_loss=zero_loss, sigmoid_focal_loss=sigmoid_focal_loss, bce_with_binary_label=bce_with_binary_label)
    _mod('luolib.lightning', LightningModule=_LightningModule)
    _mod('luolib.lightning.peft', PeftMixin=_PeftMixin)

    class StrEnum(str):
        pass

    _mod('monai')
    _mod('monai.utils', StrEnum=StrEnum, str2bool=lambda v: str(v).lower() in ('1', 'true', 'yes'))
    _mod('monai.networks')
    _mod('monai.networks.blocks', SABlock=_SABlock, TransformerBlock=_TransformerBlock)
    _mod('monai.data', box_pair_giou=box_pair_giou, convert_box_mode=convert_box_mode)
    _mod('monai.data.box_utils', CenterSizeMode=CenterSizeMode)

    _mod('xformers')
    xo = _mod('xformers.ops', memory_efficient_attention=memory_efficient_attention)
    fm = _mod('xformers.ops.fmha', BlockDiagonalMask=BlockDiagonalMask)
    ab = _mod('xformers.ops.fmha.attn_bias', BlockDiagonalCausalMask=BlockDiagonalCausalMask, BlockDiagonalMask=BlockDiagonalMask)
    xo.fmha = fm
    fm.attn_bias = ab

    _mod('peft', PeftModel=object)
    _mod('lightning')
    _mod('lightning.pytorch')

    class _Precision:
        pass

    class _HalfPrecision:
        def __init__(self, precision='bf16-true'):
            self._desired_input_dtype = torch.bfloat16

        def convert_module(self, m):
            return m.to(torch.bfloat16)

        def convert_input(self, data):
            def cv(x):
                if torch.is_tensor(x):
                    return x.to(torch.bfloat16) if x.is_floating_point() else x
                if isinstance(x, dict):
                    return {k: cv(v) for k, v in x.items()}
                if isinstance(x, (list, tuple)):
                    return type(x)(cv(v) for v in x)
                return x
            return cv(data)

    _mod('lightning.pytorch.plugins', HalfPrecision=_HalfPrecision, Precision=_Precision)

    def compose(*fs):
        def f(x):
            for g in reversed(fs):
                x = g(x)
            return x
        return f

    def dissoc(d, *keys):
        return {k: v for k, v in d.items() if k not in keys}

    _mod('cytoolz', compose=compose, dissoc=dissoc, concat=lambda seqs: [x for s in seqs for x in s])
    _mod('jsonargparse', class_from_function=lambda fn, *a, **k: fn)
    if 'torchvision' not in sys.modules:
        _mod('torchvision')
        _mod('torchvision.transforms')

    # package shells: import sub-modules by path without running the heavy __init__ files
    def shell(name: str, path: Path):
        m = types.ModuleType(name)
        m.__path__ = [str(path)]
        sys.modules[name] = m
        parent, _, child = name.rpartition('.')
        if parent:
            setattr(sys.modules[parent], child, m)
        return m

    shell('mmmm', REF / 'mmmm')
    shell('mmmm.models', REF / 'mmmm/models')
    shell('mmmm.data', REF / 'mmmm/data')
    _mod('mmmm.data.utils', LANGUAGE_TOKEN_TYPE=0, VISION_TOKEN_TYPE=1)

    class MMMMTokenizer:
        pass

    _mod('mmmm.tokenizer', MMMMTokenizer=MMMMTokenizer)
    importlib.import_module('mmmm.utils')
    importlib.import_module('mmmm.data.defs')
    res = importlib.import_module('mmmm.models.resample')
    sys.modules['mmmm.models'].resample = res
    importlib.import_module('mmmm.models.loss')
    shell('mmmm.models.cogvlm', REF / 'mmmm/models/cogvlm')
    cfg = importlib.import_module('mmmm.models.cogvlm.configuration_cogvlm')
    mc = importlib.import_module('mmmm.models.cogvlm.modeling_cogvlm')
    sys.modules['mmmm.models.cogvlm'].CogVLMConfig = cfg.CogVLMConfig
    sys.modules['mmmm.models.cogvlm'].CogVLMForCausalLM = mc.CogVLMForCausalLM
    importlib.import_module('mmmm.models.segvol')
    importlib.import_module('mmmm.models.mmmm')
    _installed = True


@dataclass
class RefModules:
    modeling_cogvlm: types.ModuleType
    visual: types.ModuleType
    mmmm: types.ModuleType
    sam: types.ModuleType
    build_sam: types.ModuleType
    loss: types.ModuleType
    resample: types.ModuleType
    utils: types.ModuleType
    config: types.ModuleType


def load() -> RefModules:
    install()
    g = sys.modules
    return RefModules(
        modeling_cogvlm=g['mmmm.models.cogvlm.modeling_cogvlm'],
        visual=g['mmmm.models.cogvlm.visual'],
        mmmm=g['mmmm.models.mmmm'],
        sam=g['mmmm.models.segvol.modeling.sam'],
        build_sam=g['mmmm.models.segvol.build_sam'],
        loss=g['mmmm.models.loss'],
        resample=g['mmmm.models.resample'],
        utils=g['mmmm.utils'],
        config=g['mmmm.models.cogvlm.configuration_cogvlm'],
    )


def load_data_utils():
    """The reference's `mmmm/data/utils.py` (prepare_vlm_inputs, get_text_position_ids) imported under a private name,
    with empty stand-ins for the I/O libraries its *other* functions use (monai MetaTensor / transforms, nibabel, luolib
    helpers, the sentencepiece-backed tokenizer class). SURVEY §8f N1. Development container only."""
    install()
    import importlib.util
    name = 'mmmm.data._ref_utils'
    if name in sys.modules:
        return sys.modules[name]
    if not hasattr(sys.modules['monai.data'], 'MetaTensor'):
        sys.modules['monai.data'].MetaTensor = object
    for n in ('nibabel', 'monai.transforms'):
        if n not in sys.modules:
            _mod(n)
    sys.modules['luolib.utils'].load_pt_zst = None
    if 'luolib.utils.misc' not in sys.modules:
        _mod('luolib.utils.misc', min_stem=None)
    if 'mmmm.tokenizer' not in sys.modules or not hasattr(sys.modules['mmmm.tokenizer'], 'MMMMTokenizer'):
        _mod('mmmm.tokenizer', MMMMTokenizer=object)
    if 'mmmm.data.sparse' not in sys.modules:
        _mod('mmmm.data.sparse', Sparse=object)
    spec = importlib.util.spec_from_file_location(name, REF / 'mmmm/data/utils.py')
    mod = importlib.util.module_from_spec(spec)
    mod.__package__ = 'mmmm.data'
    sys.modules[name] = mod
    src = (REF / 'mmmm/data/utils.py').read_text()
    # the two token-type constants live at the bottom of the file; the functions reference them as globals
    exec(compile(src, str(REF / 'mmmm/data/utils.py'), 'exec'), mod.__dict__)
    return mod
