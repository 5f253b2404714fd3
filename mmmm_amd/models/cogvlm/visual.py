"""EVA2-CLIP-E image encoder of CogVLM on the MI355X HIP kernels.

Module / parameter names follow /root/reference/mmmm/models/cogvlm/visual.py. All images of a batch are
packed into one `[sum Nv, 1792]` sequence (var-len, per-image blocks) like the reference does with
xformers' BlockDiagonalMask; attention is the non-causal var-len flash kernel at head_dim 112.
"""
from __future__ import annotations

from argparse import Namespace

import torch
from torch import nn
import torch.nn.functional as F
from torch.utils.checkpoint import checkpoint

from ... import functional as Fh
from ...param import NoWeightDecayParameter
from ..lora import ActivationBudget, Linear
from ..resample import Downsample, resample


class ParameterWrapper(nn.Module):
    """mmmm/utils.py:62-77 — a bare parameter exposed as `<name>.weight` so PEFT's modules_to_save can hold it"""
    def __init__(self, weight: torch.Tensor):
        super().__init__()
        self.weight = weight if isinstance(weight, nn.Parameter) else nn.Parameter(weight)

    @classmethod
    def wrap(cls, module: nn.Module, state_dict: dict, prefix: str):
        """checkpoints written before the wrapper existed hold `<prefix><name>`; this module tree wants
        `<prefix><name>.weight` (mmmm/utils.py:68-77)"""
        for name, child in module.named_children():
            if isinstance(child, cls) and (w := state_dict.pop(f'{prefix}{name}', None)) is not None:
                state_dict[f'{prefix}{name}.weight'] = w


class PatchEmbedding(nn.Module):
    def __init__(self, config: Namespace):
        super().__init__()
        self.proj = Downsample(config.in_channels, config.hidden_size, config.patch_size, interpolate_2d=True)
        self.pos_embed_shape = tuple(config.pos_embed_shape)
        self.cls_embedding = ParameterWrapper(NoWeightDecayParameter(torch.zeros(1, config.hidden_size)))
        self.cls_pos_embed = ParameterWrapper(NoWeightDecayParameter(torch.zeros(1, config.hidden_size)))
        self.position_embedding = ParameterWrapper(NoWeightDecayParameter(torch.zeros(1, config.hidden_size, *config.pos_embed_shape)))
        self.pt_pos_embed_shape = tuple(getattr(config, 'pt_pos_embed_shape', config.pos_embed_shape[-2:]))

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        """state-dict adapter (reference visual.py:37-57): an EVA2-CLIP / cogvlm-chat-hf checkpoint stores ONE 2-D table
        `[1 + h*w, C]` (row 0 = cls); it becomes `cls_pos_embed [1, C]` + `position_embedding [1, C, d, h, w]` (the h x w
        table resampled to this model's grid and repeated along depth). A PEFT modules_to_save copy saved at another grid
        is resampled too."""
        key = f'{prefix}position_embedding.weight'
        saved = f'{prefix}position_embedding.modules_to_save.default.weight'
        if (pe := state_dict.get(key)) is not None and pe.ndim == 2:
            h, w = self.pt_pos_embed_shape
            grid = pe[1:].reshape(h, w, -1).permute(2, 0, 1)[None]                 # [1, C, h, w]
            if (h, w) != tuple(self.pos_embed_shape[-2:]):
                grid = resample(grid, self.pos_embed_shape[-2:])
            del state_dict[key]
            state_dict[f'{prefix}cls_pos_embed'] = pe[0:1]
            state_dict[f'{prefix}position_embedding'] = grid[:, :, None].expand(-1, -1, self.pos_embed_shape[0], -1, -1).contiguous()
        elif (pe := state_dict.get(saved)) is not None and tuple(pe.shape[2:]) != tuple(self.position_embedding.weight.shape[2:]):
            state_dict[saved] = resample(pe, self.position_embedding.weight.shape[2:])
        ParameterWrapper.wrap(self, state_dict, prefix)
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def forward(self, image_list: list[torch.Tensor], patch_size_list: list[tuple]):
        """-> packed tokens [sum Nv, C], cu_seqlens (host list), grid shapes (visual.py:59-77)"""
        xs, shapes = [], []
        cls = self.cls_embedding.weight + self.cls_pos_embed.weight
        pos_cache: dict[tuple, torch.Tensor] = {}        # one resample per distinct grid, not per image
        for image, patch in zip(image_list, patch_size_list):
            x, shape = self.proj(image, patch)
            if shape not in pos_cache:
                pos = resample(self.position_embedding.weight, shape)        # [1, C, d, h, w]
                pos_cache[shape] = pos[0].flatten(1).t().to(x.dtype)
            x = x + pos_cache[shape]
            xs.append(torch.cat([cls.to(x.dtype), x], dim=0))
            shapes.append(shape)
        lens = [t.shape[0] for t in xs]
        return torch.cat(xs, dim=0), lens, shapes


class Attention(nn.Module):
    def __init__(self, config: Namespace):
        super().__init__()
        self.num_heads = config.num_heads
        self.head_dim = config.hidden_size // config.num_heads
        self.scale = self.head_dim ** -0.5
        self.query_key_value = Linear(config.hidden_size, config.hidden_size * 3)
        self.dense = Linear(config.hidden_size, config.hidden_size)

    def forward(self, x: torch.Tensor, cu: torch.Tensor, max_len: int):
        """-> (attention output, x passed through the first linear: the block's residual — Linear.forward `fork`)"""
        qkv, x = self.query_key_value(x, fork=True)
        out = Fh.attention(qkv, cu, max_len, self.num_heads, self.head_dim, self.scale, False)
        return self.dense(out), x


class MLP(nn.Module):
    def __init__(self, config: Namespace):
        super().__init__()
        self.fc1 = Linear(config.hidden_size, config.intermediate_size)
        self.fc2 = Linear(config.intermediate_size, config.hidden_size)

    def forward(self, x: torch.Tensor):
        h, x = self.fc1(x, fork=True)
        return self.fc2(Fh.gelu(h)), x


class TransformerLayer(nn.Module):
    """LayerNorm on the branch OUTPUT: x + LN(attn(x)), x + LN(mlp(x)) (visual.py:134-141)"""
    def __init__(self, config: Namespace):
        super().__init__()
        self.input_layernorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.attention = Attention(config)
        self.mlp = MLP(config)
        self.post_attention_layernorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)

    def forward(self, x: torch.Tensor, cu: torch.Tensor, max_len: int) -> torch.Tensor:
        ln1, ln2 = self.input_layernorm, self.post_attention_layernorm
        # x feeds the branch's first linear AND the residual: the linear hands x back (`fork`) so that the residual's gradient is summed
        # into the branch's input gradient by the dgrad GEMM's epilogue, not by an element-wise add over [tokens, hidden]
        a, x = self.attention(x, cu, max_len)
        x = Fh.layer_norm(a, ln1.weight, ln1.bias, ln1.eps, residual=x)
        m, x = self.mlp(x)
        return Fh.layer_norm(m, ln2.weight, ln2.bias, ln2.eps, residual=x)


class Transformer(nn.Module):
    def __init__(self, config: Namespace):
        super().__init__()
        self.gradient_checkpointing = False
        self.layers = nn.ModuleList([TransformerLayer(config) for _ in range(config.num_hidden_layers)])

    def forward(self, x: torch.Tensor, cu: torch.Tensor, max_len: int) -> torch.Tensor:
        ckpt = self.gradient_checkpointing and self.training and torch.is_grad_enabled()
        n_keep = 0
        if ckpt:
            # saved per row and layer: ~9 hidden-width tensors (LN/linear inputs, q/k/v, attention output), the fc1
            # output and its GELU, the LoRA projections; bf16
            l0 = self.layers[0]
            d, f = l0.mlp.fc1.in_features, l0.mlp.fc1.out_features
            n_keep = ActivationBudget.claim(len(self.layers), x.shape[0] * (9 * d + 2 * f + 4 * 64) * x.element_size())
        for i, layer in enumerate(self.layers):
            if ckpt and i >= n_keep:
                x = checkpoint(layer, x, cu, max_len, use_reentrant=False, preserve_rng_state=False)
            else:
                x = layer(x, cu, max_len)
        return x


class GLU(nn.Module):
    def __init__(self, config, in_features: int):
        super().__init__()
        self.linear_proj = Linear(in_features, config.hidden_size, bias=False)
        self.norm1 = nn.LayerNorm(config.hidden_size)
        self.dense_h_to_4h = Linear(config.hidden_size, config.intermediate_size, bias=False)
        self.gate_proj = Linear(config.hidden_size, config.intermediate_size, bias=False)
        self.dense_4h_to_h = Linear(config.intermediate_size, config.hidden_size, bias=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self.linear_proj(x)
        x = Fh.gelu(Fh.layer_norm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps))
        x = Fh.silu_mul(self.gate_proj(x), self.dense_h_to_4h(x))
        return self.dense_4h_to_h(x)


class EVA2CLIPModel(nn.Module):
    def __init__(self, config):
        super().__init__()
        vc = Namespace(**config.vision_config)
        self.patch_embedding = PatchEmbedding(vc)
        self.transformer = Transformer(vc)
        self.linear_proj = GLU(config, in_features=vc.hidden_size)
        self.boi = NoWeightDecayParameter(torch.zeros(1, 1, config.hidden_size))
        self.eoi = NoWeightDecayParameter(torch.zeros(1, 1, config.hidden_size))

    def forward(self, image: list[torch.Tensor], patch_size: list[tuple], pool_size_list: list[tuple]) -> list[torch.Tensor]:
        """-> per image [Np + 2, hidden] (boi, pooled patches through the GLU adapter, eoi) — visual.py:192-208"""
        x, lens, shapes = self.patch_embedding(image, patch_size)
        cu_host = [0]
        for n in lens:
            cu_host.append(cu_host[-1] + n)
        cu = Fh.cu_seqlens_tensor(lens, x.device)
        x = self.transformer(x, cu, max(lens))
        pooled, counts = [], []
        for i, (shape, pool) in enumerate(zip(shapes, pool_size_list)):
            t = x[cu_host[i] + 1:cu_host[i + 1]]
            if any(p > 1 for p in pool):
                t = t.t().reshape(1, -1, *shape)
                t = F.max_pool3d(t, tuple(pool))
                t = t.flatten(2)[0].t()
            pooled.append(t)
            counts.append(t.shape[0])
        y = self.linear_proj(torch.cat(pooled, dim=0).contiguous())       # one GEMM chain for the whole batch
        outs = []
        for t in y.split(counts):
            outs.append(torch.cat([self.boi[0].to(t.dtype), t, self.eoi[0].to(t.dtype)], dim=0))
        return outs
