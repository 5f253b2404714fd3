"""SAM ViT-B image encoder (reference segvol/modeling/image_encoder.py; pre-LN blocks with MONAI's parameter names
`blocks.N.{norm1, attn.{qkv,out_proj}, norm2, mlp.{linear1,linear2}}`), fp32, all images of the batch packed into
one var-len sequence; attention on vm_attn_*_f32 (head_dim 64)."""
from __future__ import annotations


import torch
from torch import nn
from torch.utils.checkpoint import checkpoint

from .... import functional as Fh
from ....param import NoWeightDecayParameter
from ...cogvlm.visual import ParameterWrapper
from ...lora import Linear
from ...resample import Downsample, resample

ENCODER_F32_SPLIT = 2
# arithmetic of the blocks' attention products (functional.self_attention_f32): -1 = the same as the blocks' GEMMs
ATTN_F32_SPLIT = -1


class PatchEmbeddingBlock(nn.Module):
    def __init__(self, in_channels: int, patch_size, pos_embed_shape: tuple, hidden_size: int, num_heads: int, dropout_rate: float = 0.0,
                 pt_in_channels=None, pt_patch_size=None, pt_pos_embed_shape=None):
        super().__init__()
        self.proj = Downsample(in_channels, hidden_size, patch_size)
        self.position_embeddings = ParameterWrapper(NoWeightDecayParameter(torch.zeros(1, hidden_size, *pos_embed_shape)))
        self.pt_in_channels, self.pt_patch_size, self.pt_pos_embed_shape = pt_in_channels, pt_patch_size, pt_pos_embed_shape

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        """state-dict adapter (reference image_encoder.py:82-119). A SegVol checkpoint holds the patch embedding as a
        Linear over flattened `(p0 p1 p2 ci)` patches plus a `[1, d*h*w, C]` position table: the Linear becomes the conv
        kernel `[co, ci, p0, p1, p2]` (resampled when the patch size differs, a 1-channel kernel spread over the input
        channels), the table becomes `[1, C, d, h, w]` resampled to this model's grid. A table saved by this code base at
        another grid is resampled as well."""
        lin = f'{prefix}patch_embeddings.1.weight'
        pos_key = f'{prefix}position_embeddings'
        grid = tuple(self.position_embeddings.weight.shape[2:])
        if (w := state_dict.get(lin)) is not None and w.ndim == 2:
            del state_dict[lin]
            p0, p1, p2 = self.pt_patch_size
            ci = self.pt_in_channels
            w = w.reshape(w.shape[0], p0, p1, p2, ci).permute(0, 4, 1, 2, 3)
            if tuple(self.pt_patch_size) != tuple(self.proj.kernel_size):
                w = resample(w, self.proj.kernel_size, scale=True)
            if ci == 1 and self.proj.in_channels != 1:
                w = w.expand(-1, self.proj.in_channels, -1, -1, -1) / self.proj.in_channels
            state_dict[f'{prefix}proj.weight'] = w.contiguous()
            state_dict[f'{prefix}proj.bias'] = state_dict.pop(f'{prefix}patch_embeddings.1.bias')
            d, h, w_ = self.pt_pos_embed_shape
            pe = state_dict[pos_key]
            pe = pe.reshape(1, d, h, w_, pe.shape[-1]).permute(0, 4, 1, 2, 3)
            state_dict[pos_key] = resample(pe, grid).contiguous()
        elif (pe := state_dict.get(f'{pos_key}.weight')) is not None and tuple(pe.shape[2:]) != grid:
            state_dict[f'{pos_key}.weight'] = resample(pe, grid)
        ParameterWrapper.wrap(self, state_dict, prefix)
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def forward(self, image_list, patch_size_list):
        xs, shapes = [], []
        pos_cache: dict[tuple, torch.Tensor] = {}
        for image, patch in zip(image_list, patch_size_list):
            x, shape = self.proj(image, patch)
            if shape not in pos_cache:
                pos_cache[shape] = resample(self.position_embeddings.weight, shape)[0].flatten(1).t().to(x.dtype)
            xs.append(x + pos_cache[shape])
            shapes.append(shape)
        return torch.cat(xs, dim=0), shapes, [t.shape[0] for t in xs]


class SABlock(nn.Module):
    def __init__(self, hidden_size: int, num_heads: int, qkv_bias: bool = False):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = hidden_size // num_heads
        self.scale = self.head_dim ** -0.5
        self.out_proj = Linear(hidden_size, hidden_size)
        self.qkv = Linear(hidden_size, hidden_size * 3, bias=qkv_bias)

    def forward(self, x, cu, max_len, residual=None):
        split = ATTN_F32_SPLIT if ATTN_F32_SPLIT >= 0 else self.qkv.f32_split
        out = Fh.self_attention_f32(self.qkv(x), self.num_heads, self.head_dim, self.scale, cu, max_len, f32_split=split)
        return self.out_proj(out, residual=residual)


class MLPBlock(nn.Module):
    def __init__(self, hidden_size: int, mlp_dim: int):
        super().__init__()
        self.linear1 = Linear(hidden_size, mlp_dim)
        self.linear2 = Linear(mlp_dim, hidden_size)

    def forward(self, x, residual=None):
        return self.linear2(Fh.gelu(self.linear1(x)), residual=residual)


class TransformerBlock(nn.Module):
    def __init__(self, hidden_size: int, mlp_dim: int, num_heads: int, qkv_bias: bool = False):
        super().__init__()
        self.mlp = MLPBlock(hidden_size, mlp_dim)
        self.norm1 = nn.LayerNorm(hidden_size)
        self.attn = SABlock(hidden_size, num_heads, qkv_bias)
        self.norm2 = nn.LayerNorm(hidden_size)

    def forward(self, x, cu, max_len):
        n1, n2 = self.norm1, self.norm2
        # the residual adds ride in the out_proj / linear2 GEMM epilogues; their gradients come back through the norms' second output
        # (`fork`) and are summed into the norm's input gradient by its kernel
        h, x = Fh.layer_norm(x, n1.weight, n1.bias, n1.eps, fork=True)
        x = self.attn(h, cu, max_len, residual=x)
        h, x = Fh.layer_norm(x, n2.weight, n2.bias, n2.eps, fork=True)
        return self.mlp(h, residual=x)


class ImageEncoderViT(nn.Module):
    def __init__(self, in_channels: int, patch_size, pos_embed_shape: tuple, hidden_size: int = 768, mlp_dim: int = 3072, num_layers: int = 12,
                 num_heads: int = 12, dropout_rate: float = 0.0, qkv_bias: bool = False, pt_in_channels=None, pt_patch_size=None,
                 pt_pos_embed_shape=None):
        super().__init__()
        self.patch_embedding = PatchEmbeddingBlock(in_channels, patch_size, pos_embed_shape, hidden_size, num_heads, dropout_rate,
                                                   pt_in_channels, pt_patch_size, pt_pos_embed_shape)
        self.blocks = nn.ModuleList([TransformerBlock(hidden_size, mlp_dim, num_heads, qkv_bias) for _ in range(num_layers)])
        self.norm = nn.LayerNorm(hidden_size)
        self.gradient_checkpointing = False
        # arithmetic of the 12 blocks' fp32 GEMMs (~85 % of the grounding heads' GEMM time): split-bf16 with 3 products (each operand
        # as hi + lo bf16, the lo*lo term dropped: ~2^-16 relative per product, fp32 accumulation) instead of the heads' default 6.
        # The prompt gradients — what broke the 1e-4 bar of the fp32 islands under a PROCESS-WIDE mode 2 (9e-4 through iSAM) — flow
        # through the mask decoder only, which keeps 6 products. Measured with this default at true width against the fp32 oracle
        # (tests/test_config0_gpu.py, profiles/r3_parity_report.json): masks 1.3e-6, boxes 2.8e-7, discriminator 7.3e-7, the
        # encoder's own parameter gradients 8e-7 .. 3.3e-6 (bound 1e-4). VM_ENC_F32_SPLIT=3 (or 1) restores 6 products (exact f32 MFMA).
        for m in self.blocks.modules():
            if isinstance(m, Linear):
                m.f32_split = ENCODER_F32_SPLIT

    def forward(self, image: list[torch.Tensor], patch_size: list[tuple]):
        """-> per image: channel-last tokens [Ns, C] and the (d, h, w) grid"""
        x, shapes, lens = self.patch_embedding(image, patch_size)
        cu_host = [0]
        for n in lens:
            cu_host.append(cu_host[-1] + n)
        cu = Fh.cu_seqlens_tensor(lens, x.device)
        for blk in self.blocks:
            if self.gradient_checkpointing and self.training and x.requires_grad:
                x = checkpoint(blk, x, cu, max(lens), use_reentrant=False, preserve_rng_state=False)
            else:
                x = blk(x, cu, max(lens))
        x = Fh.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        return [x[cu_host[i]:cu_host[i + 1]] for i in range(len(lens))], shapes
