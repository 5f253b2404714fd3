"""SAM two-way transformer (reference segvol/modeling/transformer.py) on the fp32 HIP kernels: projections on
vm_gemm_f32, attention on vm_attn_*_f32 (head_dim 96 for token self-attention, 48 for the cross-attentions at
embedding_dim 768), LayerNorm (+residual) fused. Same module / parameter names."""
from __future__ import annotations

import torch
from torch import nn

from .... import functional as Fh
from ...lora import Linear


def _ln(m: nn.LayerNorm, x: torch.Tensor, residual: torch.Tensor | None = None) -> torch.Tensor:
    shape = x.shape
    y = Fh.layer_norm(x.reshape(-1, shape[-1]), m.weight, m.bias, m.eps,
                      residual=None if residual is None else residual.reshape(-1, shape[-1]))
    return y.view(shape)


class Attention(nn.Module):
    def __init__(self, embedding_dim: int, num_heads: int, downsample_rate: int = 1):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.internal_dim = embedding_dim // downsample_rate
        self.num_heads = num_heads
        assert self.internal_dim % num_heads == 0, 'num_heads must divide embedding_dim.'
        self.head_dim = self.internal_dim // num_heads
        self.q_proj = Linear(embedding_dim, self.internal_dim)
        self.k_proj = Linear(embedding_dim, self.internal_dim)
        self.v_proj = Linear(embedding_dim, self.internal_dim)
        self.out_proj = Linear(self.internal_dim, embedding_dim)
        self.scale = self.head_dim ** -0.5

    def forward(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
        """q [B, Lq, C], k/v [B, Lk, C] -> [B, Lq, C]"""
        out = Fh.attention_f32(self.q_proj(q), self.k_proj(k), self.v_proj(v), self.num_heads, self.head_dim, self.scale)
        return self.out_proj(out)


class MLPBlock(nn.Module):
    def __init__(self, embedding_dim: int, mlp_dim: int):
        super().__init__()
        self.lin1 = Linear(embedding_dim, mlp_dim)
        self.lin2 = Linear(mlp_dim, embedding_dim)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.lin2(Fh.relu(self.lin1(x)))


class TwoWayAttentionBlock(nn.Module):
    """(1) token self-attention (2) tokens -> image cross-attention (3) token MLP (4) image -> tokens
    cross-attention, each followed by a LayerNorm (reference :159-190)"""

    def __init__(self, embedding_dim: int, num_heads: int, mlp_dim: int = 2048, attention_downsample_rate: int = 2,
                 skip_first_layer_pe: bool = False):
        super().__init__()
        self.self_attn = Attention(embedding_dim, num_heads)
        self.norm1 = nn.LayerNorm(embedding_dim)
        self.cross_attn_token_to_image = Attention(embedding_dim, num_heads, downsample_rate=attention_downsample_rate)
        self.norm2 = nn.LayerNorm(embedding_dim)
        self.mlp = MLPBlock(embedding_dim, mlp_dim)
        self.norm3 = nn.LayerNorm(embedding_dim)
        self.norm4 = nn.LayerNorm(embedding_dim)
        self.cross_attn_image_to_token = Attention(embedding_dim, num_heads, downsample_rate=attention_downsample_rate)
        self.skip_first_layer_pe = skip_first_layer_pe

    def forward(self, queries, keys, query_pe, key_pe):
        if self.skip_first_layer_pe:
            queries = _ln(self.norm1, self.self_attn(q=queries, k=queries, v=queries))
        else:
            q = queries + query_pe
            queries = _ln(self.norm1, queries + self.self_attn(q=q, k=q, v=queries))
        q, k = queries + query_pe, keys + key_pe
        queries = _ln(self.norm2, queries + self.cross_attn_token_to_image(q=q, k=k, v=keys))
        queries = _ln(self.norm3, queries + self.mlp(queries))
        q = queries + query_pe
        keys = _ln(self.norm4, keys + self.cross_attn_image_to_token(q=k, k=q, v=queries))
        return queries, keys


class TwoWayTransformer(nn.Module):
    def __init__(self, depth: int, embedding_dim: int, num_heads: int, mlp_dim: int, attention_downsample_rate: int = 2):
        super().__init__()
        self.depth, self.embedding_dim, self.num_heads, self.mlp_dim = depth, embedding_dim, num_heads, mlp_dim
        self.layers = nn.ModuleList([
            TwoWayAttentionBlock(embedding_dim, num_heads, mlp_dim, attention_downsample_rate, skip_first_layer_pe=(i == 0))
            for i in range(depth)
        ])
        self.final_attn_token_to_image = Attention(embedding_dim, num_heads, downsample_rate=attention_downsample_rate)
        self.norm_final_attn = nn.LayerNorm(embedding_dim)
        self.gradient_checkpointing = False

    def forward(self, image_embedding: torch.Tensor, image_pe: torch.Tensor, queries: torch.Tensor, query_pe: torch.Tensor):
        """image_embedding / image_pe: channel-last [B, Ns, C]; queries [B, n_tok, C] -> (queries, keys)"""
        keys = image_embedding
        for layer in self.layers:
            queries, keys = layer(queries, keys, query_pe, image_pe)
        q, k = queries + query_pe, keys + image_pe
        queries = _ln(self.norm_final_attn, queries + self.final_attn_token_to_image(q=q, k=k, v=keys))
        return queries, keys
