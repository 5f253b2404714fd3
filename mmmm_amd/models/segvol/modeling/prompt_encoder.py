"""SAM prompt encoder (reference segvol/modeling/prompt_encoder.py). Only the text-prompt path is on the training
step: sparse = the projected </p> hidden state, dense = `no_mask_embed` broadcast, plus the random-Fourier positional
encoding of the 3-D token grid. Point / box / mask branches keep their parameters (checkpoint compatibility) but are
frozen by `_freeze_sam_unused` and never executed."""
from __future__ import annotations

import math

import torch
from torch import nn


class LayerNorm2d(nn.Module):
    def __init__(self, num_channels: int, eps: float = 1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(num_channels))
        self.bias = nn.Parameter(torch.zeros(num_channels))
        self.eps = eps


class PositionEmbeddingRandom(nn.Module):
    def __init__(self, num_pos_feats: int = 64, scale: float | None = None):
        super().__init__()
        if scale is None or scale <= 0.0:
            scale = 1.0
        self.register_buffer('positional_encoding_gaussian_matrix', scale * torch.randn((3, num_pos_feats)))

    def forward(self, size: tuple) -> torch.Tensor:
        """-> channel-last [n_tokens, C] for a grid the reference calls (h, w, d) (prompt_encoder.py:179-191)"""
        g = self.positional_encoding_gaussian_matrix
        h, w, d = size
        grid = g.new_ones(h, w, d)
        y = (grid.cumsum(0) - 0.5) / h
        x = (grid.cumsum(1) - 0.5) / w
        z = (grid.cumsum(2) - 0.5) / d
        coords = 2 * torch.stack([x, y, z], dim=-1) - 1
        coords = 2 * math.pi * (coords @ g)
        return torch.cat([coords.sin(), coords.cos()], dim=-1).reshape(h * w * d, -1)


class PromptEncoder(nn.Module):
    def __init__(self, embed_dim: int, mask_in_chans: int = 16):
        super().__init__()
        self.embed_dim = embed_dim
        self.pe_layer = PositionEmbeddingRandom(embed_dim // 2)
        self.num_point_embeddings = 4
        self.point_embeddings = nn.ModuleList([nn.Embedding(1, embed_dim) for _ in range(4)])
        self.not_a_point_embed = nn.Embedding(1, embed_dim)
        self.mask_downscaling = nn.Sequential(
            nn.Conv2d(1, mask_in_chans // 4, kernel_size=2, stride=2), LayerNorm2d(mask_in_chans // 4), nn.GELU(),
            nn.Conv2d(mask_in_chans // 4, mask_in_chans, kernel_size=2, stride=2), LayerNorm2d(mask_in_chans), nn.GELU(),
            nn.Conv2d(mask_in_chans, embed_dim, kernel_size=1),
        )
        self.no_mask_embed = nn.Embedding(1, embed_dim)

    def get_dense_pe(self, image_embedding_shape: tuple) -> torch.Tensor:
        return self.pe_layer(image_embedding_shape)

    def forward(self, image_embed_shape: tuple, points=None, boxes=None, masks=None, text_embedding: torch.Tensor | None = None):
        if points is not None or boxes is not None or masks is not None:
            raise NotImplementedError('point / box / mask prompts are not on the VividMed training step')
        sparse = text_embedding.unsqueeze(1)                      # [P, 1, C]
        dense = self.no_mask_embed.weight                         # [1, C], broadcast over the grid by the caller
        return sparse, dense
