"""SAM / SegVol mask decoder (reference segvol/modeling/mask_decoder.py) in channel-last layout on the fp32 HIP
kernels: the two ConvTranspose3d(k=2,s=2) upscalers are GEMMs + a pixel-shuffle view, the channel LayerNorm is the
row-wise LN kernel, and the hyper-network product `einsum('n m c, n c ... -> n m ...')` plus the text-similarity term
is ONE GEMM per prompt against hyper_in + txt_align(text) (functional.hyper_product: one autograd node for all prompts)."""
from __future__ import annotations

import torch
from torch import nn

from .... import functional as Fh
from ...lora import Linear
from ...resample import Upsample
from .transformer import TwoWayTransformer


class LayerNormNd(nn.LayerNorm):
    """LayerNorm over the channel dimension of an N-d feature map (reference :15-26); here the map is already
    channel-last so it is the plain row-wise kernel."""
    def __init__(self, num_channels: int, contiguous: bool = True):
        super().__init__(num_channels)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        shape = x.shape
        return Fh.layer_norm(x.reshape(-1, shape[-1]), self.weight, self.bias, self.eps).view(shape)


class MLP(nn.Module):
    def __init__(self, input_dim: int, hidden_dim: int, output_dim: int, num_layers: int, sigmoid_output: bool = False):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))
        self.sigmoid_output = sigmoid_output

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = layer(x)
            if i < self.num_layers - 1:
                x = Fh.relu(x)
        return torch.sigmoid(x) if self.sigmoid_output else x


class _GELU(nn.Module):
    def forward(self, x):
        return Fh.gelu(x)


class MaskDecoder(nn.Module):
    def __init__(self, *, transformer_dim: int, transformer: TwoWayTransformer, num_instances: int = 3):
        super().__init__()
        self.transformer_dim = transformer_dim
        self.transformer = transformer
        self.num_instances = num_instances
        self.iou_token = nn.Embedding(1, transformer_dim)
        self.num_mask_tokens = num_instances + 1
        self.mask_tokens = nn.Embedding(self.num_mask_tokens, transformer_dim)
        self.output_upscaling = nn.Sequential(
            Upsample(transformer_dim, transformer_dim // 4, cnt=0),
            LayerNormNd(transformer_dim // 4),
            _GELU(),
            Upsample(transformer_dim // 4, transformer_dim // 8, cnt=1),
            _GELU(),
        )
        self.output_hypernetworks_mlps = nn.ModuleList([MLP(transformer_dim, transformer_dim, transformer_dim // 8, 3) for _ in range(2)])
        self.txt_align_upscaled_embedding = Linear(transformer_dim, transformer_dim // 8)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        """state-dict adapter (reference mask_decoder.py:76-87): SegVol stores the first upscaling LayerNorm as a per-voxel
        affine `[C, d, h, w]` — reduced to its per-channel mean; a checkpoint with fewer mask tokens fills the leading
        rows of this model's (freshly initialised) table."""
        ln = f'{prefix}output_upscaling.1.'
        if (w := state_dict.get(f'{ln}weight')) is not None and w.ndim == 4:
            state_dict[f'{ln}weight'] = w.flatten(1).mean(dim=1)
            state_dict[f'{ln}bias'] = state_dict[f'{ln}bias'].flatten(1).mean(dim=1)
        if (pt := state_dict.get(f'{prefix}mask_tokens.weight')) is not None:
            table = self.mask_tokens.weight.detach().clone()
            table[:pt.shape[0]] = pt.to(table.dtype)[:table.shape[0]]
            state_dict[f'{prefix}mask_tokens.weight'] = table[:self.num_mask_tokens]
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def forward(self, image_embeddings: torch.Tensor, image_pe: torch.Tensor, sparse_prompt_embeddings: torch.Tensor,
                dense_prompt_embeddings: torch.Tensor, text_embedding: torch.Tensor, patch_size_z: int, grid: tuple,
                need_masks: bool = True):
        """image_embeddings: channel-last tokens of the image each prompt refers to, already replicated per prompt
        [P, Ns, C] (the reference's `repeat_interleave`, :110-116) — prompts of several images with the same grid are
        decoded in ONE pass; image_pe [Ns, C]; sparse [P, 1, C]; dense [1, C]
        -> masks [P, M, D', H', W'], mask_tokens_out [P, M, C]   (reference :89-149)"""
        P = sparse_prompt_embeddings.shape[0]
        C = self.transformer_dim
        out_tokens = torch.cat([self.iou_token.weight, self.mask_tokens.weight], dim=0)
        tokens = torch.cat([out_tokens[None].expand(P, -1, -1), sparse_prompt_embeddings], dim=1)
        src = image_embeddings + dense_prompt_embeddings
        pos = image_pe[None].expand(P, -1, -1).contiguous()
        hs, src = self.transformer(src, pos, tokens, tokens)
        mask_tokens_out = hs[:, 1:1 + self.num_mask_tokens]
        if not need_masks:
            return None, mask_tokens_out
        d, h, w = grid
        up = src.view(P, d, h, w, C)
        for i, module in enumerate(self.output_upscaling):
            up = module(up, patch_size_z) if i % 3 == 0 else module(up)
        hyper = torch.stack([self.output_hypernetworks_mlps[int(i > 0)](mask_tokens_out[:, i]) for i in range(self.num_mask_tokens)], dim=1)
        txt = self.txt_align_upscaled_embedding(text_embedding)                                      # [P, C/8]
        Dp, Hp, Wp, c8 = up.shape[1:]
        M = self.num_mask_tokens
        # masks[p, m] = up[p] . hyper[p, m] + up[p] . txt[p] = up[p] . (hyper[p, m] + txt[p]): one product per prompt against M rows
        y = Fh.hyper_product(up.reshape(P, -1, c8), hyper + txt[:, None])                            # [P, voxels, M]
        return y.transpose(1, 2).reshape(P, M, Dp, Hp, Wp), mask_tokens_out
