"""The `Batch` contract the training step consumes (reference mmmm/data/defs.py:54-67, datamodule.py:20-39)."""
from __future__ import annotations

from typing import TypedDict

import torch

CE_IGNORE_INDEX = -100
LANGUAGE_TOKEN_TYPE, VISION_TOKEN_TYPE = 0, 1


class Batch(TypedDict, total=False):
    image: list[torch.Tensor]                 # [3, D, H, W] bf16, normalised
    grounding_image: list[torch.Tensor]       # [3, D, H, W] fp32, un-normalised
    patch_size: list[tuple]                   # (z, y, x)
    pool_size: list[tuple]
    vlm_inputs: dict[str, torch.Tensor]       # input_ids, token_type_ids, position_ids, attention_mask, labels, weight: [B, L]
    masks: list[torch.Tensor | None]          # bool [P, D, H, W]
    boxes: list[torch.Tensor | None]          # f32 [n, 6] CenterSize, normalised
    index_offsets: list[torch.Tensor | None]  # int64 [P, 2]
    instance_mask: list[bool]
    vg_label_mask: list[torch.Tensor | None]
