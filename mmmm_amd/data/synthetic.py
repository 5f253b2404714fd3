"""Synthetic batches with the exact layout `prepare_vlm_inputs` produces (reference mmmm/data/utils.py:39-145):
[bos] [boi, Np patches, eoi] [<grd>] text..., token types, explicit position ids (image patches share position 2;
the token after <p> and the </p> token do not advance the position — utils.py:20-29), next-token labels with the
prefix ignored, weight 1 (bop_weight at <p> targets). Used by tests, smoke() and bench.py."""
from __future__ import annotations

from dataclasses import dataclass
import math

import torch

from .defs import CE_IGNORE_INDEX


@dataclass
class SpecialTokens:
    """ids of the 8 MMMM tokens appended after the 32000 Vicuna tokens (mmmm/tokenizer.py:36-44 order:
    <sys> <usr> <grd> <ngrd> <p> </p> <np> </np>)"""
    base_vocab: int = 32000

    @property
    def grd_token_id(self): return self.base_vocab + 2
    @property
    def ngrd_token_id(self): return self.base_vocab + 3
    @property
    def bop_token_id(self): return self.base_vocab + 4
    @property
    def eop_token_id(self): return self.base_vocab + 5
    bos_token_id: int = 1
    eos_token_id: int = 2


def text_position_ids(text: torch.Tensor, bop: int, eop: int, start: int) -> torch.Tensor:
    """get_text_position_ids (utils.py:20-29), vectorised: position does not advance after <p> and at </p>"""
    stay = torch.zeros_like(text, dtype=torch.bool)
    stay[1:] = (text[:-1] == bop) | (text[1:] == eop)
    inc = (~stay).long()
    inc[0] = 0
    return start + inc.cumsum(0)


def num_image_tokens(image_shape, patch, pool) -> int:
    d, h, w = (image_shape[1 + i] // patch[i] for i in range(3))
    return math.prod(s // p for s, p in zip((d, h, w), pool))


def make_sample_inputs(n_img: int, n_text: int, tok: SpecialTokens, g: torch.Generator, *, grounding: bool, n_pairs: int,
                       bop_weight: float = 5.0):
    text = torch.randint(3, tok.base_vocab, (n_text,), generator=g)
    pair_pos = []
    if grounding and n_pairs > 0:
        # K non-overlapping "<p> w w </p>" spans
        stride = max(n_text // (n_pairs + 1), 5)
        for k in range(n_pairs):
            s = 2 + k * stride
            if s + 3 < n_text:
                text[s], text[s + 3] = tok.bop_token_id, tok.eop_token_id
                pair_pos.append(s + 3)
    ids = torch.cat([torch.tensor([tok.bos_token_id]), torch.zeros(n_img + 2, dtype=torch.long),
                     torch.tensor([tok.grd_token_id if grounding else tok.ngrd_token_id]), text])
    tt = torch.cat([torch.zeros(1, dtype=torch.long), torch.ones(n_img + 2, dtype=torch.long), torch.zeros(1 + n_text, dtype=torch.long)])
    pos = torch.cat([torch.tensor([0, 1]), torch.full((n_img,), 2), torch.tensor([3, 4]),
                     text_position_ids(text, tok.bop_token_id, tok.eop_token_id, 5)])
    lab_text = torch.cat([text[1:], torch.tensor([tok.eos_token_id])])
    labels = torch.cat([torch.full((1 + n_img + 2 + 1,), CE_IGNORE_INDEX), lab_text])
    w_text = torch.ones(n_text)
    w_text[lab_text == tok.bop_token_id] = bop_weight
    weight = torch.cat([torch.zeros(1 + n_img + 2 + 1), w_text])
    return dict(input_ids=ids, token_type_ids=tt, position_ids=pos, labels=labels, weight=weight), len(pair_pos)


def collate(samples: list[dict], pad_to: int | None = None) -> dict:
    """right-pad with 0 (labels with -100) like _collate_fn (datamodule.py:20-39)"""
    L = max(s['input_ids'].shape[0] for s in samples)
    if pad_to is not None:
        L = max(L, pad_to)
    out = {}
    for k in samples[0]:
        fill = CE_IGNORE_INDEX if k == 'labels' else 0
        out[k] = torch.stack([torch.cat([s[k], torch.full((L - s[k].shape[0],), fill, dtype=s[k].dtype)]) for s in samples])
    out['attention_mask'] = torch.stack([torch.cat([torch.ones(s['input_ids'].shape[0], dtype=torch.long),
                                                    torch.zeros(L - s['input_ids'].shape[0], dtype=torch.long)]) for s in samples])
    return out


def make_batch(image_shapes: list[tuple], patch_sizes: list[tuple], pool_sizes: list[tuple], text_lens: list[int], *,
               tok: SpecialTokens, seed: int = 0, grounding: bool = True, n_pairs: int = 4, instance: list[bool] | None = None,
               image_dtype=torch.bfloat16, device='cpu', boxes_per_target: int = 2) -> dict:
    g = torch.Generator().manual_seed(seed)
    B = len(image_shapes)
    instance = instance or [False] * B
    samples, images, gimages, masks, boxes, offsets = [], [], [], [], [], []
    for i in range(B):
        shp = image_shapes[i]
        n_img = num_image_tokens(shp, patch_sizes[i], pool_sizes[i])
        s, k = make_sample_inputs(n_img, text_lens[i], tok, g, grounding=grounding, n_pairs=n_pairs)
        samples.append(s)
        images.append(torch.randn(*shp, generator=g).to(image_dtype))
        gimages.append(torch.rand(*shp, generator=g))
        if not grounding:
            masks.append(None); boxes.append(None); offsets.append(None)
        elif instance[i]:
            masks.append(None)
            bx = torch.cat([torch.rand(k * boxes_per_target, 3, generator=g) * 0.4 + 0.3, torch.rand(k * boxes_per_target, 3, generator=g) * 0.3 + 0.1], 1)
            boxes.append(bx)
            offsets.append(torch.tensor([[j * boxes_per_target, (j + 1) * boxes_per_target] for j in range(k)], dtype=torch.long).reshape(-1, 2))
        else:
            masks.append(torch.rand(k, *shp[1:], generator=g) < 0.1)
            boxes.append(None); offsets.append(None)
    vi = collate(samples)

    def mv(x):
        if torch.is_tensor(x):
            return x.to(device)
        return x
    return dict(
        vlm_inputs={k: v.to(device) for k, v in vi.items()},
        image=[mv(x) for x in images], grounding_image=[mv(x) for x in gimages],
        patch_size=list(patch_sizes), pool_size=list(pool_sizes),
        masks=[mv(x) for x in masks], boxes=[mv(x) for x in boxes], index_offsets=[mv(x) for x in offsets],
        instance_mask=list(instance), vg_label_mask=[None] * B,
        # the collate output before the move to the device: the step reads token ids / box offsets on the host (models/mmmm.py)
        host=dict(input_ids=vi['input_ids'], labels=vi['labels'], index_offsets=list(offsets)),
    )
