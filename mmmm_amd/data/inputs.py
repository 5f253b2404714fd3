"""Host-side construction of `vlm_inputs` — the step immediately BEFORE the training step.

Mirrors reference `mmmm/data/utils.py:20-145` (`get_text_position_ids`, `prepare_vlm_inputs`) and the padding part of
`mmmm/data/datamodule.py:20-39` (`_collate_fn`). Pure integer work on the host; no device code. The tokenizer is any object
with `encode(text, add_special_tokens=False) -> list[int]`, the string attributes `usr_token` / `sys_token` and the ids
`bos_token_id, eos_token_id, grd_token_id, ngrd_token_id, bop_token_id, eop_token_id, bonp_token_id, eonp_token_id`
(reference `mmmm/tokenizer.py:10-44`).

Sequence layout (per sample):  bos | boi image*n eoi | <grd>/<ngrd> | text ...
  token_type_ids : 0 | 1 ... 1 | 0 | 0 ...                      (1 = vision expert)
  position_ids   : 0 | 1 2 ... 2 3 | 4 | 5 ...                  (all image patches share position 2; inside the text a token
                                                                 directly after <p> and every </p> repeat the previous position)
  labels         : -100 over the prefix and over each prompt; the answer tokens shifted by one, closed by eos
"""
from __future__ import annotations

from typing import NamedTuple, Sequence

import torch

from .defs import CE_IGNORE_INDEX, LANGUAGE_TOKEN_TYPE, VISION_TOKEN_TYPE


class ConvTurn(NamedTuple):
    """reference mmmm/data/defs.py:76-78"""
    prompt: str
    response: str


def get_text_position_ids(text_ids: torch.Tensor, tokenizer, start: int) -> torch.Tensor:
    """positions of the text tokens: +1 per token, except that the token following <p> and every </p> keep the position of
    their predecessor (so a phrase's tags do not consume positions). Reference utils.py:20-29, as a prefix sum."""
    n = text_ids.shape[0]
    step = torch.ones(n, dtype=text_ids.dtype)
    step[0] = 0
    if n > 1:
        hold = (text_ids[:-1] == tokenizer.bop_token_id) | (text_ids[1:] == tokenizer.eop_token_id)
        step[1:][hold] = 0
    return start + step.cumsum(0)


def prepare_vlm_inputs(conversation: Sequence[tuple[str, str]], tokenizer, num_image_tokens: int, *, inference: bool,
                       grounding: bool, max_seq_len: int | None = None, bop_weight: float | None = None):
    """-> (inputs dict, display text). `num_image_tokens` counts image patches only (boi / eoi are added here).
    Training (`inference=False`) adds `labels` and `weight`; negative phrases `<np> … </np>` are fed to the model as
    `<p> … </p>` but the model is never asked to PREDICT their opening tag (the label at that place is the token after it),
    while it is asked to close them. `<p>` targets get `bop_weight`. Reference utils.py:39-145."""
    assert len(conversation) > 0
    if not inference and grounding:
        assert bop_weight is not None
    usr, sys_ = tokenizer.usr_token, tokenizer.sys_token
    text = '\n'.join(f'{usr} {q}\n{sys_} {a}' for q, a in conversation)
    pieces, label_pieces = [], []
    for i, (query, answer) in enumerate(conversation):
        prompt_ids = torch.tensor(tokenizer.encode(f'{usr} {query}{sys_}', add_special_tokens=False), dtype=torch.long)
        if inference and i + 1 == len(conversation):
            pieces.append(prompt_ids)               # the model continues from here
            continue
        answer_ids = torch.tensor(tokenizer.encode(answer, add_special_tokens=False), dtype=torch.long)
        pieces += [prompt_ids, answer_ids]
        if not inference:
            # next-token targets of this turn: nothing over the prompt (its last token already predicts the answer's first)
            label_pieces += [torch.full((prompt_ids.shape[0] - 1,), CE_IGNORE_INDEX), answer_ids, torch.tensor([tokenizer.eos_token_id])]
    text_ids = torch.cat(pieces)
    # <np> / </np> are data-side markers only: the model sees <p> / </p>  (the first text token is never one of them)
    tail = text_ids[1:]
    bonp, eonp = tail == tokenizer.bonp_token_id, tail == tokenizer.eonp_token_id
    tail[bonp] = tokenizer.bop_token_id
    tail[eonp] = tokenizer.eop_token_id
    n_img = num_image_tokens + 2
    prefix = 1 + n_img + 1
    grd = tokenizer.grd_token_id if grounding else tokenizer.ngrd_token_id
    inputs = {
        'input_ids': torch.cat([torch.tensor([tokenizer.bos_token_id]), torch.zeros(n_img, dtype=torch.long), torch.tensor([grd]), text_ids]),
        'token_type_ids': torch.cat([torch.tensor([LANGUAGE_TOKEN_TYPE]), torch.full((n_img,), VISION_TOKEN_TYPE),
                                    torch.full((1 + text_ids.shape[0],), LANGUAGE_TOKEN_TYPE)]),
        'position_ids': torch.cat([torch.tensor([0, 1]), torch.full((n_img - 2,), 2), torch.tensor([3, 4]),
                                  get_text_position_ids(text_ids, tokenizer, start=5)]),
    }
    inputs['attention_mask'] = torch.ones(inputs['input_ids'].shape, dtype=torch.long)
    if not inference:
        labels = torch.cat(label_pieces)
        shifted = labels[1:].clone()
        head = labels[:-1]
        head[bonp] = shifted[bonp]                  # negative target: predict the ordinary next token instead of <p>
        head[eonp] = tokenizer.eop_token_id         # ... but do close the phrase
        weight = torch.ones(labels.shape[0], dtype=torch.float)
        weight[:-1][text_ids[1:] == tokenizer.bop_token_id] = bop_weight
        inputs['labels'] = torch.cat([torch.full((prefix,), CE_IGNORE_INDEX), labels])
        inputs['weight'] = torch.cat([torch.zeros(prefix), weight])
    if max_seq_len is not None:
        inputs = {k: v[:max_seq_len] for k, v in inputs.items()}
    return inputs, text


def collate_vlm_inputs(samples: Sequence[dict]) -> dict:
    """right-pad every field to the longest sample: 0 everywhere, -100 for labels (reference datamodule.py:28-37)"""
    out = {}
    L = max(s['input_ids'].shape[0] for s in samples)
    for key in samples[0]:
        fill = CE_IGNORE_INDEX if key == 'labels' else 0
        rows = []
        for s in samples:
            v = s[key]
            rows.append(torch.cat([v, torch.full((L - v.shape[0],), fill, dtype=v.dtype)]))
        out[key] = torch.stack(rows)
    return out
