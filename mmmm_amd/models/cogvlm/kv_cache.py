"""KV cache of the generation path.

The reference keeps `past_key_values` as per-layer tuples of `[B, H, L, hd]` tensors that grow by `torch.cat` on every
step (modeling_cogvlm.py:253-262), padding positions included. Here one allocation per model holds every layer:
`k`, `v` `[n_layers, B, max_len, H*hd]` bf16 — token-major rows in the same format as the packed qkv rows of the training
path — plus one int32 length per sample. Only VALID tokens are stored (the packed layout never materialises padding), so a
right-padded prompt batch decodes correctly and the decode attention needs no mask. A step appends one row per sample and
layer with `vm_scatter_rows`; nothing is reallocated, and the only host-side state is an upper bound of the lengths (the
launch geometry), so a decode step never synchronises with the device."""
from __future__ import annotations

import torch

from ... import kernels as K


class KVCache:
    def __init__(self, n_layers: int, batch: int, max_len: int, width: int, device, dtype=torch.bfloat16):
        self.k = torch.empty(n_layers, batch, max_len, width, dtype=dtype, device=device)
        self.v = torch.empty_like(self.k)
        self.lens = torch.zeros(batch, dtype=torch.int32, device=device)       # tokens cached per sample (device)
        self.max_len = max_len
        self.len_bound = 0            # host upper bound of lens (max over samples)
        # launch bound of the decode attention: None = len_bound + 1 (eager); a captured decode step (hipGraph) is replayed
        # with one fixed geometry, so it is sized for the whole cache and the chunks past a sample's length exit at once
        self.launch_bound: int | None = None
        self._slots: torch.Tensor | None = None
        self._pending: torch.Tensor | None = None
        # constants of a decode step (one row per sample), created once: nothing is uploaded inside a step
        self.rows = torch.arange(batch, device=device, dtype=torch.int32)
        self.zeros = torch.zeros(batch, device=device, dtype=torch.int32)
        self.ones = torch.ones(batch, device=device, dtype=torch.int32)
        self.cu_rows = torch.arange(batch + 1, device=device, dtype=torch.int32)
        self.decode_counts = torch.tensor([0, batch, 1, 0], dtype=torch.int32, device=device)     # no vision rows

    @property
    def batch(self) -> int:
        return self.k.shape[1]

    def __len__(self) -> int:         # `if past_key_values:` of the reference's prepare_inputs_for_generation
        return self.k.shape[0] if self.len_bound > 0 else 0

    def begin(self, seq_of_row: torch.Tensor, offset_in_seq: torch.Tensor, rows_per_seq: torch.Tensor, max_new: int):
        """declare the rows of one forward call: row r belongs to sample seq_of_row[r] and is its offset_in_seq[r]-th new
        token (both int32, device); rows_per_seq int32[B]; max_new: host upper bound of rows_per_seq"""
        if self.len_bound + max_new > self.max_len:
            raise ValueError(f'KV cache overflow: {self.len_bound} + {max_new} > {self.max_len}')
        seq = seq_of_row.long()
        self._slots = (seq * self.max_len + self.lens[seq].long() + offset_in_seq.long()).to(torch.int32).contiguous()
        self._pending = rows_per_seq.to(torch.int32)
        self._pending_bound = max_new

    def append(self, layer: int, qkv: torch.Tensor, width: int, nrows: torch.Tensor | None = None):
        """store the (rotated) K and V thirds of the packed qkv rows of this call into layer `layer`"""
        flat_k = self.k[layer].view(-1, width)
        flat_v = self.v[layer].view(-1, width)
        K.scatter_rows(qkv[:, width:2 * width], self._slots, flat_k, nrows=nrows)
        K.scatter_rows(qkv[:, 2 * width:], self._slots, flat_v, nrows=nrows)

    def lens_after(self) -> torch.Tensor:
        return self.lens + self._pending

    def attn_bound(self) -> int:
        return self.launch_bound if self.launch_bound is not None else self.len_bound + 1

    def commit(self):
        """all layers have appended: advance the lengths"""
        self.lens.add_(self._pending)
        self.len_bound += self._pending_bound
        self._slots = self._pending = None
