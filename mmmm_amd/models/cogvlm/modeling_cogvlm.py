"""CogVLM language model (Vicuna-7B + visual expert) on the MI355X HIP kernels.

Same module / parameter names and the same forward surface as the reference
(/root/reference/mmmm/models/cogvlm/modeling_cogvlm.py), different execution model:

* activations live in ONE packed, expert-sorted row layout `[T, hidden]` (no padding rows):
  rows [0, n_vision) are the tokens get_expert_mask routes to the vision expert, rows [n_vision, T) the
  language-expert tokens; the maps are built on the device by vm_expert_index_build (no host sync).
  Every token-type gated linear (reference :87-98, :243-245, :277-279) is then ONE 2-segment grouped
  MFMA GEMM with the LoRA update accumulated in the same fp32 tile, instead of 2 boolean gathers,
  2 GEMMs (+2 LoRA GEMM pairs) and 2 scatters into a zero tensor.
* attention is the var-len causal flash kernel, reading sequence positions through `row_of_pos`.
* lm_head + `.float()` + weighted CE (reference :701-706, :610-627) is one fused operator that never
  materialises fp32 logits.
"""
from __future__ import annotations

import os

from dataclasses import dataclass

import torch
from torch import nn
from torch.utils.checkpoint import checkpoint

from ... import functional as Fh
from ... import kernels as K
from ...param import NoWeightDecayParameter
from .kv_cache import KVCache
from ..lora import ActivationBudget, Linear, gated_linear, linear_decode
from .configuration_cogvlm import CogVLMConfig
from .visual import EVA2CLIPModel

LANGUAGE_TOKEN_TYPE, VISION_TOKEN_TYPE = 0, 1
CE_IGNORE_INDEX = -100


class RMSNorm(nn.Module):
    def __init__(self, hidden_size: int, eps: float = 1e-6):
        super().__init__()
        self.weight = NoWeightDecayParameter(torch.ones(hidden_size))      # modeling_cogvlm.py:33
        self.variance_epsilon = eps

    def forward(self, x: torch.Tensor, nrows: torch.Tensor | None = None, fork: bool = False):
        """`fork`: -> (normed, x passed through) — the second output is the pre-norm block's residual (functional._RMSNorm)"""
        return Fh.rms_norm(x, self.weight, self.variance_epsilon, nrows, fork)


class MLP(nn.Module):
    def __init__(self, config: CogVLMConfig):
        super().__init__()
        self.gate_proj = Linear(config.hidden_size, config.intermediate_size, bias=False)
        self.up_proj = Linear(config.hidden_size, config.intermediate_size, bias=False)
        self.down_proj = Linear(config.intermediate_size, config.hidden_size, bias=False)


class VisionExpertMLP(nn.Module):
    def __init__(self, config: CogVLMConfig):
        super().__init__()
        self.config = config
        self.language_mlp = MLP(config)
        self.vision_mlp = MLP(config)

    def get_lora_modules(self, prefix: str):
        from ...utils import apply_prefix, get_lora_modules_default
        if self.config.lora_lang:
            return get_lora_modules_default(self, prefix, False)
        return get_lora_modules_default(self.vision_mlp, apply_prefix(prefix, 'vision_mlp'))

    def forward(self, x: torch.Tensor, counts: torch.Tensor, residual: torch.Tensor, decode: bool = False) -> torch.Tensor:
        v, l = self.vision_mlp, self.language_mlp
        if decode:      # a single-token call only ever reaches the language expert (reference :58-70 with L == 1)
            h = Fh.silu_mul(linear_decode(x, l.gate_proj), linear_decode(x, l.up_proj))
            return linear_decode(h, l.down_proj, residual)
        gate = gated_linear(x, v.gate_proj, l.gate_proj, counts)
        up = gated_linear(x, v.up_proj, l.up_proj, counts)
        return gated_linear(Fh.silu_mul(gate, up), v.down_proj, l.down_proj, counts, residual=residual)


class RotaryEmbedding(nn.Module):
    """cos/sin tables exactly as the reference builds them (:156-170): in the dtype of `inv_freq`, which
    follows the module dtype (bf16 under bf16-true), `t = arange(dtype=inv_freq.dtype)` included — the
    bf16 position quirk of SURVEY.md §7 is reproduced by construction. The kernel reads fp32 copies."""

    def __init__(self, dim: int, base: float = 10000.0):
        super().__init__()
        self.dim, self.base = dim, base
        self.register_buffer('inv_freq', 1.0 / (base ** (torch.arange(0, dim, 2) / dim)), persistent=False)
        self._cache: tuple | None = None

    def tables(self, n_pos: int, device) -> tuple[torch.Tensor, torch.Tensor]:
        key = (n_pos, self.inv_freq.dtype, str(device))
        if self._cache is None or self._cache[0] != key:
            inv = self.inv_freq.to(device)
            t = torch.arange(n_pos, device=device, dtype=inv.dtype)
            freqs = torch.einsum('i,j->ij', t, inv)
            emb = torch.cat((freqs, freqs), dim=-1)
            self._cache = (key, emb.cos().float().contiguous(), emb.sin().float().contiguous())
        return self._cache[1], self._cache[2]


@dataclass
class Routing:
    """device-side token routing of one batch (built once per forward, shared by all layers)"""
    counts: torch.Tensor        # int32[4]: n_vision_rows, n_rows, max_seqlen, 0
    row_of_tok: torch.Tensor    # int32[B*L]
    tok_of_row: torch.Tensor    # int32[B*L]
    cu_seqlens: torch.Tensor    # int32[B+1]
    row_of_pos: torch.Tensor    # int32[B*L]
    expert_mask: torch.Tensor   # uint8[B,L]
    row_pos: torch.Tensor       # int32[B*L] position id of each row
    n_rows: torch.Tensor        # int32[1] view of counts[1]
    B: int
    L: int
    n_pos: int                  # rope table length (upper bound, host)
    kv: object = None           # KVCache being filled by this forward (generation path), else None
    kv_lens: torch.Tensor | None = None     # decode step: cached tokens per sample including the new one


class VisionExpertAttention(nn.Module):
    def __init__(self, config: CogVLMConfig, layer_idx: int = 0):
        super().__init__()
        self.config = config
        self.layer_idx = layer_idx
        self.hidden_size = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.head_dim = self.hidden_size // self.num_heads
        self.rotary_emb = RotaryEmbedding(self.head_dim)
        self.vision_expert_query_key_value = Linear(self.hidden_size, self.hidden_size * 3, bias=False)
        self.vision_expert_dense = Linear(self.hidden_size, self.hidden_size, bias=False)
        self.language_expert_query_key_value = Linear(self.hidden_size, self.hidden_size * 3, bias=False)
        self.language_expert_dense = Linear(self.hidden_size, self.hidden_size, bias=False)

    def get_lora_modules(self, prefix: str):
        from ...utils import apply_prefix, get_lora_modules_default
        if self.config.lora_lang:
            return get_lora_modules_default(self, prefix, False)
        return [apply_prefix(prefix, 'vision_expert_query_key_value'), apply_prefix(prefix, 'vision_expert_dense')], []

    def forward(self, x: torch.Tensor, rt: Routing, residual: torch.Tensor) -> torch.Tensor:
        decode = rt.kv_lens is not None
        if decode:
            qkv = linear_decode(x, self.language_expert_query_key_value)
        else:
            qkv = gated_linear(x, self.vision_expert_query_key_value, self.language_expert_query_key_value, rt.counts)
        cos, sin = self.rotary_emb.tables(rt.n_pos, x.device)
        fuse_rope = rt.kv is None and rt.kv_lens is None      # training / plain forward: rotation inside the attention node
        if not fuse_rope:
            qkv = Fh.rope_(qkv, rt.row_pos, cos, sin, self.num_heads, self.head_dim, rt.n_rows)
        if rt.kv is not None:               # generation: the rotated K / V rows of this call join the cache (:253-262)
            rt.kv.append(self.layer_idx, qkv, self.hidden_size, rt.n_rows)
        if rt.kv_lens is not None:          # decode step: one query per sample against the cache (:129-141)
            ctx = K.attn_decode(qkv[:, :self.hidden_size], rt.kv.k[self.layer_idx], rt.kv.v[self.layer_idx], rt.kv_lens,
                                self.num_heads, self.head_dim, self.head_dim ** -0.5, rt.kv.attn_bound())
        else:
            ctx = Fh.attention(qkv, rt.cu_seqlens, rt.L, self.num_heads, self.head_dim, self.head_dim ** -0.5, True,
                               row_of_pos=rt.row_of_pos, total_pos_max=rt.B * rt.L,
                               rope=(rt.row_pos, cos, sin, rt.n_rows) if fuse_rope else None)
        if decode:
            return linear_decode(ctx, self.language_expert_dense, residual)
        return gated_linear(ctx, self.vision_expert_dense, self.language_expert_dense, rt.counts, residual=residual)


class CogVLMDecoderLayer(nn.Module):
    def __init__(self, config: CogVLMConfig, layer_idx: int = 0):
        super().__init__()
        self.self_attn = VisionExpertAttention(config, layer_idx)
        self.mlp = VisionExpertMLP(config)
        self.input_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    def forward(self, x: torch.Tensor, rt: Routing) -> torch.Tensor:
        h, x = self.input_layernorm(x, rt.n_rows, fork=True)
        x = self.self_attn(h, rt, residual=x)
        h, x = self.post_attention_layernorm(x, rt.n_rows, fork=True)
        return self.mlp(h, rt.counts, residual=x, decode=rt.kv_lens is not None)


class PackedHidden:
    """hidden states in the packed layout, exposed in the reference's padded `[B, L, h]` form on demand"""

    def __init__(self, packed: list[torch.Tensor], rt: Routing):
        self._packed, self._rt = packed, rt

    def __len__(self):
        return len(self._packed)

    def packed(self, i: int) -> torch.Tensor:
        return self._packed[i]

    def __getitem__(self, i: int) -> torch.Tensor:
        rt = self._rt
        x = self._packed[i]
        out = Fh.gather_rows(x, rt.row_of_tok, rt.B * rt.L)     # padded rows -> zeros (undefined in the reference)
        return out.view(rt.B, rt.L, x.shape[-1])


class CogVLMModel(nn.Module):
    def __init__(self, config: CogVLMConfig):
        super().__init__()
        self.config = config
        self.padding_idx = config.pad_token_id
        self.vocab_size = config.vocab_size
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size, self.padding_idx)
        self.layers = nn.ModuleList([CogVLMDecoderLayer(config, i) for i in range(config.num_hidden_layers)])
        self.norm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.vision = EVA2CLIPModel(config)
        self.gradient_checkpointing = False

    def get_lora_modules(self, prefix: str):
        """the whole embedding layer is fine-tuned; everything else by the default rule (reference :412-421)"""
        from ...utils import apply_prefix, get_lora_modules_default
        targets, saves = [], [apply_prefix(prefix, 'embed_tokens')]
        for name, child in self.named_children():
            if name == 'embed_tokens':
                continue
            t, s = get_lora_modules_default(child, apply_prefix(prefix, name))
            targets.extend(t)
            saves.extend(s)
        return targets, saves

    def build_routing(self, token_type_ids, attention_mask, position_ids) -> Routing:
        B, L = token_type_ids.shape
        r = K.expert_index_build(token_type_ids, attention_mask)
        pos = position_ids.reshape(-1).to(torch.int32)
        row_pos = pos[r['tok_of_row'].clamp_min(0).long()].contiguous()
        n_pos = int(self.config.max_position_embeddings)
        return Routing(counts=r['counts'], row_of_tok=r['row_of_tok'], tok_of_row=r['tok_of_row'], cu_seqlens=r['cu_seqlens'],
                       row_of_pos=r['row_of_pos'], expert_mask=r['expert_mask'], row_pos=row_pos, n_rows=r['counts'][1:2],
                       B=B, L=L, n_pos=n_pos)

    def forward(self, input_ids, *, image=None, patch_size=None, pool_size=None, token_type_ids=None, attention_mask=None,
                position_ids=None, output_hidden_states: bool = False, kv_cache: KVCache | None = None):
        """kv_cache: an EMPTY cache to fill (prefill of the generation path, `use_cache=True` with no past)"""
        B, L = input_ids.shape
        dev = input_ids.device
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        if position_ids is None:
            position_ids = torch.arange(L, device=dev)[None].expand(B, L)
        rt = self.build_routing(token_type_ids, attention_mask, position_ids)
        if kv_cache is not None:
            assert not torch.is_grad_enabled() or not self.training, 'the KV cache is an inference-only structure'
            # packed rows are expert-sorted, not sequence-ordered: a row's cache slot is (its sample, its rank among the
            # sample's valid tokens)
            tok = rt.tok_of_row.clamp_min(0).long()
            rank = (attention_mask.to(torch.int32).cumsum(1, dtype=torch.int32) - 1).reshape(-1)
            kv_cache.begin(torch.div(tok, L, rounding_mode='floor').to(torch.int32), rank[tok],
                           rt.cu_seqlens[1:] - rt.cu_seqlens[:-1], L)
            rt.kv = kv_cache
        # token ids per packed row; rows that will be overwritten by image features look up nothing (-1)
        ids = input_ids.reshape(-1).to(torch.int32)
        feats = None
        if image is not None:
            assert len(image) == B, f'batch size mismatch: {B} {len(image)}'
            feats = self.vision(image, patch_size, pool_size)
            tok = torch.arange(B * L, device=dev).view(B, L)
            img_tok = torch.cat([tok[i, 1:1 + f.shape[0]] for i, f in enumerate(feats)])
            ids = ids.index_fill(0, img_tok, -1)          # (ids[img_tok] = -1 uploads the scalar: a blocking H2D copy)
        row_ids = torch.where(rt.tok_of_row >= 0, ids[rt.tok_of_row.clamp_min(0).long()], torch.full_like(ids, -1))
        x = Fh.embedding_rows(self.embed_tokens.weight, row_ids.contiguous())
        if feats is not None:
            x = Fh.overwrite_rows_(x, torch.cat(feats, dim=0).to(x.dtype), rt.row_of_tok[img_tok].contiguous())
        hs = [] if output_hidden_states else None
        ckpt = self.gradient_checkpointing and self.training and torch.is_grad_enabled()
        n_keep = 0
        if ckpt:
            # saved per packed row and layer: ~8 hidden-width tensors, gate/up/silu·up (3 x intermediate), LoRA
            # projections (7 x r); bf16
            h, im = self.config.hidden_size, self.config.intermediate_size
            n_keep = ActivationBudget.claim(len(self.layers), x.shape[0] * (8 * h + 3 * im + 7 * 64) * x.element_size())
        for i, layer in enumerate(self.layers):
            if hs is not None:
                hs.append(x)
            if ckpt and i >= n_keep:
                x = checkpoint(layer, x, rt, use_reentrant=False, preserve_rng_state=False)
            else:
                x = layer(x, rt)
        x = self.norm(x, rt.n_rows)
        if hs is not None:
            hs.append(x)
        if kv_cache is not None:
            kv_cache.commit()
        return x, rt, (PackedHidden(hs, rt) if hs is not None else None)

    @torch.no_grad()
    def decode_step(self, input_ids: torch.Tensor, position_ids: torch.Tensor, kv_cache: KVCache) -> torch.Tensor:
        """one new token per sample against the cache: input_ids, position_ids [B] (device). A single-token call never
        touches the vision expert (get_expert_mask with L == 1, reference :58-70) and skips the padding-masked norms
        (:306-309), so every row is a language row. No host synchronisation. -> final hidden [B, h]"""
        B = input_ids.shape[0]
        c = kv_cache
        assert B == c.batch
        c.begin(c.rows, c.zeros, c.ones, 1)
        rt = Routing(counts=c.decode_counts, row_of_tok=c.rows, tok_of_row=c.rows, cu_seqlens=c.cu_rows, row_of_pos=c.rows,
                     expert_mask=None, row_pos=position_ids.to(torch.int32).contiguous(), n_rows=c.decode_counts[1:2],
                     B=B, L=1, n_pos=int(self.config.max_position_embeddings), kv=c, kv_lens=c.lens_after())
        x = Fh.embedding_rows(self.embed_tokens.weight, input_ids.to(torch.int32).contiguous())
        for layer in self.layers:
            x = layer(x, rt)
        x = self.norm(x, rt.n_rows)
        kv_cache.commit()
        return x


LM_HEAD_LABEL_ROWS = True


class _LMHeadCE(torch.autograd.Function):
    """lm_head GEMM -> bf16 logits -> fp32 weighted CE, without an fp32 [T, vocab] round trip
    (reference :701 `self.lm_head(h).float()` then _sample_weighted_ce :610-627).

    Only rows that carry a label enter the loss (`ignore_index` -100 everywhere else: the image tokens, the prompt, the padding — 44 % of
    the benchmark's rows), so only those rows go through the [rows x 32 064 x 4096] products: the rows are compacted (labelled ones
    first, device-side — no host sync, the count stays on the device and the GEMMs skip the tiles behind it), the three GEMMs and the
    two CE passes run on the compact rows, and the hidden-state gradient is scattered back (zero rows for the unlabelled ones, which is
    what they get from the full-size product too). The per-row CE and the loss are assembled in the ORIGINAL row order: same bits as
    the full-size form."""

    @staticmethod
    def forward(ctx, h, W, labels, weight, nrows):
        V, Kd = W.shape
        Vp = (V + 63) // 64 * 64
        M = h.shape[0]
        valid = labels >= 0
        n_valid_i = valid.sum()
        compact = LM_HEAD_LABEL_ROWS and M >= 512
        if compact:
            pos = torch.cumsum(valid, 0) - 1                                         # row -> compact position (labelled rows)
            # compact position -> row: labelled rows first, in order; the others fill the tail from the end (a permutation without a sort)
            dest = torch.where(valid, pos, M - torch.cumsum(~valid, 0))
            perm = torch.empty(M, dtype=torch.int64, device=h.device).scatter_(0, dest, torch.arange(M, device=h.device)).to(torch.int32)
            inv = torch.where(valid, pos, torch.full_like(pos, -1)).to(torch.int32)
            n_lab = n_valid_i.to(torch.int32).reshape(1)
            cnt = torch.cat([n_lab, n_lab])                                          # {split, rows}: one segment of n_lab rows
            hc = K.gather_rows(h, perm, M, nrows=n_lab)
            lab_c = labels[perm.long()].contiguous()
        else:
            perm = inv = cnt = None
            n_lab, hc, lab_c = nrows, h, labels
        logits = torch.empty(M, Vp, dtype=h.dtype, device=h.device)
        if Vp > V:
            logits[:, V:].zero_()
        if compact:
            K.gemm(hc, W, w1=W, counts=cnt, out=logits[:, :V])
        else:
            K.gemm(hc, W, out=logits[:, :V])
        row_ce_c, lse = K.ce_fwd(logits, lab_c, V, n_lab)
        row_ce = torch.where(valid, row_ce_c[pos.clamp_min(0)], torch.zeros((), device=h.device)) if compact else row_ce_c
        n_valid = n_valid_i.clamp_min(1).to(torch.float32)
        w = torch.where(valid, weight.to(torch.float32), torch.zeros((), device=h.device))
        loss = torch.dot(row_ce, w) / n_valid
        scale = w / n_valid
        ctx.compact = compact
        ctx.save_for_backward(hc, W, lab_c, lse, scale[perm.long()].contiguous() if compact else scale, n_lab, logits, cnt, inv)
        ctx.mark_non_differentiable(row_ce)
        return loss, row_ce

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dloss, _):
        hc, W, lab_c, lse, scale, n_lab, logits, cnt, inv = ctx.saved_tensors
        V = W.shape[0]
        d = K.ce_bwd(logits, lab_c, lse, scale * dloss.to(torch.float32), V, n_lab, out=logits)   # in place, pad cols = 0
        dh = dW = None
        if ctx.needs_input_grad[0]:
            Wt = K.transpose(W.detach(), pad_to=64)                                                # K = padded vocab
            if ctx.compact:
                dh = K.gather_rows(K.gemm(d, Wt, w1=Wt, counts=cnt), inv)                          # unlabelled rows: zeros
            else:
                dh = K.gemm(d, Wt)
        if ctx.needs_input_grad[1]:
            if d.dtype == torch.bfloat16:
                dW = K.gemm_tn(d[:, :V], hc, nrows=n_lab)
            else:
                # fp32 towers: K-contiguous transposes (zero beyond the labelled rows) feed the NT kernel, as the fp32 islands' linears do
                dW = K.gemm(K.transpose(d[:, :V], pad_to=64, nrows=n_lab), K.transpose(hc, pad_to=64, nrows=n_lab))
        return dh, dW, None, None, None


@dataclass
class CausalLMOutputWithPast:
    """the fields of transformers' CausalLMOutputWithPast that the reference's callers read"""
    loss: torch.Tensor | None = None
    logits: torch.Tensor | None = None
    past_key_values: object = None
    hidden_states: PackedHidden | None = None
    attentions: object = None
    # extras of this implementation (packed layout)
    row_ce: torch.Tensor | None = None
    row_labels: torch.Tensor | None = None
    routing: Routing | None = None
    last_hidden_packed: torch.Tensor | None = None


class CogVLMForCausalLM(nn.Module):
    def __init__(self, config: CogVLMConfig):
        super().__init__()
        self.config = config
        self.model = CogVLMModel(config)
        self.vocab_size = config.vocab_size
        self.lm_head = Linear(config.hidden_size, config.vocab_size, bias=False)

    default_max_new_tokens = 256

    def new_kv_cache(self, batch: int, max_len: int, device) -> KVCache:
        cfg = self.config
        return KVCache(cfg.num_hidden_layers, batch, max_len, cfg.hidden_size, device, dtype=self.lm_head.weight.dtype)

    def gradient_checkpointing_enable(self, kwargs: dict | None = None):
        self.model.gradient_checkpointing = True
        self.model.vision.transformer.gradient_checkpointing = True

    def forward(self, input_ids=None, *, image=None, patch_size=None, pool_size=None, token_type_ids=None,
                attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None, use_cache=None,
                output_attentions=None, output_hidden_states=None, return_dict=None, labels=None, weight=None,
                materialize_logits: bool | None = None) -> CausalLMOutputWithPast:
        if inputs_embeds is not None:
            raise NotImplementedError('inputs_embeds is not part of the reference\'s own call sites of this path')
        if past_key_values is not None and len(past_key_values) > 0:
            # decode step of the generation path (reference :441-442, 253-262): the last column is the new token
            assert labels is None and input_ids.shape[1] >= 1
            x = self.model.decode_step(input_ids[:, -1], position_ids[:, -1], past_key_values)
            logits = linear_decode(x, self.lm_head).float()
            return CausalLMOutputWithPast(logits=logits[:, None], past_key_values=past_key_values, last_hidden_packed=x)
        cache = None
        if use_cache:
            cache = past_key_values if past_key_values is not None else self.new_kv_cache(
                input_ids.shape[0], input_ids.shape[1] + self.default_max_new_tokens, input_ids.device)
        x, rt, hs = self.model(input_ids, image=image, patch_size=patch_size, pool_size=pool_size, token_type_ids=token_type_ids,
                               attention_mask=attention_mask, position_ids=position_ids,
                               output_hidden_states=bool(output_hidden_states), kv_cache=cache)
        out = CausalLMOutputWithPast(hidden_states=hs, routing=rt, last_hidden_packed=x, past_key_values=cache)
        if labels is not None:
            tok = rt.tok_of_row.clamp_min(0).long()
            row_labels = torch.where(rt.tok_of_row >= 0, labels.reshape(-1)[tok], torch.full_like(tok, CE_IGNORE_INDEX)).contiguous()
            if weight is None:
                row_w = torch.ones(row_labels.shape, device=x.device)
            else:
                row_w = weight.reshape(-1)[tok].float()
            loss, row_ce = _LMHeadCE.apply(x, self.lm_head.weight, row_labels, row_w, rt.n_rows)
            if weight is None:
                pass  # F.cross_entropy mean over valid labels == weighted form with unit weights
            out.loss, out.row_ce = loss, row_ce
            out.row_labels = row_labels
        if materialize_logits or (materialize_logits is None and labels is None):
            lg = K.gemm(x.detach(), self.lm_head.weight.detach())
            full = K.gather_rows(lg, rt.row_of_tok, rt.B * rt.L)
            out.logits = full.view(rt.B, rt.L, -1).float()
        return out
