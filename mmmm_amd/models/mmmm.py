"""MMMMForCausalLM — the VividMed training step on MI355X behind the reference's LightningModule surface.

Mirrors /root/reference/mmmm/models/mmmm.py: `build(...)` keyword arguments, `training_step(batch)`,
`forward(...)`, `visual_grounding(...)`, `get_lora_modules`, `get_fp32_children`, `on_fit_start`,
`on_load_checkpoint`, `MyPrecision`. Lightning itself is not in the image; the class is a plain nn.Module with
the hooks the Trainer calls (log / log_dict record into `self.logged`), so a Lightning shell can subclass it
unchanged.
"""
from __future__ import annotations

import contextlib
import os

from dataclasses import dataclass, field
from typing import Any

import torch
from torch import nn

from .. import functional as Fh
from .. import kernels as K
from ..utils import apply_prefix, get_lora_modules_default, get_lora_modules_finetune_all
from .cogvlm.configuration_cogvlm import CogVLMConfig
from .cogvlm.modeling_cogvlm import CausalLMOutputWithPast, CogVLMForCausalLM
from .lora import ActivationBudget, Linear, LoraTransposes, StepState, linear_decode
from .loss import DiceFocalLoss

__all__ = ['MMMMForCausalLM', 'build', 'VisionArgs', 'MyPrecision']


@dataclass
class VisionArgs:
    pos_embed_shape: tuple
    pt_pos_embed_shape: tuple | None = None
    patch_size: Any = 16


class _HostCopy:
    """device -> pinned host copy started early (non-blocking) and read late: by the time `get()` is called the copy has
    long completed, so the host never waits for the kernels enqueued in between"""

    def __init__(self, t: torch.Tensor):
        if t.is_cuda:
            self.host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            self.host.copy_(t, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record()
        else:
            self.host, self.event = t, None

    def get(self) -> torch.Tensor:
        if self.event is not None:
            self.event.synchronize()
            self.event = None
        return self.host


@dataclass
class VisualGroundingOutput:
    masks_logits: list = field(default_factory=list)
    masks_logits_ds: list = field(default_factory=list)
    boxes: list = field(default_factory=list)
    disc_logit: list = field(default_factory=list)


def zero_loss(*tensors) -> torch.Tensor:
    """luolib.losses.zero_loss: a graph-connected 0 so every parameter receives a (zero) gradient"""
    out = None
    for t in tensors:
        if t is None:
            continue
        z = 0 * t.sum()
        out = z if out is None else out + z
    return out if out is not None else torch.zeros((), device='cuda')


def _add_prefix(d: dict, prefix: str) -> dict:
    if prefix and not prefix.endswith('/'):
        prefix += '/'
    return {f'{prefix}{k}': v for k, v in d.items()}


class _Trainer:
    is_parallel = False


class MMMMForCausalLM(CogVLMForCausalLM):
    def __init__(self, vlm_config: CogVLMConfig, *, vision_override: VisionArgs | None = None):
        if vision_override is not None:
            vc = vlm_config.vision_config
            vc['pos_embed_shape'] = tuple(vision_override.pos_embed_shape)
            if vision_override.pt_pos_embed_shape is not None:
                vc['pt_pos_embed_shape'] = tuple(vision_override.pt_pos_embed_shape)
            p = vision_override.patch_size
            vc['patch_size'] = (p, p, p) if isinstance(p, int) else tuple(p)
        super().__init__(vlm_config)
        self.tokenizer = None
        self.lm_loss_weight = 1.0
        self.sam = None
        self.mask_loss = None
        self.isam_model = None
        self.isam_loss = None
        self.trainer = _Trainer()
        self.logged: dict[str, torch.Tensor] = {}

    # -- construction ---------------------------------------------------------------------------
    @classmethod
    def build(cls, pretrained_model_name_or_path=None, *args, vision_override: VisionArgs, tokenizer=None,
              torch_dtype='auto', freeze_vision: bool = False, lm_loss_weight: float = 1., sam=None, freeze_sam: bool = True,
              mask_loss: DiceFocalLoss | None = None, isam=None, freeze_isam: bool = True, isam_loss=None,
              config: CogVLMConfig | None = None):
        """Same keyword surface as the reference's `build` (mmmm.py:82-135). Base weights are loaded from
        `pretrained_model_name_or_path` when it is a local checkpoint directory (checkpoint loaders: SURVEY §8f N3);
        with None the model is randomly initialised (benchmarks / tests: no weights or network in the image)."""
        if isinstance(vision_override, dict):          # the YAML's mapping (jsonargparse builds the dataclass on the reference)
            vision_override = VisionArgs(**vision_override)
        self = cls(config or CogVLMConfig(), vision_override=vision_override)
        if pretrained_model_name_or_path is not None:
            from .checkpoint import load_pretrained
            load_pretrained(self, pretrained_model_name_or_path)
        self.tokenizer = tokenizer
        self.lm_loss_weight = lm_loss_weight
        self.sam, self.mask_loss, self.isam_model, self.isam_loss = sam, mask_loss, isam, isam_loss
        if sam is not None:
            assert isam is not None
            if freeze_sam:
                sam.requires_grad_(False)
                sam.eval()
            if freeze_isam:
                isam.requires_grad_(False)
                isam.eval()
            self._freeze_sam_unused()
            assert sam.prompt_dim == isam.prompt_dim
            h = self.config.hidden_size
            self.vg_proj = nn.Sequential(Linear(h, h), nn.ReLU(inplace=True), Linear(h, sam.prompt_dim))
            if isam_loss is not None:
                isam_loss.mask_loss = mask_loss
        if freeze_vision:
            self.model.vision.requires_grad_(False)
        self.model.config.lora_lang = not freeze_vision
        return self

    def _freeze_sam_unused(self):
        sam, isam = self.sam, self.isam_model
        for m in (sam.prompt_encoder.point_embeddings, sam.prompt_encoder.not_a_point_embed, sam.prompt_encoder.mask_downscaling,
                  isam.prompt_encoder.point_embeddings, isam.prompt_encoder.not_a_point_embed, isam.prompt_encoder.mask_downscaling,
                  isam.mask_decoder.output_upscaling, isam.mask_decoder.output_hypernetworks_mlps,
                  isam.mask_decoder.txt_align_upscaled_embedding):
            m.requires_grad_(False)

    concurrent_heads: bool = True      # SAM and iSAM on two HIP streams (visual_grounding)

    def get_fp32_children(self) -> list[str]:
        return ['sam', 'isam_model', 'vg_proj']

    def on_load_checkpoint(self, checkpoint: dict):
        checkpoint['state_dict'] = {}
        self.strict_loading = False

    # -- PEFT hooks (luolib.lightning.peft.PeftMixin on the reference: scripts/cli.py:82-88, mmmm.py:154-155) -------------
    def set_peft_model(self, peft_model) -> None:
        """called by the CLI right after `get_peft_model(model, lora_config)` (mmmm_amd.peft or a compatible handle)"""
        object.__setattr__(self, '_peft_model', peft_model)      # (not a submodule: the handle points back at this model)

    @property
    def peft_model(self):
        pm = self.__dict__.get('_peft_model')
        if pm is None:
            raise AttributeError('no PEFT model attached: call set_peft_model(get_peft_model(model, lora_config)) first (scripts/cli.py:82-85)')
        return pm

    def load_default_adapter(self, ckpt_dir):
        """mmmm.py:154-155: the adapter saved next to a training checkpoint (`<ckpt_dir>/adapter`)"""
        from pathlib import Path
        self.peft_model.load_adapter(str(Path(ckpt_dir) / 'adapter'), 'default')

    def on_fit_start(self) -> None:
        self.gradient_checkpointing_enable({'use_reentrant': False})
        self.freeze_python_gc()

    @staticmethod
    def freeze_python_gc() -> None:
        """Move everything alive now (the model: ~20k modules / parameters and their dicts) into the permanent generation of
        Python's cyclic garbage collector. A training step creates ~100k short-lived Python objects, so full collections keep
        being triggered during the step, and each one walks every tracked object — milliseconds of host stall in the middle of
        the launch stream, at positions that shift with any change in allocation pattern (measured: a code change that REMOVED
        GPU work made the step 8 ms slower until the collector was taken out of the picture). After `gc.freeze()` a collection
        only looks at objects created since. Call again after building optimizer / gradient buckets."""
        import gc
        gc.collect()
        gc.freeze()

    def get_lora_modules(self, prefix: str):
        targets, saves = get_lora_modules_default(self.model, apply_prefix(prefix, 'model'))
        for name, child in self.named_children():
            if name == 'model':
                continue
            saves.extend(get_lora_modules_finetune_all(child, apply_prefix(prefix, name)))
        return targets, saves

    # -- logging hooks (Lightning's names) -------------------------------------------------------
    def log(self, name: str, value, **kwargs):
        self.logged[name] = value.detach() if torch.is_tensor(value) else value

    def log_dict(self, d: dict, **kwargs):
        for k, v in d.items():
            self.log(k, v)

    @property
    def device(self):
        return self.lm_head.weight.device

    # -- grounding -------------------------------------------------------------------------------
    def _get_vg_prompts(self, token_ids, hidden_states, prompt_mask, token_ids_host=None):
        """mmmm.py:167-178: hidden states at </p> -> vg_proj (fp32).
        With `token_ids_host` (a host copy of token_ids started early in the step, see `_HostCopy`) the </p> positions
        are found on the host and gathered by index: no boolean-mask indexing, i.e. no device->host synchronisation after
        the language-model forward has been enqueued."""
        eop = self.tokenizer.eop_token_id
        if token_ids_host is not None:
            th = token_ids_host.get()
            mask_h = th == eop
            counts = mask_h.sum(dim=-1).tolist()
            idx = mask_h.flatten().nonzero().flatten().pin_memory().to(hidden_states.device, non_blocking=True)
            x = hidden_states.reshape(-1, hidden_states.shape[-1]).index_select(0, idx)
        else:
            eop_mask = token_ids == eop
            x = hidden_states[eop_mask]
            counts = eop_mask.sum(dim=-1).tolist()
        x = self.vg_proj[2](Fh.relu(self.vg_proj[0](x)))
        prompts = x.split(counts)
        return [p if m is None else p[m] for p, m in zip(prompts, prompt_mask)]

    def visual_grounding(self, token_ids, hidden_states, image, patch_size, prompt_mask, instance_mask, token_ids_host=None):
        """mmmm.py:180-223"""
        if instance_mask is None:
            raise NotImplementedError
        B = len(image)
        vg_prompts = self._get_vg_prompts(token_ids, hidden_states, prompt_mask, token_ids_host)
        masks_logits: list = [None] * B
        boxes: list = [None] * B
        disc_logit: list = [None] * B
        sem = [i for i in range(B) if not instance_mask[i]]
        ins = [i for i in range(B) if instance_mask[i]]
        # SAM (semantic samples) and iSAM (instance samples) are independent fp32 networks whose linears (M ~ 3k rows)
        # each fill barely half of the 256 CUs: run them on two HIP streams. Autograd replays every backward node on the
        # stream of its forward, so the two backward passes overlap as well.
        side = None
        if sem and ins and self.concurrent_heads and vg_prompts[0].is_cuda:
            cur = torch.cuda.current_stream()
            if getattr(self, '_side_stream', None) is None:
                object.__setattr__(self, '_side_stream', torch.cuda.Stream(device=vg_prompts[0].device))
            side = self._side_stream
            side.wait_stream(cur)
        if ins:
            with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
                out_i = self.isam_model([image[i] for i in ins], [patch_size[i] for i in ins], [vg_prompts[i] for i in ins])
        if sem:
            out = self.sam([image[i] for i in sem], [patch_size[i] for i in sem], [vg_prompts[i] for i in sem])
            for i, m in zip(sem, out):
                masks_logits[i] = m
        if ins:
            if side is not None:
                cur.wait_stream(side)
                for t in (*out_i.boxes, *out_i.disc_logit):
                    t.record_stream(cur)
            for i, b, d in zip(ins, out_i.boxes, out_i.disc_logit):
                boxes[i], disc_logit[i] = b, d
        return masks_logits, boxes, disc_logit

    def _compute_vg_loss(self, masks_logits, boxes_reg, disc_logit, masks_label, boxes_label, index_offsets):
        """mmmm.py:225-285"""
        B = len(masks_logits)
        loss_list, log_dict = [], {}
        # Hungarian matching of all instance samples, solved on the device (no device->host synchronisation)
        inst = [i for i in range(B) if boxes_label[i] is not None and masks_label[i] is None and disc_logit[i].shape[0] > 0]
        matches = dict(zip(inst, self.isam_loss.match_samples(
            [(boxes_reg[i], disc_logit[i], boxes_label[i], index_offsets[i]) for i in inst]))) if inst else {}
        for i in range(B):
            if boxes_label[i] is not None:
                if masks_label[i] is not None:
                    raise NotImplementedError('instance segmentation is not supported yet')
                dummy = boxes_reg[i].new_empty((*boxes_reg[i].shape[:2], 0, 0, 0))
                loss_, log_ = self.isam_loss.compute_loss(dummy, dummy, boxes_reg[i], disc_logit[i], None, boxes_label[i], index_offsets[i],
                                                          match=matches.get(i))
                loss_ = loss_ + zero_loss(masks_logits[i])
            elif masks_label[i] is not None and masks_label[i].shape[0] > 0:
                log_ = self.mask_loss(masks_logits[i][:, None], masks_label[i][:, None], return_dict=True)
                loss_ = log_.pop('total') + zero_loss(disc_logit[i], boxes_reg[i])
            else:
                loss_ = zero_loss(masks_logits[i], disc_logit[i], boxes_reg[i])
                log_ = {}
            loss_list.append(loss_)
            for k, v in log_.items():
                log_dict.setdefault(k, []).append(v)
        loss = torch.stack(loss_list).mean()
        if self.trainer.is_parallel:
            # every rank must touch every trainable head so the gradient all-reduce sees identical buckets
            # (reference mmmm.py:263-278); the bucketed all-reduce of mmmm_amd.ddp zero-fills instead, so
            # the dummy forward is only needed when SAM parameters are trainable AND unused on this rank.
            dev = self.device
            if all(m is None for m in masks_logits) and any(p.requires_grad for p in self.sam.parameters()):
                loss = loss + zero_loss(self.sam([torch.zeros(3, 2, 32, 32, device=dev)], [(1, 16, 16)],
                                                 [torch.zeros(1, self.sam.prompt_dim, device=dev)])[0])
            if all(b is None for b in boxes_reg) and any(p.requires_grad for p in self.isam_model.parameters()):
                out = self.isam_model([torch.zeros(3, 2, 32, 32, device=dev)], [(1, 16, 16)],
                                      [torch.zeros(1, self.sam.prompt_dim, device=dev)])
                loss = loss + zero_loss(out.disc_logit[0], out.boxes[0])
        with torch.no_grad():
            log_dict = {k: torch.stack(v).mean() for k, v in log_dict.items()}
        return loss, log_dict

    # -- the step ----------------------------------------------------------------------------------
    def training_step(self, batch: dict, *args, **kwargs):
        """mmmm.py:296-352"""
        StepState.step += 1
        ActivationBudget.reset()
        if getattr(self, '_lora_transposes', None) is None:
            object.__setattr__(self, '_lora_transposes', LoraTransposes(self))
        self._lora_transposes.refresh()          # one launch: K-contiguous LoRA factors for this step's backward
        vlm_inputs = batch['vlm_inputs']
        input_ids = vlm_inputs['input_ids']
        # Small integer tensors the host needs after the forward has been enqueued (token ids to locate </p>, box index
        # offsets). The batch is born on the host (collate); when it still carries those copies (`batch['host']`, see
        # data/synthetic.py) nothing is transferred. Otherwise they are copied out NOW; the copy queues behind everything already
        # on the stream, so reading it later makes the host wait until the GPU has finished the previous step (measured harmless
        # for the step time, 368.1 vs 367.9 ms: the step is GPU-bound and the host re-builds its lead during the forward).
        host = batch.get('host') or {}
        ids_src = host['input_ids'] if host.get('input_ids') is not None else input_ids
        ids_host = _HostCopy(ids_src[:, 1:]) if self.sam is not None else None
        # the reference logs `train/token-lm/{bop,eop}_loss` only when such a target exists (mmmm.py:333-341: `if token_mask.any()`,
        # a device->host synchronisation there); here the labels' host copy answers that without one
        lab_src = host['labels'] if host.get('labels') is not None else vlm_inputs.get('labels')
        labels_host = _HostCopy(lab_src) if (self.sam is not None and lab_src is not None) else None
        offs_host = None
        offs_src = host.get('index_offsets') if host.get('index_offsets') is not None else batch.get('index_offsets')
        if self.sam is not None and offs_src is not None:
            offs_host = [None if o is None else _HostCopy(o) for o in offs_src]
        out: CausalLMOutputWithPast = self(**vlm_inputs, image=batch['image'], patch_size=batch['patch_size'],
                                           pool_size=batch['pool_size'], return_dict=True, output_hidden_states=True)
        if self.sam is None:
            self.log('train/loss', out.loss, sync_dist=True)
            return self.lm_loss_weight * out.loss
        B = input_ids.shape[0]
        masks_logits, boxes, disc_logit = self.visual_grounding(
            input_ids[:, 1:], out.hidden_states[-1][:, :-1].float(), batch['grounding_image'], batch['patch_size'],
            batch.get('vg_label_mask') or [None] * B, batch['instance_mask'], token_ids_host=ids_host,
        )
        vg_loss, vg_log = self._compute_vg_loss(masks_logits, boxes, disc_logit, batch['masks'], batch['boxes'],
                                                batch['index_offsets'] if offs_host is None else [None if o is None else o.get() for o in offs_host])
        loss = out.loss * self.lm_loss_weight + vg_loss
        logs = {'train/loss': loss, 'train/lm_loss': out.loss, 'train/vg_loss': vg_loss, **_add_prefix(vg_log, 'train/vg')}
        # per-token CE on the <p> / </p> targets: the unweighted row CE is already a by-product of the fused lm_head+CE
        with torch.no_grad():
            lab_h = labels_host.get() if labels_host is not None else None
            for name, tid in (('bop', self.tokenizer.bop_token_id), ('eop', self.tokenizer.eop_token_id)):
                if lab_h is not None and not bool((lab_h == tid).any()):
                    continue                      # no such target in this batch: the reference omits the key
                m = out.row_labels == tid
                logs[f'train/token-lm/{name}_loss'] = (out.row_ce * m).sum() / m.sum().clamp_min(1)
        self.log_dict(logs)
        return loss


    # -- generation path --------------------------------------------------------
    def prepare_inputs_for_generation(self, input_ids, *, token_type_ids, position_ids, image=None, past_key_values=None,
                                      attention_mask=None, inputs_embeds=None, patch_size, pool_size, **kwargs):
        """mmmm.py:368-406. With a non-empty cache only the last column is fed, and its position is corrected IN PLACE
        (as the reference does): the token right after <p> and a </p> itself keep the position of their predecessor."""
        if past_key_values:
            keep_position = (input_ids[:, -2] == self.tokenizer.bop_token_id) | (input_ids[:, -1] == self.tokenizer.eop_token_id)
            position_ids[:, -1] -= keep_position.long()
            input_ids = input_ids[:, -1:]
            token_type_ids = token_type_ids[:, -1:]
            position_ids = position_ids[:, -1:]
        if inputs_embeds is not None and past_key_values is None:
            model_inputs = {'inputs_embeds': inputs_embeds}
        else:
            model_inputs = {'input_ids': input_ids}
        model_inputs.update({
            'image': image, 'token_type_ids': token_type_ids, 'position_ids': position_ids, 'past_key_values': past_key_values,
            'attention_mask': attention_mask, 'patch_size': patch_size, 'pool_size': pool_size, 'use_cache': kwargs.get('use_cache'),
        })
        return model_inputs

    @torch.no_grad()
    def generate(self, input_ids, *, token_type_ids, position_ids, image, patch_size, pool_size, attention_mask=None,
                 max_new_tokens: int = 32, eos_token_id: int | None = None, forced_tokens: torch.Tensor | None = None,
                 eos_check_every: int = 16, return_logits: bool = False, use_graph: bool = False) -> GenerateOutput:
        """Greedy decoding (`num_beams=1`, what scripts/demo.py and `evaluate`, mmmm.py:426-452, use): one prefill that fills
        the KV cache, then one token per sample and step. Unlike the reference (whose image scatter assumes column 1 and
        whose HF loop wants left padding, so images force batch size 1) a RIGHT-padded prompt batch is decoded together:
        the cache holds valid tokens only. The <p>/</p> position rule (mmmm.py:354-366, 383-386) runs on the device; the
        loop synchronises with the host only every `eos_check_every` steps to test for early termination.
        `forced_tokens` [B, steps] replaces the arg-max choice (teacher forcing for parity tests). `use_graph` captures one
        decode step (32 layers of ~25 launches + token choice) into a hipGraph and replays it: a step is launch-bound
        otherwise (16.5 ms eager vs ~3 ms of HBM time at batch 1)."""
        was_training = self.training
        self.eval()
        try:
            B, L = input_ids.shape
            dev = input_ids.device
            if attention_mask is None:
                attention_mask = torch.ones_like(input_ids)
            steps = max_new_tokens if forced_tokens is None else forced_tokens.shape[1]
            cache = self.new_kv_cache(B, L + steps, dev)
            out = self(input_ids, image=image, patch_size=patch_size, pool_size=pool_size, token_type_ids=token_type_ids,
                       attention_mask=attention_mask, position_ids=position_ids, past_key_values=cache, use_cache=True,
                       materialize_logits=False)
            # last valid token of every sample: its hidden state predicts the first new token
            am = attention_mask.bool()
            n_valid = am.sum(1)
            last_col = (torch.arange(L, device=dev)[None] * am).argmax(1)
            rt = out.routing
            last_row = rt.row_of_tok.long()[torch.arange(B, device=dev) * L + last_col]
            logits = linear_decode(out.last_hidden_packed[last_row].contiguous(), self.lm_head).float()
            prev_tok = input_ids[torch.arange(B, device=dev), last_col]
            pos = position_ids[torch.arange(B, device=dev), last_col]
            bop, eop = self.tokenizer.bop_token_id, self.tokenizer.eop_token_id
            done = torch.zeros(B, dtype=torch.bool, device=dev)
            head = self.lm_head.weight.detach()

            def choose(logits, prev_tok, pos, done, t):
                """next token (arg-max or forced), eos bookkeeping and the <p>/</p> position rule (mmmm.py:354-366,383-386)"""
                tok = forced_tokens[:, t] if forced_tokens is not None else logits.argmax(-1)
                if eos_token_id is not None:
                    tok = torch.where(done, torch.full_like(tok, eos_token_id), tok)
                    done = done | (tok == eos_token_id)
                return tok, pos + 1 - ((prev_tok == bop) | (tok == eop)).long(), done

            if use_graph and forced_tokens is None and not return_logits and steps > 2:
                new_tokens, new_pos = self._generate_graphed(cache, head, logits, prev_tok, pos, done, steps, eos_token_id, bop, eop,
                                                             eos_check_every)
                return GenerateOutput(new_tokens=new_tokens, new_position_ids=new_pos, prompt_lengths=n_valid, past_key_values=cache)
            new_tokens, new_pos, all_logits = [], [], ([logits] if return_logits else None)
            for t in range(steps):
                tok, pos, done = choose(logits, prev_tok, pos, done, t)
                new_tokens.append(tok)
                new_pos.append(pos)
                if t + 1 == steps:
                    break
                if forced_tokens is None and eos_token_id is not None and (t + 1) % eos_check_every == 0 and bool(done.all()):
                    break
                logits = linear_decode(self.model.decode_step(tok, pos, cache), self.lm_head).float()
                if return_logits:
                    all_logits.append(logits)
                prev_tok = tok
            return GenerateOutput(new_tokens=torch.stack(new_tokens, 1), new_position_ids=torch.stack(new_pos, 1), prompt_lengths=n_valid,
                                  logits=all_logits, past_key_values=cache)
        finally:
            self.train(was_training)


    def _generate_graphed(self, cache, head, logits, prev_tok, pos, done, steps, eos_token_id, bop, eop, eos_check_every):
        """greedy loop with the decode step captured in a hipGraph. State lives in static device buffers that the graph
        updates in place: (tok, pos, done) of the newest token, a step counter, and the [B, steps] outputs it scatters into.
        One eager step first: it sets the lazily-initialised kernel attributes and warms the allocator outside the capture."""
        B = prev_tok.shape[0]
        dev = prev_tok.device
        out_tok = torch.full((B, steps), eos_token_id if eos_token_id is not None else 0, dtype=torch.long, device=dev)
        out_pos = torch.zeros(B, steps, dtype=torch.long, device=dev)

        def choose(logits, prev_tok, pos, done):
            tok = logits.argmax(-1)
            if eos_token_id is not None:
                tok = torch.where(done, torch.full_like(tok, eos_token_id), tok)
                done = done | (tok == eos_token_id)
            return tok, pos + 1 - ((prev_tok == bop) | (tok == eop)).long(), done

        tok, pos, done = choose(logits, prev_tok, pos, done)
        out_tok[:, 0], out_pos[:, 0] = tok, pos
        logits = linear_decode(self.model.decode_step(tok, pos, cache), self.lm_head).float()          # eager step 1
        s_prev, s_pos, s_done = tok.clone(), pos.clone(), done.clone()
        s_logits = logits.clone()
        s_col = torch.ones(B, 1, dtype=torch.long, device=dev)
        cache.launch_bound = cache.max_len

        def step():
            tok, pos, done = choose(s_logits, s_prev, s_pos, s_done)
            out_tok.scatter_(1, s_col, tok[:, None])
            out_pos.scatter_(1, s_col, pos[:, None])
            s_col.add_(1)
            s_logits.copy_(linear_decode(self.model.decode_step(tok, pos, cache), self.lm_head).float())
            s_prev.copy_(tok); s_pos.copy_(pos); s_done.copy_(done)

        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        bound0 = cache.len_bound
        with torch.cuda.graph(graph):
            step()
        cache.len_bound = bound0                  # the capture ran the host bookkeeping once without executing anything
        t = 1
        while t < steps - 1:
            graph.replay()
            cache.len_bound += 1
            t += 1
            if eos_token_id is not None and t % eos_check_every == 0 and bool(s_done.all()):
                break
        if t == steps - 1:                        # the last token needs no decode step after it
            tok, pos, _ = choose(s_logits, s_prev, s_pos, s_done)
            out_tok[:, t], out_pos[:, t] = tok, pos
        cache.launch_bound = None
        return out_tok, out_pos


@dataclass
class GenerateOutput:
    new_tokens: torch.Tensor            # [B, steps] generated (or forced) tokens; eos-filled after a sample has finished
    new_position_ids: torch.Tensor      # [B, steps] their position ids under the <p>/</p> rule
    prompt_lengths: torch.Tensor        # [B] valid prompt tokens per sample
    logits: list | None = None          # per step [B, V] fp32: logits[t] chose new_tokens[:, t]
    past_key_values: object = None

    def sequences(self, input_ids: torch.Tensor, pad_token_id: int = 0) -> torch.Tensor:
        """prompt and continuation in one right-padded [B, L + steps] tensor (HF `generate` layout when nothing is padded)"""
        B, L = input_ids.shape
        steps = self.new_tokens.shape[1]
        seq = torch.full((B, L + steps), pad_token_id, dtype=input_ids.dtype, device=input_ids.device)
        ar = torch.arange(L + steps, device=input_ids.device)[None]
        n = self.prompt_lengths[:, None]
        seq[:, :L] = torch.where(ar[:, :L] < n, input_ids, seq[:, :L])
        idx = (n + torch.arange(steps, device=input_ids.device)[None])
        return seq.scatter(1, idx, self.new_tokens.to(seq.dtype))


def build(*args, **kwargs) -> MMMMForCausalLM:
    return MMMMForCausalLM.build(*args, **kwargs)


class MyPrecision:
    """the reference's precision plugin (mmmm.py:468-492): everything bf16 except the fp32 islands
    `sam`, `isam_model`, `vg_proj` and the inputs `grounding_image`, `boxes`."""
    fp32_input_keys = ('grounding_image', 'boxes')

    def convert_input(self, data: dict) -> dict:
        def cv(x):
            if torch.is_tensor(x):
                return x.to(torch.bfloat16) if x.is_floating_point() else x
            if isinstance(x, dict):
                return {k: cv(v) for k, v in x.items()}
            if isinstance(x, (list, tuple)):
                return type(x)(cv(v) for v in x)
            return x
        out = {k: cv(v) for k, v in data.items() if k not in self.fp32_input_keys}
        out.update({k: data[k] for k in self.fp32_input_keys if data.get(k) is not None})
        return out

    def convert_module(self, module: MMMMForCausalLM) -> MMMMForCausalLM:
        assert isinstance(module, MMMMForCausalLM)
        fp32 = set(module.get_fp32_children())
        for name, child in module.named_children():
            if name not in fp32:
                child.to(torch.bfloat16)
        return module
