"""Sam / InstanceSam grounding heads and InstanceSamLoss (reference segvol/modeling/sam.py) on the fp32 HIP kernels.

Differences in execution, not in results: the image encoder runs once over the packed batch; the mask decoder works
channel-last; on the GPU the Hungarian matching never leaves the device (every cost matrix of the batch from one launch,
`vm_box_match_cost`; every assignment from one more, `vm_lsap_f32`: SciPy's algorithm, SciPy's result — the reference
synchronises once per target, sam.py:243) and the instance losses of a sample are one launch each way (`vm_instance_loss_*`).
CPU tensors take the reference's route (element-wise torch ops, SciPy on the host)."""
from __future__ import annotations

from dataclasses import dataclass

import torch
from torch import nn
import torch.nn.functional as F

from .... import functional as Fh
from .... import kernels as K
from ...lora import Linear
from ...loss import DiceFocalLoss, sigmoid_focal_loss
from .image_encoder import ImageEncoderViT
from .mask_decoder import MaskDecoder
from .prompt_encoder import PromptEncoder

MATCH_NEGATIVE, MATCH_UNCERTAIN = -1, -2


class Sam(nn.Module):
    def __init__(self, image_encoder: ImageEncoderViT, prompt_encoder: PromptEncoder, mask_decoder: MaskDecoder):
        super().__init__()
        self.image_encoder, self.prompt_encoder, self.mask_decoder = image_encoder, prompt_encoder, mask_decoder

    @property
    def prompt_dim(self):
        return self.prompt_encoder.embed_dim

    @property
    def mask_embed_dim(self):
        return self.mask_decoder.transformer_dim

    @property
    def num_mask_tokens(self):
        return self.mask_decoder.num_mask_tokens

    def _decode_groups(self, tokens, grids, patch_size, text_embedding, need_masks=True):
        """Run the mask decoder once per group of samples that share (grid, patch_size_z) instead of once per sample
        (reference sam.py:76-80 loops samples): prompts of a group are concatenated along the prompt axis, each
        prompt attends to its own image's tokens. Returns per-sample (low-res masks | None, mask-token embeddings)."""
        B = len(tokens)
        groups: dict[tuple, list[int]] = {}
        for i in range(B):
            groups.setdefault((tuple(grids[i]), patch_size[i][0]), []).append(i)
        lows: list = [None] * B
        embs: list = [None] * B
        for (grid, pz), idx in groups.items():
            counts = [text_embedding[i].shape[0] for i in idx]
            text = torch.cat([text_embedding[i] for i in idx], dim=0)
            if text.shape[0] == 0:
                for i in idx:
                    c = self.mask_embed_dim
                    embs[i] = text.new_zeros(0, self.num_mask_tokens, c) + 0 * tokens[i].sum()
                    lows[i] = text.new_zeros(0, self.num_mask_tokens, *(1,) * 3) if need_masks else None
                continue
            src = torch.cat([tokens[i][None].expand(n, -1, -1) for i, n in zip(idx, counts) if n > 0], dim=0)
            sparse, dense = self.prompt_encoder(grid, text_embedding=text)
            low, emb = self.mask_decoder(src, self.prompt_encoder.get_dense_pe(grid), sparse.to(text.dtype), dense, text, pz, grid,
                                         need_masks=need_masks)
            off = 0
            for i, n in zip(idx, counts):
                embs[i] = emb[off:off + n]
                lows[i] = low[off:off + n] if low is not None else None
                off += n
        return lows, embs

    def forward(self, image: list[torch.Tensor], patch_size: list[tuple], text_embedding: list[torch.Tensor]):
        """-> per sample [P, D, H, W] semantic mask logits at image resolution (reference :72-87)"""
        tokens, grids = self.image_encoder(image, patch_size)
        lows, _ = self._decode_groups(tokens, grids, patch_size, text_embedding)
        outs = []
        for i in range(len(image)):
            low = lows[i][:, 0]
            outs.append(Fh.upsample_trilinear(low, image[i].shape[1:]) if low.shape[0] > 0 else low)
        return outs


@dataclass
class InstanceSamOutput:
    masks_logits: list
    masks_logits_low_res: list
    boxes: list
    disc_logit: list


class InstanceSam(Sam):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        c = self.mask_embed_dim
        self.box_head = nn.Sequential(Linear(c, c), nn.ReLU(inplace=True), Linear(c, c), nn.ReLU(inplace=True), Linear(c, 6))
        self.disc_head = nn.Sequential(Linear(c, c), nn.ReLU(inplace=True), Linear(c, 1))

    def forward(self, image, patch_size, text_embedding, need_masks: bool = False):
        """training_step only consumes boxes / disc_logit (mmmm.py:214-219): the mask branch of the instance decoder
        (frozen by _freeze_sam_unused) is skipped unless need_masks=True."""
        tokens, grids = self.image_encoder(image, patch_size)
        lows, embs = self._decode_groups(tokens, grids, patch_size, text_embedding, need_masks=need_masks)
        boxes, discs = [], []
        for emb in embs:
            x = self.box_head[4](Fh.relu(self.box_head[2](Fh.relu(self.box_head[0](emb)))))
            boxes.append(x.float().sigmoid())
            discs.append(self.disc_head[2](Fh.relu(self.disc_head[0](emb[:, 1:].contiguous())))[..., 0])
        masks = [Fh.upsample_trilinear(m, image[i].shape[1:]) if m is not None else None for i, m in enumerate(lows)]
        return InstanceSamOutput(masks, lows, boxes, discs)


def box_cs_to_cc(b: torch.Tensor) -> torch.Tensor:
    c, s = b[..., :3], b[..., 3:]
    return torch.cat([c - s / 2, c + s / 2], -1)


def box_pair_giou(b1: torch.Tensor, b2: torch.Tensor) -> torch.Tensor:
    """monai.data.box_utils.box_pair_giou [external]: fp32, eps-regularised denominators"""
    eps = torch.finfo(torch.float32).eps
    b1, b2 = b1.float(), b2.float()

    def vol(e):     # product of the 3 edge lengths, written out: Tensor.prod's backward inspects zeros on the HOST (a sync)
        return e[..., 0] * e[..., 1] * e[..., 2]

    a1 = vol(b1[..., 3:] - b1[..., :3])
    a2 = vol(b2[..., 3:] - b2[..., :3])
    inter = vol((torch.min(b1[..., 3:], b2[..., 3:]) - torch.max(b1[..., :3], b2[..., :3])).clamp(min=0))
    union = a1 + a2 - inter
    iou = inter / (union + eps)
    enc = vol((torch.max(b1[..., 3:], b2[..., 3:]) - torch.min(b1[..., :3], b2[..., :3])).clamp(min=0))
    return iou - (enc - union) / (enc + eps)


class InstanceSamLoss(nn.Module):
    fused = True        # False: element-wise torch form on the GPU as well (a class constant like DiceFocalLoss.fused: tests set it per instance,
                        # `bench.py --set models.segvol.modeling.sam.InstanceSamLoss.fused=False` for A/B runs)

    def __init__(self, *, mask_loss: DiceFocalLoss | None = None, use_neg_mask: bool, box_l1_weight: float, box_giou_weight: float,
                 disc_weight: float, disc_focal_gamma: float, disc_focal_alpha: float | None = None, match_ce: bool = True):
        super().__init__()
        self.mask_loss = mask_loss
        self.use_neg_mask = use_neg_mask
        self.box_l1_weight, self.box_giou_weight, self.disc_weight = box_l1_weight, box_giou_weight, disc_weight
        self.disc_focal_gamma, self.disc_focal_alpha = disc_focal_gamma, disc_focal_alpha
        self.match_ce = match_ce

    def box_loss(self, input, target, reduce_batch: bool = True, return_dict: bool = False):
        l1 = F.l1_loss(input, target) if reduce_batch else F.l1_loss(input, target, reduction='none').mean(dim=-1)
        giou = box_pair_giou(box_cs_to_cc(input), box_cs_to_cc(target))
        if reduce_batch:
            giou = giou.mean()
        giou = 1 - giou
        total = self.box_l1_weight * l1 + self.box_giou_weight * giou
        return {'l1': l1, 'giou': giou, 'total': total} if return_dict else total

    def disc_loss(self, input, label, reduce_batch: bool = True, return_dict: bool = False, alpha: bool = True):
        if isinstance(label, bool):
            label = (torch.ones_like if label else torch.zeros_like)(input)
        d = sigmoid_focal_loss(input, label, self.disc_focal_gamma, self.disc_focal_alpha if alpha else None)
        if reduce_batch:
            d = d.mean()
        total = self.disc_weight * d
        return {f'focal-{self.disc_focal_gamma:.1f}': d, 'total': total} if return_dict else total

    @torch.no_grad()
    def _match_costs(self, boxes_reg, disc_logit, boxes_label, index_offsets_host: list[tuple[int, int]]):
        """device part of the box-only Hungarian matching of one sample (reference _match_instances :178-250, masks_label
        None, num_uncertain 0): one cost matrix per target -> (costs, metas)"""
        nt, nq = disc_logit.shape
        costs, metas = [], []
        prob = disc_logit.sigmoid()
        for i, (s, e) in enumerate(index_offsets_host):
            npos = e - s
            nneg = max(nq - npos, 0)
            if nq == nneg:
                metas.append(None)
                continue
            if self.match_ce:
                cp, cn = self.disc_weight * (1 - prob[i]), self.disc_weight * prob[i]
            else:
                cp = self.disc_loss(disc_logit[i], True, reduce_batch=False)
                cn = self.disc_loss(disc_logit[i], False, reduce_batch=False)
            disc_cost = torch.cat([cp[:, None].expand(nq, npos), cn[:, None].expand(nq, nneg)], dim=1)
            a = boxes_reg[i][:, None].expand(nq, npos, 6).reshape(-1, 6)
            b = boxes_label[s:e][None].expand(nq, npos, 6).reshape(-1, 6)
            pair = self.box_loss(a, b, reduce_batch=False).reshape(nq, npos)
            cost = torch.cat([pair, disc_cost.new_zeros(nq, nneg)], dim=1) + disc_cost
            metas.append((len(costs), npos, cost.shape[1], s))
            costs.append(cost.float())
        return costs, metas

    @staticmethod
    def _assign(host_costs, metas, nt: int, nq: int) -> torch.Tensor:
        """host part: scipy linear_sum_assignment per target -> int64 [nt, nq] (target index | MATCH_NEGATIVE)"""
        from scipy.optimize import linear_sum_assignment
        match = torch.full((nt, nq), MATCH_NEGATIVE, dtype=torch.int64)
        for i, meta in enumerate(metas):
            if meta is None:
                continue
            j, npos, ncol, off = meta
            row, col = linear_sum_assignment(host_costs[j][:, :ncol])
            m = torch.empty(nq, dtype=torch.int64)
            m[torch.as_tensor(row)] = torch.as_tensor(col)
            m[m >= npos] = MATCH_NEGATIVE
            m[m >= 0] += off
            match[i] = m
        return match          # host tensor: the caller derives index tensors from it without touching the device

    @torch.no_grad()
    def match_samples(self, samples: list[tuple]) -> list[torch.Tensor]:
        """Hungarian matching of SEVERAL samples [(boxes_reg [nt, 1 + nq, 6], disc_logit [nt, nq], boxes_label, index_offsets
        (host))] -> per sample int64 [nt, nq] (index of the matched target box | MATCH_NEGATIVE).
        On the GPU the assignment problems are solved where the costs are (`vm_lsap_f32`: SciPy's algorithm, SciPy's result)
        and the match stays on the device: the step has NO device->host synchronisation. (The reference synchronises once per
        target, sam.py:243.) CPU tensors, or a cost matrix wider than the kernel supports, take the reference's route: SciPy
        on the host, one transfer for all samples."""
        if samples and samples[0][1].is_cuda and self.fused:
            out = self._match_on_device(samples)
            if out is not None:
                return out
        built = []
        for boxes_reg, disc_logit, boxes_label, index_offsets in samples:
            offs = [tuple(x) for x in index_offsets.tolist()]
            built.append(self._match_costs(boxes_reg[:, 1:], disc_logit.float(), boxes_label, offs))
        flat = [c for costs, _ in built for c in costs]
        width = max((c.shape[1] for c in flat), default=0)
        rows = max((c.shape[0] for c in flat), default=0)
        if flat and flat[0].is_cuda and width <= K.LSAP_MAX_COLS:
            return self._assign_on_device(flat, built, samples, rows, width)
        host = None
        if flat:
            host = torch.stack([F.pad(c, (0, width - c.shape[1], 0, rows - c.shape[0])) for c in flat]).cpu().numpy()
        out, k = [], 0
        for (costs, metas), (boxes_reg, disc_logit, _, _) in zip(built, samples):
            nt, nq = disc_logit.shape
            part = [host[k + j][:nq] for j in range(len(costs))] if costs else []
            k += len(costs)
            out.append(self._assign(part, metas, nt, nq))
        return out

    def _match_on_device(self, samples: list[tuple]) -> list[torch.Tensor] | None:
        """every cost matrix of every sample from ONE launch (`vm_box_match_cost`), every assignment from one more
        (`vm_lsap_f32`), the bookkeeping of `_assign` as a few tensor ops; one small metadata upload. None: a matrix is wider than
        the assignment kernel supports (the caller takes the host route)."""
        dev = samples[0][1].device
        nq = samples[0][1].shape[1]
        assert all(s[1].shape[1] == nq for s in samples)
        keep, rows_meta, desc, row0, width = [], [], [], 0, 0
        for boxes_reg, disc_logit, boxes_label, index_offsets in samples:
            reg, logit, label = boxes_reg.float().contiguous(), disc_logit.float().contiguous(), boxes_label.float().contiguous()
            keep.append((reg, logit, label))
            assert reg.shape[1:] == (nq + 1, 6)
            for i, (s, e) in enumerate(tuple(x) for x in index_offsets.tolist()):
                npos = e - s
                if npos > 0:
                    ncol = max(nq, npos)
                    width = max(width, ncol)
                    desc.append((reg.data_ptr() + (i * (nq + 1) + 1) * 24, logit.data_ptr() + i * nq * 4, label.data_ptr() + s * 24,
                                 npos, ncol, nq))
                    rows_meta.append((row0 + i, npos, s, nq, ncol, 0))
            row0 += disc_logit.shape[0]
        if width > K.LSAP_MAX_COLS:
            return None
        match = torch.full((row0, nq), MATCH_NEGATIVE, dtype=torch.int64, device=dev)
        if desc:
            n = len(desc)
            packed = torch.tensor(desc + rows_meta, dtype=torch.int64).pin_memory().to(dev, non_blocking=True)
            cost = K.box_match_cost(packed, n, nq, width, self.box_l1_weight, self.box_giou_weight, self.disc_weight, self.match_ce,
                                    self.disc_focal_gamma, self.disc_focal_alpha)
            meta = packed[n:]
            col = K.lsap(cost, meta[:, 3:5].to(torch.int32).contiguous(), width)[:, :nq].long()
            npos, off = meta[:, 1:2], meta[:, 2:3]
            match.index_copy_(0, meta[:, 0], torch.where(col >= npos, torch.full_like(col, MATCH_NEGATIVE), col + off))
        out, row0 = [], 0
        for _, disc_logit, _, _ in samples:
            out.append(match[row0:row0 + disc_logit.shape[0]])
            row0 += disc_logit.shape[0]
        return out

    @staticmethod
    def _assign_on_device(flat, built, samples, rows: int, width: int) -> list[torch.Tensor]:
        """device counterpart of `_assign`: all problems of all samples in one `vm_lsap_f32` launch; the bookkeeping of
        `_assign` (columns past the real targets -> MATCH_NEGATIVE, real ones shifted by the target's box offset) as a few
        tensor ops driven by ONE small metadata upload"""
        dev = flat[0].device
        cost = torch.stack([F.pad(c, (0, width - c.shape[1], 0, rows - c.shape[0])) for c in flat])
        meta, dims, row0 = [], [], 0
        for (costs, metas), (_, disc_logit, _, _) in zip(built, samples):
            nt, nq = disc_logit.shape
            for i, m in enumerate(metas):
                if m is not None:
                    _, npos, ncol, off = m
                    meta.append((row0 + i, npos, off))
                    dims.append((nq, ncol))
            row0 += nt
        nq = samples[0][1].shape[1]
        assert all(s[1].shape[1] == nq for s in samples)
        packed = torch.tensor([(*a, *b) for a, b in zip(meta, dims)], dtype=torch.int64).pin_memory().to(dev, non_blocking=True)
        col = K.lsap(cost, packed[:, 3:5].to(torch.int32).contiguous(), width)[:, :nq].long()
        npos, off = packed[:, 1:2], packed[:, 2:3]
        m = torch.where(col >= npos, torch.full_like(col, MATCH_NEGATIVE), col + off)
        match = torch.full((row0, nq), MATCH_NEGATIVE, dtype=torch.int64, device=dev)
        match.index_copy_(0, packed[:, 0], m)
        out, row0 = [], 0
        for _, disc_logit, _, _ in samples:
            out.append(match[row0:row0 + disc_logit.shape[0]])
            row0 += disc_logit.shape[0]
        return out

    def _match_all(self, boxes_reg_full, disc_logit, boxes_label, index_offsets):
        return self.match_samples([(boxes_reg_full, disc_logit, boxes_label, index_offsets)])[0]

    _weights: dict = {}

    def _compute_loss_fused(self, boxes_reg, disc_logit, boxes_label, index_offsets, match):
        """the device branch of `compute_loss` as ONE launch each way (`vm_instance_loss_fwd / _bwd`): same terms, same log keys.
        Which log entries exist depends on how many entries are matched, which is host-known (see compute_loss)."""
        nt, nq = disc_logit.shape
        n_pos = sum(min(e - s, nq) for s, e in (tuple(x) for x in index_offsets.tolist()) if e > s)
        out = Fh.instance_loss(disc_logit.float().contiguous(), boxes_reg.float().contiguous(), boxes_label.float().contiguous(),
                               match.contiguous(), self.disc_focal_gamma, self.disc_focal_alpha)
        key = (out.device, self.disc_weight, self.box_l1_weight, self.box_giou_weight)
        w = self._weights.get(key)
        if w is None:
            w = self._weights[key] = torch.tensor([self.disc_weight, 0., 0., self.box_l1_weight, self.box_giou_weight, 0.], device=out.device)
        loss = torch.dot(out, w)
        v = out.detach()
        g = f'focal-{self.disc_focal_gamma:.1f}'
        log = {f'instance-disc-{g}': v[0]}
        if n_pos > 0:
            log.update({f'instance-disc-pos-{g}': v[1], 'instance-box-l1': v[3], 'instance-box-giou': v[4]})
        if n_pos < nt * nq:
            log[f'instance-disc-neg-{g}'] = v[2]
        return loss, log

    def compute_loss(self, masks_logits, masks_logits_ds, boxes_reg, disc_logit, masks_label, boxes_label, index_offsets, match=None):
        """reference :252-361, branch used by the training step (no instance masks)"""
        if masks_label is not None:
            raise NotImplementedError('instance segmentation labels are not supported yet (reference mmmm.py:239-241)')
        nt = disc_logit.shape[0]
        assert nt == index_offsets.shape[0]
        loss = 0 * disc_logit.sum()
        log = {}
        if nt > 0:
            if match is None:
                match = self._match_all(boxes_reg, disc_logit, boxes_label, index_offsets)      # int64 [nt, nq]
            if match.is_cuda and self.fused:
                return self._compute_loss_fused(boxes_reg, disc_logit, boxes_label, index_offsets, match)
            boxes_reg = boxes_reg[:, 1:]
            disc_logit = disc_logit.float()
            dev = disc_logit.device
            flat = match.flatten()
            if match.is_cuda:
                # the assignment lives on the device and nothing may synchronise: which entries are positive is data, HOW
                # MANY is not — a target with n > 0 boxes always matches min(n, nq) queries (the dummy negative columns only
                # fill the square) — so the index tensors have host-known sizes. A stable sort keeps the entry order that
                # boolean-mask indexing would give, so every mean below sums in the reference's order.
                nq = disc_logit.shape[1]
                n_pos = sum(min(e - s, nq) for s, e in (tuple(x) for x in index_offsets.tolist()) if e > s)
                is_pos = flat >= 0
                order = torch.argsort((~is_pos).to(torch.uint8), stable=True)
                pos_i, neg_i, cert_i = order[:n_pos], order[n_pos:], None
                up = lambda t: t
                cert_label = is_pos
            else:
                # host assignment (CPU tensors): turn its masks into index tensors on the host
                up = lambda t: t.pin_memory().to(dev, non_blocking=True) if dev.type == 'cuda' else t
                pos_i, neg_i, cert_i = (m.nonzero().flatten() for m in (flat >= 0, flat == MATCH_NEGATIVE, flat != MATCH_UNCERTAIN))
                n_pos = pos_i.numel()
                cert_label = flat[cert_i] >= 0
            dflat = disc_logit.flatten()
            # (no uncertain entries exist in this branch: on the device every entry is "certain")
            d = self.disc_loss(dflat if cert_i is None else dflat.index_select(0, up(cert_i)), up(cert_label), return_dict=True)
            loss = loss + d.pop('total')
            log.update({f'instance-disc-{k}': v for k, v in d.items()})
            if n_pos > 0:
                pos_d = up(pos_i)
                with torch.no_grad():
                    d = self.disc_loss(dflat.index_select(0, pos_d), True, return_dict=True, alpha=False)
                    d.pop('total')
                    log.update({f'instance-disc-pos-{k}': v for k, v in d.items()})
                d = self.box_loss(boxes_reg.reshape(-1, boxes_reg.shape[-1]).index_select(0, pos_d),
                                  boxes_label.index_select(0, up(flat.index_select(0, pos_i) if match.is_cuda else flat[pos_i])), return_dict=True)
                loss = loss + d.pop('total')
                log.update({f'instance-box-{k}': v for k, v in d.items()})
            if n_pos < flat.numel():
                with torch.no_grad():
                    d = self.disc_loss(dflat.index_select(0, up(neg_i)), False, return_dict=True, alpha=False)
                    d.pop('total')
                    log.update({f'instance-disc-neg-{k}': v for k, v in d.items()})
        return loss, log
