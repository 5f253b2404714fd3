"""DiceFocalLoss of the reference (mmmm/models/loss.py) — same constructor and return keys.
On the GPU (fp32 logits, boolean targets — what the training step passes) both terms come from the fused kernels
`vm_dice_focal_fwd / _bwd` (one streaming pass each way); the element-wise torch form below is the CPU / generic-dtype path
and what the fused kernels are tested against."""
from __future__ import annotations


import torch
from torch import nn
import torch.nn.functional as F

_EPS = 1e-8


def sigmoid_focal_loss(x: torch.Tensor, y: torch.Tensor, gamma: float, alpha: float | None = None) -> torch.Tensor:
    """luolib.losses.sigmoid_focal_loss [external]: torchvision formula, reduction='none'"""
    y = y.to(x.dtype)
    p = torch.sigmoid(x)
    ce = F.binary_cross_entropy_with_logits(x, y, reduction='none')
    p_t = p * y + (1 - p) * (1 - y)
    loss = ce * (1 - p_t) ** gamma
    if alpha is not None:
        loss = (alpha * y + (1 - alpha) * (1 - y)) * loss
    return loss


class DiceFocalLoss(nn.Module):
    fused = True        # 0: element-wise torch form on the GPU as well (A/B measurements)

    def __init__(self, dice_weight: float, focal_weight: float, focal_gamma: float, focal_alpha: float | None = None):
        super().__init__()
        assert focal_gamma >= 0
        self.dice_weight, self.focal_weight = dice_weight, focal_weight
        self.focal_gamma, self.focal_alpha = focal_gamma, focal_alpha

    def dice(self, input: torch.Tensor, target: torch.Tensor | None):
        if target is None:
            return input.new_ones(input.shape[:2])
        p = torch.sigmoid(input)
        t = target.to(p.dtype)
        inter = (t * p).flatten(2).sum(-1)
        denom = t.flatten(2).sum(-1) + p.flatten(2).sum(-1)
        return 1.0 - 2.0 * inter / torch.clip(denom, min=_EPS)

    def focal(self, input: torch.Tensor, target: torch.Tensor | None):
        if target is None:
            target = torch.zeros_like(input)
        if self.focal_gamma < _EPS:
            return F.binary_cross_entropy_with_logits(input, target.to(input.dtype), reduction='none')
        return sigmoid_focal_loss(input, target, self.focal_gamma, self.focal_alpha)

    def forward(self, input: torch.Tensor, target: torch.Tensor | None = None, *, reduce_batch: bool = True,
                return_dict: bool = False):
        assert input.ndim == 5
        if target is not None:
            assert input.shape == target.shape
        if self.fused and input.is_cuda and input.dtype == torch.float32 and (target is None or target.dtype in (torch.bool, torch.uint8)):
            # fused HIP path (vm_dice_focal_*): per-(prompt, channel) Dice and focal sums in one pass over the logits
            from .. import functional as Fh
            P, Cn = input.shape[:2]
            n = input[0, 0].numel()
            t = None if target is None else target.reshape(P * Cn, n).contiguous().view(torch.uint8)
            plain_ce = self.focal_gamma < _EPS                  # reference :51-52: plain BCE, no alpha
            d, fsum = Fh.dice_focal(input.reshape(P * Cn, n).contiguous(), t, 0.0 if plain_ce else self.focal_gamma,
                                    None if plain_ce else self.focal_alpha)
            d, fsum = d.view(P, Cn), fsum.view(P, Cn)
            if reduce_batch:
                dice, focal = d.mean(), fsum.sum() / (P * Cn * n)
            else:
                dice, focal = d.mean(1), fsum.sum(1) / (Cn * n)
        else:
            dice, focal = self.dice(input, target), self.focal(input, target)
            if reduce_batch:
                dice, focal = dice.mean(), focal.mean()
            else:
                dice, focal = dice.flatten(1).mean(1), focal.flatten(1).mean(1)
        total = self.dice_weight * dice + self.focal_weight * focal
        if return_dict:
            key = 'ce' if self.focal_gamma < _EPS else f'focal-{self.focal_gamma:.1f}'
            return {'dice': dice, key: focal, 'total': total}
        return total
