"""Patch (un)embedding convolutions of the reference (mmmm/models/resample.py) as GEMMs on the HIP kernels.

`Downsample` = conv3d with stride == kernel: im2col (vm_im2col3d) + MFMA GEMM; the z-kernel is folded by
summation when the requested patch z is smaller than the stored kernel z (resample.py:55-62).
`Upsample` = conv_transpose3d with kernel = stride = 2: one GEMM [voxels, C_in] x [C_in, C_out*k] followed by a
pixel-shuffle view; the z kernel is collapsed by its mean when patch_size_z < 2^(cnt+1) (resample.py:86-94).
"""
from __future__ import annotations

import math

import torch
from torch import nn
import torch.nn.functional as F

from .. import functional as Fh


def resample(x: torch.Tensor, shape, scale: bool = False) -> torch.Tensor:
    """luolib.models.spadop.resample — the source is NOT in /root/reference (empty submodule): identity when the
    spatial shape matches, otherwise linear interpolation (align_corners=False); `scale` keeps the kernel sum.
    Semantics are unpinned; kept in this one place."""
    shape = tuple(int(s) for s in shape)
    nd = len(shape)
    if tuple(x.shape[-nd:]) == shape:
        return x
    mode = {1: 'linear', 2: 'bilinear', 3: 'trilinear'}[nd]
    lead = x.shape[:-nd]
    if nd == 3 and x.is_cuda:
        # the position tables ([C, 8, 32, 32] -> the image's patch grid, three times per step): ATen's kernel walks the channels inside
        # one thread per output POSITION (566 us forward + 539 us backward for 1.4 M outputs); the gather-form HIP kernels take one
        # thread per output voxel over all channels (same index rule and association, deterministic backward)
        y = Fh.upsample_trilinear(x.reshape(-1, *x.shape[-nd:]).float(), shape).reshape(*lead, *shape).to(x.dtype)
    else:
        y = F.interpolate(x.reshape(1, -1, *x.shape[-nd:]).float(), size=shape, mode=mode, align_corners=False)
        y = y.reshape(*lead, *shape).to(x.dtype)
    if scale:
        y = y * (math.prod(x.shape[-nd:]) / math.prod(shape))
    return y


class Downsample(nn.Module):
    """parameters named like nn.Conv3d: weight [out, in, kz, ky, kx], bias [out]"""

    def __init__(self, in_channels: int, out_channels: int, kernel_size, bias: bool = True, inflation: str = 'mean',
                 interpolate_2d: bool = False):
        super().__init__()
        if isinstance(kernel_size, int):
            kernel_size = (kernel_size,) * 3
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, tuple(kernel_size)
        self.inflation, self.interpolate_2d = inflation, interpolate_2d
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, *self.kernel_size))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        nn.init.normal_(self.weight, std=0.02)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        """2-D -> 3-D kernel inflation at load time (reference resample.py:31-53): a `[out, in, ky, kx]` kernel becomes
        `[out, in, kz, ky, kx]` — 'mean': every depth slice = w / kz (a constant volume gives the 2-D response);
        'center': the middle slice (or the two middle slices, halved) carries w."""
        key = f'{prefix}weight'
        w = state_dict.get(key)
        if w is not None and w.ndim + 1 == self.weight.ndim:
            if tuple(w.shape[2:]) != tuple(self.kernel_size[1:]) and self.interpolate_2d:
                w = resample(w, self.kernel_size[1:], scale=True)
            kz = self.kernel_size[0]
            if self.inflation == 'mean':
                w = (w / kz)[:, :, None].expand(-1, -1, kz, -1, -1).contiguous()
            elif self.inflation == 'center':
                vol = w.new_zeros(*w.shape[:2], kz, *w.shape[2:])
                if kz % 2:
                    vol[:, :, kz // 2] = w
                else:
                    vol[:, :, kz // 2 - 1] = w / 2
                    vol[:, :, kz // 2] = w / 2
                w = vol
            else:
                raise ValueError(self.inflation)
            state_dict[key] = w
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def folded_weight(self, patch_z: int) -> torch.Tensor:
        w = self.weight
        kz = self.kernel_size[0]
        if kz != patch_z:
            if kz % patch_z != 0:
                raise NotImplementedError
            w = w.reshape(*w.shape[:2], patch_z, kz // patch_z, *w.shape[3:]).sum(dim=3)
        return w.flatten(1)

    def forward(self, image: torch.Tensor, patch_size) -> tuple[torch.Tensor, tuple]:
        """image [C,D,H,W] -> tokens [n_patch, out] (d,h,w order) and the grid shape"""
        C, D, H, W = image.shape
        pz, py, px = patch_size
        cols = Fh.im2col3d(image.to(self.weight.dtype), (pz, py, px))
        w = self.folded_weight(pz)
        y = Fh.linear(cols, w, b0=self.bias)
        return y, (D // pz, H // py, W // px)


class Upsample(nn.Module):
    """parameters named like nn.ConvTranspose3d(in, out, 2, 2): weight [in, out, 2, 2, 2], bias [out]"""

    def __init__(self, in_channels: int, out_channels: int, bias: bool = True, *, cnt: int):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels, 2, 2, 2))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        self.patch_size_th = 1 << (cnt + 1)
        nn.init.normal_(self.weight, std=0.02)

    def forward(self, x: torch.Tensor, patch_size_z: int) -> torch.Tensor:
        """x [n, d, h, w, C_in] channel-last -> [n, d*kz, 2h, 2w, C_out] channel-last"""
        n, d, h, w, ci = x.shape
        wt = self.weight
        if patch_size_z < self.patch_size_th:
            wt = wt.mean(dim=2, keepdim=True)
        kz = wt.shape[2]
        co = self.out_channels
        # [C_in, C_out, kz, 2, 2] -> GEMM weight [(kz 2 2 C_out), C_in]
        wg = wt.permute(2, 3, 4, 1, 0).reshape(kz * 4 * co, ci)
        bias = self.bias.repeat(kz * 4) if self.bias is not None else None
        y = Fh.linear(x.reshape(-1, ci), wg, b0=bias)                       # [n d h w, kz*2*2*co]
        y = y.view(n, d, h, w, kz, 2, 2, co).permute(0, 1, 4, 2, 5, 3, 6, 7).reshape(n, d * kz, h * 2, w * 2, co)
        return y
