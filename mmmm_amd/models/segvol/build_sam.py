"""build_sam / build_instance_sam with the reference's keyword arguments (segvol/build_sam.py:12-27, 79-95)."""
from __future__ import annotations

from pathlib import Path

import torch

from .modeling import ImageEncoderViT, InstanceSam, MaskDecoder, PromptEncoder, Sam, TwoWayTransformer


def _make(cls, *, embed_dim=768, encoder_mlp_ratio=4, encoder_num_layers=12, num_heads=12, dropout_rate=0.0, patch_size,
          pos_embed_shape, num_instances, pt_in_channels=None, pt_patch_size=None, pt_pos_embed_shape=None, checkpoint=None,
          state_dict_key=None, weight_prefix=''):
    model = cls(
        image_encoder=ImageEncoderViT(in_channels=3, pos_embed_shape=tuple(pos_embed_shape), patch_size=patch_size, hidden_size=embed_dim,
                                      mlp_dim=embed_dim * encoder_mlp_ratio, num_layers=encoder_num_layers, num_heads=num_heads,
                                      dropout_rate=dropout_rate, pt_in_channels=pt_in_channels, pt_patch_size=pt_patch_size,
                                      pt_pos_embed_shape=pt_pos_embed_shape),
        prompt_encoder=PromptEncoder(embed_dim=embed_dim),
        mask_decoder=MaskDecoder(num_instances=num_instances, transformer_dim=embed_dim,
                                 transformer=TwoWayTransformer(depth=2, embedding_dim=embed_dim, mlp_dim=2048, num_heads=8)),
    )
    if checkpoint is not None:
        load_checkpoint(model, Path(checkpoint), state_dict_key, weight_prefix)
    return model


def load_checkpoint(model, ckpt_path: Path, state_dict_key, weight_prefix: str):
    """reference build_sam.py:58-77: select `state_dict_key`, strip `weight_prefix`, drop the text encoder"""
    sd = torch.load(ckpt_path, map_location='cpu')
    if state_dict_key is not None:
        sd = sd[state_dict_key]
    sd = {k[len(weight_prefix):]: v for k, v in sd.items()
          if k.startswith(weight_prefix) and not k.startswith(f'{weight_prefix}text_encoder')}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    print(f'load pre-trained checkpoint from {ckpt_path}; missing: {missing}; unexpected: {unexpected}')


def build_sam(*, patch_size, pos_embed_shape, **kwargs) -> Sam:
    return _make(Sam, patch_size=patch_size, pos_embed_shape=pos_embed_shape, num_instances=3, **kwargs)


def build_instance_sam(*, patch_size, pos_embed_shape, num_instances: int, **kwargs) -> InstanceSam:
    return _make(InstanceSam, patch_size=patch_size, pos_embed_shape=pos_embed_shape, num_instances=num_instances, **kwargs)
