"""State-dict adapters against fixture F10 — outputs of the reference's own `_load_from_state_dict`
methods on synthetic checkpoints (oracle/make_golden.py:f10_state_dict_adapters). CPU only: loading touches no kernel."""
from argparse import Namespace
from pathlib import Path

import torch

GOLD = torch.load(Path(__file__).parent / 'golden' / 'f10_adapters.pt', weights_only=False)


def _same(a: dict, b: dict):
    assert a.keys() == b.keys(), (sorted(a), sorted(b))
    for k in a:
        assert a[k].shape == b[k].shape, k
        torch.testing.assert_close(a[k], b[k], rtol=1e-6, atol=1e-7, msg=k)


def test_cogvlm_patch_embedding_from_2d_checkpoint():
    from mmmm_amd.models.cogvlm.visual import PatchEmbedding
    f = GOLD['patch_embedding']
    m = PatchEmbedding(Namespace(**f['cfg']))
    missing, unexpected = m.load_state_dict({k: v.clone() for k, v in f['ckpt'].items()}, strict=False)
    assert not unexpected
    _same({k: v.detach() for k, v in m.state_dict().items()}, f['loaded'])


def test_downsample_center_inflation():
    from mmmm_amd.models.resample import Downsample
    for kz in (3, 4):
        f = GOLD[f'downsample_center_{kz}']
        m = Downsample(2, 5, (kz, 2, 2), inflation='center')
        m.load_state_dict({'weight': f['weight_2d'].clone(), 'bias': torch.zeros(5)})
        torch.testing.assert_close(m.weight.detach(), f['loaded'], rtol=1e-6, atol=1e-7)


def test_sam_patch_embedding_from_segvol_checkpoint():
    from mmmm_amd.models.segvol.modeling.image_encoder import PatchEmbeddingBlock
    f = GOLD['sam_patch_embedding']
    m = PatchEmbeddingBlock(in_channels=3, patch_size=(2, 4, 4), pos_embed_shape=(2, 2, 3), hidden_size=8, num_heads=2,
                            pt_in_channels=1, pt_patch_size=(2, 4, 4), pt_pos_embed_shape=(2, 2, 3))
    missing, unexpected = m.load_state_dict({k: v.clone() for k, v in f['ckpt'].items()}, strict=False)
    assert not unexpected and not missing
    _same({k: v.detach() for k, v in m.state_dict().items()}, f['loaded'])


def test_mask_decoder_layernorm_and_mask_tokens():
    from mmmm_amd.models.segvol.modeling.mask_decoder import MaskDecoder
    from mmmm_amd.models.segvol.modeling.transformer import TwoWayTransformer
    f = GOLD['mask_decoder']
    m = MaskDecoder(transformer_dim=16, transformer=TwoWayTransformer(depth=1, embedding_dim=16, mlp_dim=32, num_heads=2), num_instances=6)
    with torch.no_grad():
        m.mask_tokens.weight.copy_(f['init_tokens'])        # the rows a shorter checkpoint leaves untouched
    m.load_state_dict({k: v.clone() for k, v in f['ckpt'].items()}, strict=False)
    torch.testing.assert_close(m.output_upscaling[1].weight.detach(), f['ln_weight'], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(m.output_upscaling[1].bias.detach(), f['ln_bias'], rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(m.mask_tokens.weight.detach(), f['mask_tokens'], rtol=0, atol=0)
