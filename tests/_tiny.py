"""Shapes of the tiny reference models behind tests/golden (must mirror oracle/make_golden.py)."""
from oracle import vividmed as O

BOP, EOP = 150, 151


def lm_cfg(n_lm=2, n_vit=2) -> O.Cfg:
    return O.Cfg(vocab_size=160, hidden_size=64, intermediate_size=128, num_hidden_layers=n_lm, num_attention_heads=2,
                 rms_norm_eps=1e-6,
                 vision=O.VisionCfg(hidden_size=32, num_heads=2, num_hidden_layers=n_vit, intermediate_size=64, layer_norm_eps=1e-6,
                                    patch_size=(4, 8, 8), pos_embed_shape=(2, 2, 4), in_channels=3))


def sam_cfg(instance: bool) -> O.SamCfg:
    return O.SamCfg(embed_dim=32, num_layers=2, num_heads=2, mlp_dim=128, patch_size=(4, 8, 8), pos_embed_shape=(2, 2, 4),
                    num_instances=6 if instance else 3, instance=instance)


def step_cfg() -> O.StepCfg:
    return O.StepCfg(lm=lm_cfg(), sam=sam_cfg(False), isam=sam_cfg(True),
                     mask_loss=dict(dice_weight=2, focal_weight=2, focal_gamma=2),
                     isam_loss=O.ISamLossCfg(), lm_loss_weight=1.0, bop_token_id=BOP, eop_token_id=EOP)
