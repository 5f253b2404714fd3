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


# ---------------------------------------------------------------------------------------------------------------------------
# true-width fixtures (tests/golden/f2_true_width.pt): weights and inputs are closed-form functions of (name, index), restated
# here from oracle/make_golden.py:spec_tensor (the generator cannot be imported on the GPU box: it needs /root/reference).
def spec_tensor(name: str, shape, scale: float, dtype=None):
    import math
    import zlib
    import torch
    n = math.prod(shape)
    salt = float(zlib.crc32(name.encode()) % 10007)
    i = torch.arange(n, dtype=torch.float64)
    u = torch.sin(i * 12.9898 + salt * 78.233) * 43758.5453
    u = u - torch.floor(u)
    return ((2 * u - 1) * (scale * math.sqrt(3.0))).to(dtype or torch.float32).reshape(shape)


def spec_state(prefix: str, shapes: dict, param_sums: dict | None = None) -> dict:
    """{name: tensor} for parameter `shapes` under the fixture's rules: 1-D `weight` = 1 + spec(0.1), `bias` = spec(0.02), matrices =
    spec(1/sqrt(fan_in)); checked against the checksums the generator stored"""
    import math
    out = {}
    for n, shp in shapes.items():
        leaf = n.rsplit('.', 1)[-1]
        if leaf == 'weight' and len(shp) == 1:
            t = 1 + spec_tensor(prefix + n, shp, 0.1)
        elif leaf == 'bias':
            t = spec_tensor(prefix + n, shp, 0.02)
        else:
            t = spec_tensor(prefix + n, shp, 1.0 / math.sqrt(shp[1]))
        if param_sums is not None:
            assert abs(float(t.double().sum()) - param_sums[n]) <= 1e-6 * max(1.0, abs(param_sums[n])), n
        out[n] = t
    return out


def decoder_layer_shapes(h=4096, i=11008) -> dict:
    s = {'input_layernorm.weight': (h,), 'post_attention_layernorm.weight': (h,)}
    for e in ('vision', 'language'):
        s[f'self_attn.{e}_expert_query_key_value.weight'] = (3 * h, h)
        s[f'self_attn.{e}_expert_dense.weight'] = (h, h)
        s[f'mlp.{e}_mlp.gate_proj.weight'] = (i, h)
        s[f'mlp.{e}_mlp.up_proj.weight'] = (i, h)
        s[f'mlp.{e}_mlp.down_proj.weight'] = (h, i)
    return s


def vit_layer_shapes(d=1792, f=15360) -> dict:
    return {'input_layernorm.weight': (d,), 'input_layernorm.bias': (d,), 'post_attention_layernorm.weight': (d,),
            'post_attention_layernorm.bias': (d,), 'attention.query_key_value.weight': (3 * d, d), 'attention.query_key_value.bias': (3 * d,),
            'attention.dense.weight': (d, d), 'attention.dense.bias': (d,), 'mlp.fc1.weight': (f, d), 'mlp.fc1.bias': (f,),
            'mlp.fc2.weight': (d, f), 'mlp.fc2.bias': (d,)}


def two_way_block_shapes(c=768, mlp=2048, down=2) -> dict:
    s = {}
    for a, ci in (('self_attn', c), ('cross_attn_token_to_image', c // down), ('cross_attn_image_to_token', c // down)):
        for pr in ('q_proj', 'k_proj', 'v_proj'):
            s[f'{a}.{pr}.weight'], s[f'{a}.{pr}.bias'] = (ci, c), (ci,)
        s[f'{a}.out_proj.weight'], s[f'{a}.out_proj.bias'] = (c, ci), (c,)
    for n in ('norm1', 'norm2', 'norm3', 'norm4'):
        s[f'{n}.weight'], s[f'{n}.bias'] = (c,), (c,)
    s['mlp.lin1.weight'], s['mlp.lin1.bias'], s['mlp.lin2.weight'], s['mlp.lin2.bias'] = (mlp, c), (mlp,), (c, mlp), (c,)
    return s
