"""helpers shared by the GPU parity tests: random non-degenerate weights, product -> oracle state/config"""
import math

import torch

from oracle import vividmed as O


def randomize_(module: torch.nn.Module, seed: int):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            leaf = name.rsplit('.', 1)[-1]
            if p.ndim == 1 and leaf == 'weight':
                v = 1 + 0.1 * torch.randn(p.shape, generator=g)
            elif leaf == 'bias':
                v = 0.05 * torch.randn(p.shape, generator=g)
            else:
                fan_in = math.prod(p.shape[1:]) if p.ndim >= 2 else p.shape[0]
                std = 0.3 if (p.ndim < 2 or 'embed' in name or name.endswith(('boi', 'eoi'))) else 1.0 / math.sqrt(fan_in)
                if 'lora_B' in name:
                    std = 0.05
                # keep softmax out of saturation: with saturated attention dq/dk are pure cancellation noise
                # (P one-hot => dS = P*(dP - delta) ~ 0) and a bf16-vs-fp32 comparison of them is meaningless
                if 'query_key_value' in name or name.endswith(('qkv.weight', 'q_proj.weight', 'k_proj.weight')):
                    std *= 0.35
                v = std * torch.randn(p.shape, generator=g)
            p.copy_(v.to(p.dtype))
        for name, b in module.named_buffers():
            if 'positional_encoding_gaussian_matrix' in name:
                b.copy_(torch.randn(b.shape, generator=g).to(b.dtype))


def oracle_state(module: torch.nn.Module) -> dict:
    return {k: v.detach().float().cpu() for k, v in module.state_dict().items()}


def oracle_cfg(config) -> O.Cfg:
    vc = config.vision_config
    return O.Cfg(vocab_size=config.vocab_size, hidden_size=config.hidden_size, intermediate_size=config.intermediate_size,
                 num_hidden_layers=config.num_hidden_layers, num_attention_heads=config.num_attention_heads,
                 rms_norm_eps=config.rms_norm_eps,
                 vision=O.VisionCfg(hidden_size=vc['hidden_size'], num_heads=vc['num_heads'], num_hidden_layers=vc['num_hidden_layers'],
                                    intermediate_size=vc['intermediate_size'], layer_norm_eps=vc['layer_norm_eps'],
                                    patch_size=tuple(vc['patch_size']), pos_embed_shape=tuple(vc['pos_embed_shape']),
                                    in_channels=vc['in_channels']))


def rel(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-20)).item()


def cpu(x):
    if torch.is_tensor(x):
        return x.detach().cpu()
    if isinstance(x, dict):
        return {k: cpu(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(cpu(v) for v in x)
    return x
