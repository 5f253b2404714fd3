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


class LoraMasks:
    """The keep-masks of every LoRA dropout site of the HIP model for ONE forward, in the oracle's row order: the callable of
    `oracle.vividmed.LORA_DROPOUT` (conf/lora.yaml:3 lora_dropout 0.05 — the benchmarked arithmetic).

    The HIP path's mask is a pure function of (seed of the site, flat element index row * K + col) in the layout the kernel sees:
      * ViT-E layers and the GLU adapter: the packed [sum Nv, C] / [sum Np, C] rows in image order — the oracle concatenates (ViT) or
        walks (adapter: one call per image -> a running row offset per site) the images in the same order;
      * decoder linears: the packed expert-sorted rows (vision expert [0, nv), language expert [nv, n)), BOTH experts under the seed of
        the vision module (models/lora.gated_linear); the oracle calls each expert with its tokens in (sample, position) order, so row
        i of such a call is the packed row of the i-th smallest token index of that expert (`Routing.tok_of_row`).
    What is read here is `vm_dropout` itself applied to ones — the kernel the model-level replay tests already pin the fused consumers to
    (tests/test_kernels_gpu.py: lora_down / tn / dgrad masks == vm_dropout's)."""

    def __init__(self, model, vlm_inputs, p: float):
        from mmmm_amd.models.lora import Linear
        self.p = p
        self.sites = {n: mod for n, mod in model.named_modules() if isinstance(mod, Linear) and mod.lora_cfg is not None}
        rt = model.model.build_routing(vlm_inputs['token_type_ids'], vlm_inputs['attention_mask'], vlm_inputs['position_ids'])
        self.nv, self.n = (int(v) for v in rt.counts[:2].tolist())
        tor = rt.tok_of_row[:self.n].long().cpu()
        self.order_v, self.order_l = torch.argsort(tor[:self.nv]), torch.argsort(tor[self.nv:])
        self.dev = rt.tok_of_row.device
        self.offsets: dict = {}
        self.used: set = set()

    def begin(self):
        """call before every oracle forward (the adapter's running row offsets restart)"""
        self.offsets.clear()

    def _keep(self, mod, rows: int, cols: int) -> torch.Tensor:
        from mmmm_amd import kernels as K
        from mmmm_amd.models.lora import StepState
        ones = torch.ones(rows, cols, dtype=torch.bfloat16, device=self.dev)
        return (K.dropout(ones, self.p, StepState.seed_for(mod._site)) != 0).cpu()

    def __call__(self, name: str, x: torch.Tensor):
        gated = name.startswith('model.layers.')
        src = name.replace('language_expert', 'vision_expert').replace('language_mlp', 'vision_mlp') if gated else name
        mod = self.sites.get(src)
        if mod is None or self.p <= 0:
            return None
        cols = x.shape[-1]
        rows = x.numel() // cols
        self.used.add(name)
        if gated:
            full = self._keep(mod, self.n, cols)
            if src != name:
                assert rows == self.n - self.nv, (name, rows, self.n - self.nv)
                m = full[self.nv:][self.order_l]
            else:
                assert rows == self.nv, (name, rows, self.nv)
                m = full[:self.nv][self.order_v]
        else:
            off = self.offsets.get(name, 0)
            self.offsets[name] = off + rows
            m = self._keep(mod, off + rows, cols)[off:]
        return m.reshape(x.shape)


class lora_dropout_on:
    """context: every LoRA linear of `model` drops with probability p, and the oracle drops the same elements"""

    def __init__(self, model, vlm_inputs, p: float = 0.05):
        self.model, self.vi, self.p = model, vlm_inputs, p

    def __enter__(self) -> LoraMasks:
        from mmmm_amd.models.lora import Linear
        self.mods = [m for m in self.model.modules() if isinstance(m, Linear) and m.lora_cfg is not None]
        self.old = [m.lora_cfg.lora_dropout for m in self.mods]
        # the site ids (-> seeds -> which elements drop) are handed out in construction order over the whole process: number them here, so
        # that a test sees the same masks whatever ran before it
        self.old_sites = [m._site for m in self.mods]
        for i, m in enumerate(self.mods):
            m._site = 100000 + i
        for m in self.mods:
            m.lora_cfg.lora_dropout = self.p
        masks = LoraMasks(self.model, self.vi, self.p)
        O.LORA_DROPOUT, O.LORA_DROPOUT_P = masks, self.p
        return masks

    def __exit__(self, *exc):
        for m, v, st in zip(self.mods, self.old, self.old_sites):
            m.lora_cfg.lora_dropout = v
            m._site = st
        O.LORA_DROPOUT, O.LORA_DROPOUT_P = None, 0.0
