"""Stand-in for the two `peft` calls on the reference's boundary (scripts/cli.py:77-88; `peft` itself is not in the image):

    lora_config.target_modules, lora_config.modules_to_save = get_lora_modules_default(model)
    peft_model = get_peft_model(model, lora_config)
    model.set_peft_model(peft_model)
    peft_model.load_adapter(str(adapter_path), 'default', is_trainable=...)

`get_peft_model` here adds the rank-r factors IN PLACE to this package's fused `Linear` layers (models/lora.py) — there is no
wrapper module tree, because the LoRA product is part of the GEMM kernel (K-extension), not a second module — and returns a
`PeftModel` handle with the methods the reference calls: `load_adapter`, `save_pretrained`, `base_model`, `peft_config`,
`print_trainable_parameters`. Adapter files use PEFT's key names (models/checkpoint.py) so released `adapter_model.safetensors`
files load. LoRA numerics (`y = Wx + B A drop(x) * alpha / sqrt(r)`, rsLoRA) are restated from PEFT's documentation: parity with the
package itself is unpinned.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from pathlib import Path

from torch import nn

from .models.lora import LoraConfig as _CoreLoraConfig
from .utils import apply_lora


@dataclass
class LoraConfig(_CoreLoraConfig):
    """the fields of `peft.LoraConfig` the reference sets (conf/lora.yaml + scripts/cli.py:80)"""
    target_modules: list[str] | None = None
    modules_to_save: list[str] | None = None
    bias: str = 'none'
    peft_type: str = 'LORA'
    extra: dict = field(default_factory=dict)


class PeftModel:
    """handle returned by `get_peft_model`; `base_model.model` is the adapted model itself (PEFT's attribute path)"""

    def __init__(self, model: nn.Module, config: LoraConfig):
        self.model = model
        self.peft_config = {'default': config}
        self.active_adapter = 'default'

    @property
    def base_model(self):
        return self

    def load_adapter(self, model_id, adapter_name: str = 'default', is_trainable: bool = False, strict: bool = True):
        """reference call sites: scripts/cli.py:87, mmmm.py:154-155. The 'default' adapter always exists by the time either runs
        (`get_peft_model` registered it, cli.py:82-84), and for an EXISTING adapter PEFT's `load_adapter` only loads the weights —
        its `inference_mode` freeze applies to an adapter name it has to create. So `requires_grad` is left exactly as it was
        (LoRA factors, `modules_to_save` copies and the unfrozen heads stay trainable through `load_default_adapter`, whose
        `is_trainable` defaults to False); `is_trainable=False` only puts the model in eval mode, as PEFT does."""
        if adapter_name != 'default':
            raise NotImplementedError('one adapter ("default") per model on this path')
        from .models.checkpoint import load_adapter
        result = load_adapter(self.model, Path(model_id), strict=strict)
        if not is_trainable:
            self.model.eval()
        return result

    def save_pretrained(self, save_directory, **kwargs):
        from .models.checkpoint import save_adapter
        save_adapter(self.model, save_directory, self.peft_config['default'])

    def get_nb_trainable_parameters(self) -> tuple[int, int]:
        ps = list(self.model.parameters())
        return sum(p.numel() for p in ps if p.requires_grad), sum(p.numel() for p in ps)

    def print_trainable_parameters(self):
        t, a = self.get_nb_trainable_parameters()
        print(f'trainable params: {t:,d} || all params: {a:,d} || trainable%: {100 * t / max(a, 1):.4f}')

    def __getattr__(self, name):          # (only reached for names this handle does not define: forward to the model like PEFT does)
        return getattr(self.__dict__['model'], name)

    def __call__(self, *args, **kwargs):
        return self.model(*args, **kwargs)


def get_peft_model(model: nn.Module, peft_config: LoraConfig) -> PeftModel:
    """freeze the base model, add adapters to `peft_config.target_modules`, keep `modules_to_save` trainable (utils.apply_lora)"""
    apply_lora(model, peft_config, peft_config.target_modules, peft_config.modules_to_save)
    return PeftModel(model, peft_config)
