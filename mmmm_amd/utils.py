"""LoRA target discovery (which modules get adapters, which are fully fine-tuned) — the rule set of the
reference's mmmm/utils.py:19-61, restated for this package's module classes, plus `apply_lora`, the
stand-in for `peft.get_peft_model` (scripts/cli.py:82-85)."""
from __future__ import annotations

from torch import nn

from .models.lora import Linear, LoraConfig


def apply_prefix(prefix: str, path: str) -> str:
    return f'{prefix}{path}' if prefix.endswith('.') or not prefix else f'{prefix}.{path}'


def _trainable_leaf(module: nn.Module) -> bool:
    flags = [p.requires_grad for p in module.parameters()]
    if flags and any(flags):
        assert all(flags), 'a leaf module must be entirely trainable or entirely frozen'
        return True
    return False


def get_lora_modules_default(module: nn.Module, prefix: str = '', recursive: bool = True) -> tuple[list[str], list[str]]:
    """linear / embedding layers -> LoRA targets; other trainable leaves -> modules_to_save; a module that
    defines get_lora_modules decides for its own subtree (mmmm/utils.py:19-43)."""
    targets: list[str] = []
    saves: list[str] = []

    def walk(m: nn.Module, pre: str):
        if recursive and hasattr(m, 'get_lora_modules'):
            t, s = m.get_lora_modules(prefix='')
            targets.extend(apply_prefix(pre, n) for n in t if _trainable_leaf(m.get_submodule(n)))
            saves.extend(apply_prefix(pre, n) for n in s if _trainable_leaf(m.get_submodule(n)))
        elif isinstance(m, (Linear, nn.Linear, nn.Embedding)):
            targets.append(pre)
        else:
            children = list(m.named_children())
            if not children:
                if _trainable_leaf(m):
                    saves.append(pre)
            else:
                for name, child in children:
                    walk(child, apply_prefix(pre, name))

    walk(module, prefix)
    return targets, saves


def get_lora_modules_finetune_all(module: nn.Module, prefix: str) -> list[str]:
    """every trainable leaf module is fully fine-tuned (mmmm/utils.py:45-58)"""
    saves: list[str] = []

    def walk(m: nn.Module, pre: str):
        children = list(m.named_children())
        if not children:
            if _trainable_leaf(m):
                saves.append(pre)
        else:
            for name, child in children:
                walk(child, apply_prefix(pre, name))

    walk(module, prefix)
    return saves


def apply_lora(model: nn.Module, cfg: LoraConfig, target_modules: list[str] | None = None,
               modules_to_save: list[str] | None = None) -> nn.Module:
    """Freeze the base model, add rank-r adapters to `target_modules`, keep `modules_to_save` trainable."""
    if target_modules is None or modules_to_save is None:
        target_modules, modules_to_save = model.get_lora_modules(prefix='')
    for p in model.parameters():
        p.requires_grad_(False)
    for name in target_modules:
        m = model.get_submodule(name)
        if not isinstance(m, Linear):
            raise TypeError(f'LoRA target {name} is {type(m).__name__}; only linear layers carry adapters on this path')
        m.add_lora(cfg)
    for name in modules_to_save:
        for p in model.get_submodule(name).parameters():
            p.requires_grad_(True)
    model.lora_target_modules, model.lora_modules_to_save = list(target_modules), list(modules_to_save)
    return model


def instantiate(spec, *, remap=(('mmmm.', 'mmmm_amd.'),), overrides=None, path: str = ''):
    """Resolve a `{class_path, init_args}` mapping of the reference's YAML configs (conf/phase-*/model.yaml) the way jsonargparse
    does for the LightningCLI: import `class_path` (with the package prefix remapped onto this package), instantiate nested specs
    depth first, call it with `init_args`. `overrides(path, class_path, init_args) -> init_args` lets the caller edit the
    arguments of any node (tests null the checkpoint paths and shrink the widths). Lightning / jsonargparse are not in the image;
    this is the part of them the drop-in needs."""
    import importlib
    if isinstance(spec, dict) and 'class_path' in spec:
        cp = spec['class_path']
        for old, new in remap:
            if cp.startswith(old):
                cp = new + cp[len(old):]
                break
        args = {k: instantiate(v, remap=remap, overrides=overrides, path=f'{path}.{k}' if path else k)
                for k, v in (spec.get('init_args') or {}).items()}
        if overrides is not None:
            args = overrides(path, cp, args)
        mod, _, name = cp.rpartition('.')
        return getattr(importlib.import_module(mod), name)(**args)
    if isinstance(spec, dict):
        return {k: instantiate(v, remap=remap, overrides=overrides, path=f'{path}.{k}' if path else k) for k, v in spec.items()}
    return spec
