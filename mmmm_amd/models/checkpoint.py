"""Checkpoint I/O for the drop-in surface.

* `load_pretrained(model, path)` — base weights of `THUDM/cogvlm-chat-hf` from a LOCAL directory in the Hugging Face layout
  (`model.safetensors.index.json` + shards, a single `model.safetensors`, or `pytorch_model*.bin`). Parameter names are the
  reference's, so tensors are matched by name; what differs in SHAPE is handled where the reference handles
  it: 2-D patch kernels / position tables by the modules' `_load_from_state_dict` adapters (visual.py:37-57,
  resample.py:31-53), and the vocabulary rows added for the 8 special tokens (the reference loads 32000 rows and then calls
  `resize_token_embeddings(len(tokenizer))`, mmmm.py:109: the new rows keep their fresh initialisation).
* `load_adapter` / `save_adapter` — the PEFT on-disk format written by the reference's trainer (`adapter_model.safetensors`
  + `adapter_config.json`; scripts/cli.py:82-88): LoRA factors as `base_model.model.<module>.lora_{A,B}.weight`, fully
  fine-tuned modules (`modules_to_save`) under their plain parameter names. peft is not in the image: the key convention is
  restated from its public save/load code and is UNPINNED like the LoRA formula itself.
No network access: `path` must exist locally.
"""
from __future__ import annotations

import json
from pathlib import Path

import torch
from torch import nn

PEFT_PREFIX = 'base_model.model.'


def _shards(path: Path) -> list[Path]:
    if path.is_file():
        return [path]
    for index in ('model.safetensors.index.json', 'pytorch_model.bin.index.json'):
        if (path / index).exists():
            names = sorted(set(json.loads((path / index).read_text())['weight_map'].values()))
            return [path / n for n in names]
    for single in ('model.safetensors', 'pytorch_model.bin'):
        if (path / single).exists():
            return [path / single]
    raise FileNotFoundError(f'no Hugging Face checkpoint under {path} (no network in this image: pass a local directory)')


def _read(file: Path) -> dict[str, torch.Tensor]:
    if file.suffix == '.safetensors':
        from safetensors.torch import load_file
        return load_file(str(file))
    return torch.load(file, map_location='cpu', weights_only=True)


def _extend_rows(model: nn.Module, sd: dict[str, torch.Tensor]):
    """a checkpoint with FEWER vocabulary rows than the model fills the leading rows (resize_token_embeddings semantics)"""
    own = dict(model.named_parameters())
    for k in [k for k in sd if k in own]:
        v, p = sd[k], own[k]
        if v.ndim == 2 and p.ndim == 2 and v.shape[1] == p.shape[1] and v.shape[0] < p.shape[0] and k.endswith(('embed_tokens.weight', 'lm_head.weight')):
            full = p.detach().clone()
            full[:v.shape[0]] = v.to(full.dtype)
            sd[k] = full


def load_pretrained(model: nn.Module, path, *, verbose: bool = True) -> tuple[list[str], list[str]]:
    """-> (missing, unexpected) over the whole checkpoint. Shard by shard, so a 35 GB checkpoint never sits in host memory
    twice."""
    path = Path(path)
    expected = set(model.state_dict().keys())
    seen: set[str] = set()
    unexpected: list[str] = []
    for shard in _shards(path):
        sd = _read(shard)
        _extend_rows(model, sd)
        before = set(sd.keys())
        res = model.load_state_dict(sd, strict=False)          # the modules' adapters may rename / reshape entries of `sd`
        unexpected += list(res.unexpected_keys)
        seen |= expected - set(res.missing_keys)
        del sd, before
    missing = sorted(expected - seen)
    if verbose:
        print(f'loaded {len(seen)} tensors from {path}; missing {len(missing)}; unexpected {len(unexpected)}')
    return missing, unexpected


# ----------------------------------------------------------------------------- PEFT adapters
def _to_peft_key(name: str) -> str:
    return PEFT_PREFIX + name.replace('.lora_A.default.', '.lora_A.').replace('.lora_B.default.', '.lora_B.')


def _from_peft_key(key: str) -> str:
    k = key[len(PEFT_PREFIX):] if key.startswith(PEFT_PREFIX) else key
    k = k.replace('.modules_to_save.default.', '.').replace('.modules_to_save.', '.').replace('.original_module.', '.')
    for f in ('lora_A', 'lora_B'):
        k = k.replace(f'.{f}.weight', f'.{f}.default.weight')
    return k


def adapter_state_dict(model: nn.Module) -> dict[str, torch.Tensor]:
    """every TRAINABLE tensor under its PEFT name (LoRA factors + fully fine-tuned modules)"""
    return {_to_peft_key(n): p.detach().cpu().contiguous() for n, p in model.named_parameters() if p.requires_grad}


def save_adapter(model: nn.Module, directory, lora_config=None):
    from safetensors.torch import save_file
    directory = Path(directory)
    directory.mkdir(parents=True, exist_ok=True)
    save_file(adapter_state_dict(model), str(directory / 'adapter_model.safetensors'))
    if lora_config is not None:
        targets = sorted({n.rsplit('.lora_A', 1)[0] for n, _ in model.named_parameters() if '.lora_A.' in n})
        saves = sorted({n.rsplit('.', 1)[0] for n, p in model.named_parameters() if p.requires_grad and '.lora_' not in n})
        cfg = dict(peft_type='LORA', r=lora_config.r, lora_alpha=lora_config.lora_alpha, lora_dropout=lora_config.lora_dropout,
                   use_rslora=lora_config.use_rslora, target_modules=targets, modules_to_save=saves, bias='none')
        (directory / 'adapter_config.json').write_text(json.dumps(cfg, indent=1))


def load_adapter(model: nn.Module, directory, *, strict: bool = True) -> tuple[list[str], list[str]]:
    """-> (trainable tensors the adapter did not provide, adapter tensors the model has no place for)"""
    directory = Path(directory)
    file = directory / 'adapter_model.safetensors'
    sd = _read(file if file.exists() else directory / 'adapter_model.bin')
    own = dict(model.named_parameters())
    unexpected, loaded = [], set()
    with torch.no_grad():
        for key, v in sd.items():
            name = _from_peft_key(key)
            p = own.get(name)
            if p is None or p.shape != v.shape:
                unexpected.append(key)
                continue
            p.copy_(v.to(p.dtype))
            loaded.add(name)
    missing = sorted(n for n, p in own.items() if p.requires_grad and n not in loaded)
    if strict and (missing or unexpected):
        raise RuntimeError(f'adapter at {directory}: missing {missing[:5]}… ({len(missing)}), unexpected {unexpected[:5]}… ({len(unexpected)})')
    return missing, unexpected
