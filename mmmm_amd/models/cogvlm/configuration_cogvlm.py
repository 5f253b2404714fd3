"""CogVLMConfig with the reference's field names (mmmm/models/cogvlm/configuration_cogvlm.py:5-45) as a plain
dataclass — no transformers dependency on the hot path. Defaults = THUDM/cogvlm-chat-hf + the 8 MMMM tokens."""
from __future__ import annotations

from dataclasses import dataclass, field


def default_vision_config() -> dict:
    # THUDM/cogvlm-chat-hf config.json (EVA2-CLIP-E) with the MMMM vision_override of conf/model.yaml
    return dict(in_channels=3, hidden_size=1792, num_heads=16, num_hidden_layers=63, intermediate_size=15360,
                hidden_act='gelu', dropout_prob=0.0, layer_norm_eps=1e-6, patch_size=(16, 16, 16),
                pos_embed_shape=(8, 32, 32), pt_pos_embed_shape=(35, 35))


@dataclass
class CogVLMConfig:
    vocab_size: int = 32008          # 32000 + 8 special tokens (mmmm/tokenizer.py:36-44, mmmm.py:109)
    hidden_size: int = 4096
    intermediate_size: int = 11008
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    hidden_act: str = 'silu'
    max_position_embeddings: int = 2048
    initializer_range: float = 0.02
    rms_norm_eps: float = 1e-6
    template_version: str = 'chat'
    pad_token_id: int = 0
    bos_token_id: int = 1
    eos_token_id: int = 2
    tie_word_embeddings: bool = False
    use_cache: bool = True
    vision_config: dict = field(default_factory=default_vision_config)
    lora_lang: bool = True

    def __post_init__(self):
        p = self.vision_config.get('patch_size', 16)
        if isinstance(p, int):
            self.vision_config['patch_size'] = (p, p, p)
        self.vision_config['patch_size'] = tuple(self.vision_config['patch_size'])
        if self.vision_config.get('pos_embed_shape') is not None:      # (else it arrives through `vision_override`, mmmm.py:147-151)
            self.vision_config['pos_embed_shape'] = tuple(self.vision_config['pos_embed_shape'])
