"""Checkpoint I/O: Hugging Face sharded safetensors -> model (with the reference's 2-D -> 3-D adapters and
the vocabulary extension), PEFT adapter round trip. CPU only; the checkpoints are synthesised by the test."""
import json

import torch

from mmmm_amd.models.checkpoint import adapter_state_dict, load_adapter, load_pretrained, save_adapter


def _tiny(vocab=40):
    from mmmm_amd.models.cogvlm.configuration_cogvlm import CogVLMConfig
    from mmmm_amd.models.cogvlm.modeling_cogvlm import CogVLMForCausalLM
    cfg = CogVLMConfig(vocab_size=vocab, hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2,
                       vision_config=dict(in_channels=3, hidden_size=16, num_heads=2, num_hidden_layers=1, intermediate_size=32,
                                          layer_norm_eps=1e-6, patch_size=(4, 2, 2), pos_embed_shape=(2, 3, 4), pt_pos_embed_shape=(3, 4)))
    return CogVLMForCausalLM(cfg)


def _hf_style_state(model, base_vocab):
    """what THUDM/cogvlm-chat-hf would hold for this architecture: 2-D patch kernel, flat position table with the cls row,
    bare cls parameter, `base_vocab` vocabulary rows"""
    g = torch.Generator().manual_seed(0)
    sd = {}
    for k, v in model.state_dict().items():
        if k.endswith('patch_embedding.proj.weight'):
            sd[k] = torch.randn(v.shape[0], v.shape[1], *v.shape[3:], generator=g)
        elif k.endswith('patch_embedding.position_embedding.weight'):
            sd[k] = torch.randn(1 + 12, v.shape[1], generator=g)
        elif k.endswith('patch_embedding.cls_pos_embed.weight'):
            continue                                                    # lives in row 0 of the flat table
        elif k.endswith('patch_embedding.cls_embedding.weight'):
            sd[k[:-len('.weight')]] = torch.randn(v.shape, generator=g)  # pre-ParameterWrapper name
        elif k.endswith(('embed_tokens.weight', 'lm_head.weight')):
            sd[k] = torch.randn(base_vocab, v.shape[1], generator=g)
        else:
            sd[k] = torch.randn(v.shape, generator=g)
    return sd


def test_load_pretrained_sharded_safetensors(tmp_path):
    from safetensors.torch import save_file
    model = _tiny(vocab=40)
    init_embed = model.model.embed_tokens.weight.detach().clone()
    sd = _hf_style_state(model, base_vocab=32)
    keys = sorted(sd)
    shards = {'model-00001-of-00002.safetensors': keys[:len(keys) // 2], 'model-00002-of-00002.safetensors': keys[len(keys) // 2:]}
    for name, ks in shards.items():
        save_file({k: sd[k].contiguous() for k in ks}, str(tmp_path / name))
    (tmp_path / 'model.safetensors.index.json').write_text(json.dumps({'weight_map': {k: n for n, ks in shards.items() for k in ks}}))
    missing, unexpected = load_pretrained(model, tmp_path, verbose=False)
    assert not unexpected and not missing
    got = model.state_dict()
    pe = 'model.vision.patch_embedding.'
    # vocabulary: checkpoint rows first, the 8 extra rows keep their initialisation
    assert torch.equal(got['model.embed_tokens.weight'][:32], sd['model.embed_tokens.weight'])
    assert torch.equal(got['model.embed_tokens.weight'][32:], init_embed[32:])
    assert torch.equal(got['lm_head.weight'][:32], sd['lm_head.weight'])
    # 2-D kernel -> mean inflation over depth 4; flat table -> cls row + [1, C, d, h, w] grid repeated along depth
    w2 = sd[pe + 'proj.weight']
    assert torch.allclose(got[pe + 'proj.weight'], (w2 / 4)[:, :, None].expand(-1, -1, 4, -1, -1))
    table = sd[pe + 'position_embedding.weight']
    assert torch.equal(got[pe + 'cls_pos_embed.weight'], table[0:1])
    grid = table[1:].reshape(3, 4, -1).permute(2, 0, 1)
    assert torch.equal(got[pe + 'position_embedding.weight'], grid[None, :, None].expand(1, -1, 2, -1, -1))
    assert torch.equal(got[pe + 'cls_embedding.weight'], sd[pe + 'cls_embedding'])
    k = 'model.layers.0.mlp.language_mlp.gate_proj.weight'
    assert torch.equal(got[k], sd[k])


def test_peft_adapter_round_trip(tmp_path):
    from mmmm_amd.models.lora import LoraConfig
    from mmmm_amd.utils import apply_lora
    from mmmm_amd.models.mmmm import MMMMForCausalLM, VisionArgs
    cfg = LoraConfig(r=64, lora_alpha=8, lora_dropout=0.05, use_rslora=True)
    mk = lambda: MMMMForCausalLM(_tiny().config, vision_override=VisionArgs(pos_embed_shape=(2, 3, 4), pt_pos_embed_shape=(3, 4),
                                                                            patch_size=(4, 2, 2)))
    a, b = mk(), mk()
    for m in (a, b):
        apply_lora(m, cfg)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for p in a.parameters():
            if p.requires_grad:
                p.copy_(torch.randn(p.shape, generator=g))
    save_adapter(a, tmp_path, cfg)
    keys = adapter_state_dict(a).keys()
    assert all(k.startswith('base_model.model.') and '.default.' not in k for k in keys)
    assert any(k.endswith('self_attn.vision_expert_query_key_value.lora_A.weight') for k in keys)
    conf = json.loads((tmp_path / 'adapter_config.json').read_text())
    assert conf['r'] == 64 and conf['use_rslora'] and 'model.embed_tokens' in conf['modules_to_save']
    missing, unexpected = load_adapter(b, tmp_path)
    assert not missing and not unexpected
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        if p.requires_grad:
            assert torch.equal(p, q), n
