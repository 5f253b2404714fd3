"""The outer drop-in boundary driven from the reference's OWN config files (scripts/cli.py:77-92):
conf/phase-{vg,vlm,grg}/model.yaml + conf/lora.yaml are parsed as they are, `class_path: mmmm.…` is remapped onto this package,
every `init_args` key must be accepted by the builders, and the CLI's PEFT sequence (get_lora_modules_default -> get_peft_model ->
set_peft_model -> load_adapter / load_default_adapter) must run. Needs /root/reference (build container only; the GPU box has no
copy, and nothing there reads it)."""
from pathlib import Path

import pytest
import torch
import yaml

CONF = Path('/root/reference/conf')
pytestmark = pytest.mark.skipif(not CONF.exists(), reason='the reference configs only exist in the build container')


def _tiny_overrides(path, class_path, args):
    """what a user without the 35 GB checkpoints overrides on the command line: checkpoint paths off, tiny widths"""
    from mmmm_amd.models.cogvlm.configuration_cogvlm import CogVLMConfig
    args = dict(args)
    if class_path.endswith(('build_sam', 'build_instance_sam')):
        assert set(args) <= {'patch_size', 'pos_embed_shape', 'num_instances', 'checkpoint', 'state_dict_key', 'weight_prefix'}
        assert str(args['checkpoint']).startswith('pre-trained/') and args['state_dict_key'] == 'state_dict' and args['weight_prefix'] == 'sam.'
        args.update(checkpoint=None, embed_dim=32, encoder_num_layers=2, num_heads=2)
    if class_path.endswith('models.build'):
        assert args['pretrained_model_name_or_path'] == 'THUDM/cogvlm-chat-hf'
        assert isinstance(args['vision_override'], dict) and args['vision_override']['patch_size'] == 16
        args['pretrained_model_name_or_path'] = None
        args['config'] = CogVLMConfig(vocab_size=192, hidden_size=128, intermediate_size=256, num_hidden_layers=1, num_attention_heads=2,
                                      vision_config=dict(in_channels=3, hidden_size=128, num_heads=2, num_hidden_layers=1,
                                                         intermediate_size=256, layer_norm_eps=1e-6))
        if isinstance(args.get('tokenizer'), str):       # `tokenizer: ../tokenizer.yaml` is linked in by the CLI (cli.py:69-70)
            from mmmm_amd.data.synthetic import SpecialTokens
            args['tokenizer'] = SpecialTokens(base_vocab=184)
    return args


def _build(phase: str):
    from mmmm_amd.utils import instantiate
    spec = yaml.safe_load((CONF / phase / 'model.yaml').read_text())
    assert spec['class_path'] == 'mmmm.models.build'
    return instantiate(spec, overrides=_tiny_overrides), spec


@pytest.mark.parametrize('phase', ['phase-vg', 'phase-grg'])
def test_grounding_phase_model_yaml_instantiates(phase):
    from mmmm_amd.models import InstanceSam, MMMMForCausalLM, Sam
    from mmmm_amd.models.loss import DiceFocalLoss
    from mmmm_amd.models.segvol.modeling.sam import InstanceSamLoss
    m, spec = _build(phase)
    assert isinstance(m, MMMMForCausalLM) and isinstance(m.sam, Sam) and isinstance(m.isam_model, InstanceSam)
    assert isinstance(m.mask_loss, DiceFocalLoss) and isinstance(m.isam_loss, InstanceSamLoss) and m.isam_loss.mask_loss is m.mask_loss
    ia = spec['init_args']
    vc = m.config.vision_config
    assert tuple(vc['pos_embed_shape']) == tuple(ia['vision_override']['pos_embed_shape']) == (8, 32, 32)
    assert tuple(vc['pt_pos_embed_shape']) == (35, 35) and tuple(vc['patch_size']) == (16, 16, 16)
    ml, il = ia['mask_loss']['init_args'], ia['isam_loss']['init_args']
    assert (m.mask_loss.dice_weight, m.mask_loss.focal_weight, m.mask_loss.focal_gamma) == (ml['dice_weight'], ml['focal_weight'], ml['focal_gamma'])
    for k, v in il.items():
        assert getattr(m.isam_loss, k) == v, k
    assert m.isam_model.mask_decoder.num_mask_tokens - 1 == ia['isam']['init_args']['num_instances'] or \
        ia['isam']['init_args']['num_instances'] == 6
    # defaults of build(): heads frozen, vg_proj in place, fp32 islands named as the reference names them
    assert not any(p.requires_grad for p in m.sam.parameters()) and not any(p.requires_grad for p in m.isam_model.parameters())
    assert m.get_fp32_children() == ['sam', 'isam_model', 'vg_proj'] and m.vg_proj[2].out_features == m.sam.prompt_dim


def test_phase_vlm_model_yaml_instantiates_without_heads():
    m, _ = _build('phase-vlm')
    assert m.sam is None and m.isam_model is None and not hasattr(m, 'vg_proj')


def test_cli_peft_sequence_from_lora_yaml(tmp_path):
    """scripts/cli.py:77-88 with conf/lora.yaml, on the phase-vg model: the target / modules_to_save lists are the reference's (F9)"""
    from mmmm_amd.peft import LoraConfig, get_peft_model
    from mmmm_amd.utils import get_lora_modules_default
    m, _ = _build('phase-vg')
    lora_yaml = yaml.safe_load((CONF / 'lora.yaml').read_text())
    assert set(lora_yaml) == {'r', 'lora_alpha', 'lora_dropout', 'use_rslora'}
    lora_config = LoraConfig(**lora_yaml)
    lora_config.target_modules, lora_config.modules_to_save = get_lora_modules_default(m)
    ref = torch.load('tests/golden/f9_lora_targets.pt', weights_only=False)['freeze_sam=True']
    assert lora_config.target_modules == ref['target_modules'] and lora_config.modules_to_save == ref['modules_to_save']
    with pytest.raises(AttributeError):
        m.peft_model
    peft_model = get_peft_model(m, lora_config)
    m.set_peft_model(peft_model)
    assert m.peft_model is peft_model and 'peft_model' not in dict(m.named_modules())
    lin = m.model.layers[0].self_attn.language_expert_dense
    assert lin.lora_cfg.r == 64 and abs(lin.lora_cfg.scale - 8 / 64 ** 0.5) < 1e-12 and lin.lora_cfg.lora_dropout == 0.05
    trainable = {n for n, p in m.named_parameters() if p.requires_grad}
    assert 'model.layers.0.self_attn.language_expert_dense.lora_A.default.weight' in trainable and 'lm_head.weight' in trainable
    assert not any(n.startswith(('sam.', 'isam_model.')) for n in trainable) and 'vg_proj.0.weight' in trainable
    # save -> perturb -> load_default_adapter (mmmm.py:154-155: <ckpt_dir>/adapter) restores the weights; the adapter already exists,
    # so PEFT leaves requires_grad alone (its inference-mode freeze is for adapter names it creates) and only calls eval()
    with torch.no_grad():
        lin.B.normal_()
    want = {n: p.detach().clone() for n, p in m.named_parameters() if p.requires_grad}
    peft_model.save_pretrained(tmp_path / 'adapter')
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.requires_grad:
                p.add_(1.0)
    m.load_default_adapter(tmp_path)
    for n, p in m.named_parameters():
        if n in want:
            assert torch.equal(p.detach(), want[n]), n
    assert {n for n, p in m.named_parameters() if p.requires_grad} == set(want) and not m.training
    m.train()
    # `fit` passes is_trainable=True (cli.py:87)
    m2, _ = _build('phase-vg')
    lc = LoraConfig(**lora_yaml)
    lc.target_modules, lc.modules_to_save = get_lora_modules_default(m2)
    pm2 = get_peft_model(m2, lc)
    m2.set_peft_model(pm2)
    pm2.load_adapter(str(tmp_path / 'adapter'), 'default', is_trainable=True)
    assert {n for n, p in m2.named_parameters() if p.requires_grad} == set(want)
    assert torch.equal(m2.model.layers[0].self_attn.language_expert_dense.B.detach(), want['model.layers.0.self_attn.language_expert_dense.lora_B.default.weight'])


def test_fit_yaml_optimizer_and_schedule_values_are_the_ones_bench_uses():
    fit = yaml.safe_load((CONF / 'phase-vg' / 'fit.yaml').read_text())
    assert fit['trainer']['precision'] == 'bf16-true' and fit['trainer']['gradient_clip_val'] == 1
    oa = fit['optim']['optimizer']['init_args']
    assert (float(oa['lr']), oa['weight_decay']) == (5e-5, 0.01)          # (YAML 1.1 reads '5e-5' as a string)
    text = Path(__file__).resolve().parents[1].joinpath('bench.py').read_text()
    assert 'lr=5e-5, weight_decay=0.01, max_grad_norm=1.0' in text
