"""Host-side logic that needs no GPU: LoRA target discovery vs the reference's lists (fixture F9), state-dict key
parity of the grounding heads, synthetic batch layout vs prepare_vlm_inputs' rules, FLOP model vs SURVEY §8d."""
from pathlib import Path

import torch

from tests import _tiny  # noqa: F401


def _tiny_product(n=1):
    from mmmm_amd.models.cogvlm.configuration_cogvlm import CogVLMConfig
    from mmmm_amd.models.mmmm import MMMMForCausalLM, VisionArgs
    cfg = CogVLMConfig(vocab_size=192, hidden_size=128, intermediate_size=256, num_hidden_layers=n, num_attention_heads=2,
                       vision_config=dict(in_channels=3, hidden_size=128, num_heads=2, num_hidden_layers=n, intermediate_size=256,
                                          layer_norm_eps=1e-6, patch_size=(4, 8, 8), pos_embed_shape=(2, 2, 4)))
    return cfg, VisionArgs(pos_embed_shape=(2, 2, 4), patch_size=(4, 8, 8))


def test_lora_target_discovery_matches_reference_lists():
    from mmmm_amd.models import build_instance_sam, build_sam
    from mmmm_amd.models.mmmm import MMMMForCausalLM
    ref = torch.load('tests/golden/f9_lora_targets.pt', weights_only=False)
    for freeze in (True, False):
        cfg, va = _tiny_product()
        sam = build_sam(embed_dim=32, encoder_num_layers=2, num_heads=2, patch_size=(4, 8, 8), pos_embed_shape=(2, 2, 4))
        isam = build_instance_sam(embed_dim=32, encoder_num_layers=2, num_heads=2, patch_size=(4, 8, 8), pos_embed_shape=(2, 2, 4), num_instances=6)
        m = MMMMForCausalLM.build(None, vision_override=va, sam=sam, isam=isam, freeze_sam=freeze, freeze_isam=freeze, config=cfg)
        targets, saves = m.get_lora_modules(prefix='')
        want = ref[f'freeze_sam={freeze}']
        assert targets == want['target_modules']
        assert saves == want['modules_to_save']


def test_apply_lora_trainable_set():
    from mmmm_amd.models.lora import LoraConfig
    from mmmm_amd.models.mmmm import MMMMForCausalLM
    from mmmm_amd.utils import apply_lora
    cfg, va = _tiny_product()
    m = MMMMForCausalLM(cfg, vision_override=va)
    apply_lora(m, LoraConfig())
    names = [n for n, p in m.named_parameters() if p.requires_grad]
    assert 'model.embed_tokens.weight' in names and 'lm_head.weight' in names
    assert 'model.layers.0.mlp.vision_mlp.up_proj.lora_A.default.weight' in names
    assert 'model.layers.0.mlp.vision_mlp.up_proj.weight' not in names
    assert 'model.vision.boi' not in names and 'model.vision.patch_embedding.proj.weight' in names


def test_grounding_head_state_dict_keys_equal_reference():
    from mmmm_amd.models import build_instance_sam, build_sam
    ref = torch.load('tests/golden/f6_sam.pt', weights_only=False)
    s = build_sam(embed_dim=32, encoder_num_layers=2, num_heads=2, patch_size=(4, 8, 8), pos_embed_shape=(2, 2, 4))
    i = build_instance_sam(embed_dim=32, encoder_num_layers=2, num_heads=2, patch_size=(4, 8, 8), pos_embed_shape=(2, 2, 4), num_instances=6)
    assert {k: tuple(v.shape) for k, v in s.state_dict().items()} == {k: tuple(v.shape) for k, v in ref['sam_state'].items()}
    assert {k: tuple(v.shape) for k, v in i.state_dict().items()} == {k: tuple(v.shape) for k, v in ref['isam_state'].items()}


def test_lm_state_dict_keys_equal_reference():
    from mmmm_amd.models.mmmm import MMMMForCausalLM, VisionArgs
    from mmmm_amd.models.cogvlm.configuration_cogvlm import CogVLMConfig
    ref = torch.load('tests/golden/f5_tiny_lm.pt', weights_only=False)['state_dict']
    cfg = CogVLMConfig(vocab_size=160, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                       vision_config=dict(in_channels=3, hidden_size=32, num_heads=2, num_hidden_layers=2, intermediate_size=64,
                                          layer_norm_eps=1e-6, patch_size=(4, 8, 8), pos_embed_shape=(2, 2, 4)))
    m = MMMMForCausalLM(cfg, vision_override=VisionArgs(pos_embed_shape=(2, 2, 4), patch_size=(4, 8, 8)))
    mine = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    theirs = {k: tuple(v.shape) for k, v in ref.items() if 'inv_freq' not in k}
    assert mine == theirs


def test_synthetic_batch_follows_prepare_vlm_inputs_layout():
    from mmmm_amd.data.synthetic import SpecialTokens, make_batch
    from oracle import vividmed as O
    tok = SpecialTokens(base_vocab=184)
    b = make_batch([(3, 1, 16, 32), (3, 8, 16, 32)], [(1, 8, 8), (4, 8, 8)], [(1, 2, 2), (2, 2, 2)], [24, 30], tok=tok, instance=[False, True])
    vi = b['vlm_inputs']
    ids, tt, pos, am, lab = vi['input_ids'], vi['token_type_ids'], vi['position_ids'], vi['attention_mask'], vi['labels']
    assert ids[0, 0] == 1 and ids[0, 5] == tok.grd_token_id and tt[0, 1:5].tolist() == [1, 1, 1, 1] and tt[0, 5] == 0
    assert pos[0, :6].tolist() == [0, 1, 2, 2, 3, 4]
    n0 = int(am[0].sum())
    assert torch.equal(lab[0, 6:n0 - 1], ids[0, 7:n0]) and lab[0, n0 - 1] == 2 and torch.all(lab[0, :6] == -100)
    # position does not advance after <p> and at </p> (mmmm/data/utils.py:20-29)
    t = ids[0, 6:n0]
    p = pos[0, 6:n0]
    for i in range(1, len(t)):
        stay = t[i - 1] == tok.bop_token_id or t[i] == tok.eop_token_id
        assert p[i] == p[i - 1] + (0 if stay else 1)
    v, l = O.get_expert_mask(tt, am.bool())
    assert int(v[0].sum()) == 3          # boi + 2 patches; eoi is routed to the language expert (reference quirk)
    assert b['index_offsets'][1].shape[1] == 2 and b['masks'][0].dtype == torch.bool


def test_flop_model_matches_survey_table():
    import bench
    from mmmm_amd.models.cogvlm.configuration_cogvlm import CogVLMConfig
    f = bench.train_flops_per_sample(bench.WORKLOADS['phase-vg-448'], CogVLMConfig(), True) / 1e12
    assert abs(f - 28.59) < 0.02
    f = bench.train_flops_per_sample(bench.WORKLOADS['phase-grg-3d'], CogVLMConfig(), True) / 1e12
    assert abs(f - 59.60) < 0.02


def test_lora_factor_cache_follows_replaced_parameters():
    """`Linear.A` / `.B` cache the Parameter objects (the step reads them ~3 700 times); whatever REPLACES the parameters — load_state_dict(assign=True),
    to_empty(), a dtype / device conversion — must drop the cache, or forward and the factor-gradient queue would keep using tensors the module no
    longer owns (advisor, round 5)"""
    from mmmm_amd.models.lora import Linear, LoraConfig
    lin = Linear(8, 8)
    lin.add_lora(LoraConfig(r=4))
    a0, b0 = lin.A, lin.B
    assert a0 is lin.lora_A['default'].weight and b0 is lin.lora_B['default'].weight
    lin.load_state_dict({k: v.clone() + 1 for k, v in lin.state_dict().items()}, assign=True)
    assert lin.A is lin.lora_A['default'].weight and lin.A is not a0 and lin.B is lin.lora_B['default'].weight and lin.B is not b0
    lin.to(torch.float64)
    assert lin.A is lin.lora_A['default'].weight and lin.A.dtype == torch.float64
    _ = lin.A, lin.B
    lin.to_empty(device='cpu')
    assert lin.A is lin.lora_A['default'].weight and lin.B is lin.lora_B['default'].weight
    # a plain in-place load keeps the objects (and the cache)
    a1 = lin.A
    lin.load_state_dict({k: torch.zeros_like(v) for k, v in lin.state_dict().items()})
    assert lin.A is a1 and float(lin.A.detach().abs().sum()) == 0.0


def test_bench_set_reports_a_broken_import_instead_of_hiding_it(tmp_path, monkeypatch):
    """bench.apply_sets probes module prefixes: only 'this prefix is not a module' may be swallowed (advisor, round 5)"""
    import bench
    import pytest
    bench.apply_sets(['models.segvol.modeling.sam.InstanceSamLoss.fused=True'])        # a class constant behind a module prefix
    from mmmm_amd.models.segvol.modeling.sam import InstanceSamLoss
    assert InstanceSamLoss.fused is True
    with pytest.raises(SystemExit):
        bench.apply_sets(['functional.NO_SUCH_CONSTANT=1'])
    import mmmm_amd
    (Path(mmmm_amd.__file__).parent / '_broken_probe.py').write_text('import a_module_that_does_not_exist_anywhere\n')
    try:
        with pytest.raises(ModuleNotFoundError):
            bench.apply_sets(['_broken_probe.X=1'])
    finally:
        (Path(mmmm_amd.__file__).parent / '_broken_probe.py').unlink()
