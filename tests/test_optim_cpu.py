"""Host logic of the optimizer side: the warm-up + cosine schedule of conf/phase-vg/fit.yaml:33-42, the
torch.optim.AdamW-layout state dict of FlatAdamW (tensor bookkeeping only — the fused kernel itself is tested on the GPU)."""
import math

import pytest
import torch


def test_cosine_schedule_closed_form_and_frequency():
    from mmmm_amd.optim import CosineLRSchedule
    base, T, W = 5e-5, 40000, 2000                      # conf/phase-vg/fit.yaml: lr, max_steps, warmup_t
    s1 = CosineLRSchedule(base, t_initial=T, warmup_t=W, warmup_prefix=True, frequency=1)
    assert s1(0) == 0.0
    assert s1(1000) == pytest.approx(base * 0.5)
    assert s1(W) == pytest.approx(base)                 # warm-up prefix: the cosine starts at its maximum right after warm-up
    assert s1(W + T // 2) == pytest.approx(base * 0.5)
    assert s1(W + T) == 0.0 and s1(W + T + 123) == 0.0  # cycle_limit 1: lr_min after the cycle
    for t in (W + 1, W + 777, W + 39999):
        assert s1(t) == pytest.approx(0.5 * base * (1 + math.cos(math.pi * (t - W) / T)))
    # without the prefix the cosine clock includes the warm-up steps
    s0 = CosineLRSchedule(base, t_initial=T, warmup_t=W, warmup_prefix=False)
    assert s0(W) == pytest.approx(0.5 * base * (1 + math.cos(math.pi * W / T)))
    # frequency 250: piecewise constant, the constructor's value holds for the first 250 steps
    s250 = CosineLRSchedule(base, t_initial=T, warmup_t=W, warmup_prefix=True, frequency=250)
    assert all(s250(t) == 0.0 for t in (0, 1, 249))
    assert s250(250) == s250(499) == pytest.approx(base * 250 / W)
    assert s250(500) == pytest.approx(base * 500 / W)
    vals = [s250(t) for t in range(0, W + T + 500, 250)]
    assert max(vals) == pytest.approx(base) and vals[-1] == 0.0


def test_flat_adamw_state_dict_has_torch_layout_and_round_trips():
    from mmmm_amd.ddp import BucketedGradAllReduce
    from mmmm_amd.optim import FlatAdamW
    from mmmm_amd.param import NoWeightDecayParameter
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(6, 5)), NoWeightDecayParameter(torch.randn(5)), torch.nn.Parameter(torch.randn(11))]
    ddp = BucketedGradAllReduce(ps, world_size=1, bucket_bytes=64)
    opt = FlatAdamW(ddp, lr=1e-3, weight_decay=0.01)
    for p, v in zip(ps, (ps[0].detach().clone(), ps[1].detach().clone(), ps[2].detach().clone())):
        assert torch.equal(p, v)                        # re-homing into the flat buffers keeps the values
    assert opt.state_dict()['state'] == {}
    opt.step_count = 5
    for m, v in zip(opt.exp_avg, opt.exp_avg_sq):
        m.normal_(); v.uniform_()
    sd = opt.state_dict()
    assert [g['weight_decay'] for g in sd['param_groups']] == [0.01, 0.0]
    assert sd['param_groups'][0]['params'] == [0, 2] and sd['param_groups'][1]['params'] == [1]
    assert all(tuple(sd['state'][i]['exp_avg'].shape) == tuple(p.shape) and float(sd['state'][i]['step']) == 5 for i, p in enumerate(ps))
    # the same dict loads into torch.optim.AdamW with the same grouping
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    ropt = torch.optim.AdamW([{'params': [ref[0], ref[2]], 'weight_decay': 0.01}, {'params': [ref[1]], 'weight_decay': 0.0}], lr=1e-3)
    remap = {0: 0, 2: 1, 1: 2}                           # torch numbers parameters group by group
    rsd = {'state': {remap[i]: st for i, st in sd['state'].items()},
           'param_groups': [dict(g, params=[remap[i] for i in g['params']]) for g in sd['param_groups']]}
    ropt.load_state_dict(rsd)
    assert torch.equal(ropt.state[ref[1]]['exp_avg'], sd['state'][1]['exp_avg'])
    # and back
    ps2 = [torch.nn.Parameter(torch.randn(6, 5)), NoWeightDecayParameter(torch.randn(5)), torch.nn.Parameter(torch.randn(11))]
    ddp2 = BucketedGradAllReduce(ps2, world_size=1, bucket_bytes=64)
    opt2 = FlatAdamW(ddp2, lr=1e-3, weight_decay=0.01)
    opt2.load_state_dict(ropt.state_dict())
    assert opt2.step_count == 5
    sd2 = opt2.state_dict()
    for i in range(3):
        assert torch.equal(sd2['state'][i]['exp_avg'], sd['state'][i]['exp_avg'])
        assert torch.equal(sd2['state'][i]['exp_avg_sq'], sd['state'][i]['exp_avg_sq'])
    with pytest.raises(ValueError):
        opt2.load_state_dict({'state': {}, 'param_groups': [{'params': [0, 1]}]})


def test_state_dict_is_numbered_by_registration_order_whatever_the_bucket_order():
    """buckets filled in gradient-production order (order='given': heads first, layer 0 last) must not renumber the optimizer state: a
    torch.optim.AdamW(model.parameters()) checkpoint has to land on the same parameters — same-shaped LoRA factors of different layers
    cannot be told apart by a shape check (advisor, round 4)"""
    from mmmm_amd.ddp import BucketedGradAllReduce
    from mmmm_amd.optim import FlatAdamW
    torch.manual_seed(1)
    reg = [torch.nn.Parameter(torch.randn(4, 3)) for _ in range(5)]            # five same-shaped parameters, registration order 0..4
    ropt = torch.optim.AdamW(reg, lr=1e-3, weight_decay=0.01)
    for i, p in enumerate(reg):
        p.grad = torch.full_like(p, float(i + 1))
    ropt.step()
    sd = ropt.state_dict()
    mine = [torch.nn.Parameter(p.detach().clone()) for p in reg]
    production = [mine[i] for i in (4, 2, 3, 0, 1)]                            # what grad_production_order would hand over
    ddp = BucketedGradAllReduce(production, world_size=1, bucket_bytes=64, order='given', registration=mine)
    opt = FlatAdamW(ddp, lr=1e-3, weight_decay=0.01)
    opt.load_state_dict(sd)
    out = opt.state_dict()
    for i in range(5):
        assert torch.equal(out['state'][i]['exp_avg'], sd['state'][i]['exp_avg']), i
        # and the moments sit in the slot of THAT parameter inside the flat buffers
        p, bi, off = opt._slots()[i]
        assert p is mine[i]
        assert torch.equal(opt.exp_avg[bi][off:off + p.numel()].view_as(p), sd['state'][i]['exp_avg'])
    with pytest.raises(ValueError):
        BucketedGradAllReduce(production, world_size=1, order='given', registration=mine[:4])


def test_parameters_resolve_by_address_after_the_optimizer_moved_them():
    """FlatAdamW re-homes every parameter into its flat buffers; the address registry that lets a checkpointed layer find the real
    Parameter behind a detached alias must follow (advisor, round 4: it kept the old addresses and every alias missed)"""
    from mmmm_amd import functional as Fh
    from mmmm_amd.ddp import BucketedGradAllReduce
    from mmmm_amd.optim import FlatAdamW
    ps = [torch.nn.Parameter(torch.randn(8, 8)), torch.nn.Parameter(torch.randn(16))]
    ddp = BucketedGradAllReduce(ps, world_size=1, bucket_bytes=1 << 10)
    before = [p.data_ptr() for p in ps]
    assert all(Fh._real_param(p.detach()) is p for p in ps)
    opt = FlatAdamW(ddp, lr=1e-3)
    assert all(p.data_ptr() != b for p, b in zip(ps, before))
    assert all(Fh._real_param(p.detach()) is p for p in ps)
    assert all(Fh.PARAM_BY_PTR.get(b) is None for b in before)
    # a deferred 1/world stays pending until the buckets are zeroed (a norm taken after step() still has to apply it)
    ddp.grad_scale = 0.25
    assert float(opt.ddp.grad_scale) == 0.25
    ddp.zero_grad()
    assert ddp.grad_scale == 1.0
    ddp.remove()


def test_reference_no_weight_decay_set_is_marked():
    """the parameters the reference declares as NoWeightDecayParameter (modeling_cogvlm.py:33, visual.py:32-35,189-190,
    segvol/modeling/image_encoder.py:56) carry the marker here — and nothing else does"""
    from mmmm_amd.models.cogvlm.configuration_cogvlm import CogVLMConfig
    from mmmm_amd.models.cogvlm.modeling_cogvlm import CogVLMForCausalLM
    from mmmm_amd.models.segvol import build_sam
    from mmmm_amd.param import no_weight_decay
    cfg = CogVLMConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2, vocab_size=128)
    cfg.vision_config.update(hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_heads=2, pos_embed_shape=(2, 4, 4),
                             patch_size=(4, 16, 16))
    m = CogVLMForCausalLM(cfg)
    marked = sorted(n for n, p in m.named_parameters() if no_weight_decay(p))
    assert marked == sorted([
        'model.layers.0.input_layernorm.weight', 'model.layers.0.post_attention_layernorm.weight', 'model.norm.weight',
        'model.vision.boi', 'model.vision.eoi', 'model.vision.patch_embedding.cls_embedding.weight',
        'model.vision.patch_embedding.cls_pos_embed.weight', 'model.vision.patch_embedding.position_embedding.weight'])
    sam = build_sam(patch_size=16, pos_embed_shape=(2, 4, 4))
    assert [n for n, p in sam.named_parameters() if no_weight_decay(p)] == ['image_encoder.patch_embedding.position_embeddings.weight']
