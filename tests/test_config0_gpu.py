"""BASELINE.json configs[0] on the HIP path, end to end at the TRUE layer widths: phase-vg single step, one 1x224x224 2-D image
(Nv = 197 ViT tokens incl. cls, Np = 49 pooled image tokens, T = 64 text tokens, L = 1 + (49 + 2) + 1 + 64 = 117), batch 1,
CogVLM at 4096 / 11008 / 32 heads + EVA-ViT-E at 1792 / 15360 / 16 heads of 112 with REDUCED DEPTH (2 decoder + 2 ViT layers: the
plan of SURVEY.md §8c for this config — 17.6 B parameters in fp32 do not fit the oracle's host), vocabulary 32 008, LoRA r64 on every
linear, SAM-B + iSAM at full width and depth, unfrozen. The reference runs this config on the CPU in fp32; the product has no CPU
path, so the shape (ragged 197-row tiles, 49 image tokens, a 117-row decoder) runs on the GPU and is compared with the oracle
evaluated on the box's host cores in fp32 AND in bf16-true mode (MyPrecision: fp32 islands stay fp32):

    training_step: loss, every logged scalar, 16 parameter gradients (LoRA factors of both towers, embedding, lm_head, norm gains,
                   vg_proj, SAM / iSAM weights)
    forward:       logits on valid rows, every hidden state, vg prompts, masks / boxes / discriminator logits

with the evidence-based bf16 bound of tests/test_truewidth_gpu.py (|hip - fp32| <= a e_ref, |hip - bf16| <= b e_ref, e_ref = the
oracle's own bf16-vs-fp32 gap) and 1e-4 for the fp32 islands when they are fed identical prompts.

The second test is the DEPTH SWEEP the round-2 verdict asked for: 1, 2, 4, 8 decoder layers at width 4096 on the same input,
|hip - oracle_fp32| next to |oracle_bf16 - oracle_fp32| for the last hidden state and the logits, written to
gpurun_out/r3_parity_config0.json (committed as profiles/r3_parity_report.json) — how the bf16 gap grows with depth, so the full-depth
(32-layer) logits error is an evidenced extrapolation next to north_star's 1e-3."""
import json
import math
import os
from pathlib import Path

import pytest
import torch

import contextlib

from tests._gpu_common import cpu, lora_dropout_on, oracle_cfg, oracle_state, rel

pytestmark = pytest.mark.gpu
REPORT: dict = {}
FP32_PREFIXES = ('sam.', 'isam_model.', 'vg_proj.')


def _randomize_on_device_(module: torch.nn.Module, seed: int):
    """tests/_gpu_common.randomize_ with the device's generator (1.3 - 3.5 B parameters: the host generator takes minutes)"""
    dev = next(module.parameters()).device
    g = torch.Generator(device=dev).manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            leaf = name.rsplit('.', 1)[-1]
            if p.ndim == 1 and leaf == 'weight':
                v = 1 + 0.1 * torch.randn(p.shape, generator=g, device=dev)
            elif leaf == 'bias':
                v = 0.05 * torch.randn(p.shape, generator=g, device=dev)
            else:
                fan_in = math.prod(p.shape[1:]) if p.ndim >= 2 else p.shape[0]
                std = 0.3 if (p.ndim < 2 or 'embed' in name or name.endswith(('boi', 'eoi'))) else 1.0 / math.sqrt(fan_in)
                if 'lora_B' in name:
                    std = 0.05
                if 'query_key_value' in name or name.endswith(('qkv.weight', 'q_proj.weight', 'k_proj.weight')):
                    std *= 0.35          # keep softmax out of saturation (see _gpu_common.randomize_)
                v = std * torch.randn(p.shape, generator=g, device=dev)
            p.copy_(v.to(p.dtype))
        for name, b in module.named_buffers():
            if 'positional_encoding_gaussian_matrix' in name:
                b.copy_(torch.randn(b.shape, generator=g, device=dev).to(b.dtype))


def _bounds(what, got, ref16, ref32, a=1.3, b=1.6, floor=1e-4):
    e_ref, e_got, d = rel(ref16.float(), ref32), rel(got.float(), ref32), rel(got.float(), ref16.float())
    REPORT[what] = dict(e_ref=e_ref, e_hip=e_got, hip_vs_oracle_bf16=d)
    if os.environ.get('VM_PARITY_NOASSERT') == '1':       # (diagnostic runs: collect every tensor's numbers, tools/parity_seeds.sh)
        return
    assert e_got <= a * e_ref + floor and d <= b * e_ref + floor, (what, REPORT[what])


def _bf16_true(sd: dict) -> dict:
    """MyPrecision.convert_module (mmmm.py:481-492)"""
    return {k: (v if (not v.is_floating_point() or k.startswith(FP32_PREFIXES)) else v.bfloat16()) for k, v in sd.items()}


@pytest.fixture(scope='module')
def cfg0(dev):
    from mmmm_amd.data.synthetic import SpecialTokens
    from mmmm_amd.models import build_instance_sam, build_sam
    from mmmm_amd.models.cogvlm.configuration_cogvlm import CogVLMConfig
    from mmmm_amd.models.lora import LoraConfig
    from mmmm_amd.models.loss import DiceFocalLoss
    from mmmm_amd.models.mmmm import MMMMForCausalLM, MyPrecision, VisionArgs
    from mmmm_amd.models.segvol.modeling.sam import InstanceSamLoss
    from mmmm_amd.utils import apply_lora
    cfg = CogVLMConfig(num_hidden_layers=2)
    cfg.vision_config['num_hidden_layers'] = 2
    tok = SpecialTokens(base_vocab=32000)
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device(dev):
            torch.set_default_dtype(torch.float32)
            sam = build_sam(patch_size=16, pos_embed_shape=(8, 32, 32))
            isam = build_instance_sam(patch_size=16, num_instances=6, pos_embed_shape=(8, 32, 32))
            torch.set_default_dtype(torch.bfloat16)
            m = MMMMForCausalLM.build(None, vision_override=VisionArgs(pos_embed_shape=(8, 32, 32), pt_pos_embed_shape=(35, 35), patch_size=16),
                                      tokenizer=tok, sam=sam, isam=isam, mask_loss=DiceFocalLoss(dice_weight=2, focal_weight=2, focal_gamma=2),
                                      isam_loss=InstanceSamLoss(use_neg_mask=False, box_l1_weight=5, box_giou_weight=2, disc_weight=2,
                                                                disc_focal_gamma=2, disc_focal_alpha=0.85),
                                      config=cfg, freeze_sam=False, freeze_isam=False)
    finally:
        torch.set_default_dtype(torch.float32)
    m.vg_proj.float()
    apply_lora(m, LoraConfig(r=64, lora_alpha=8, lora_dropout=0.0, use_rslora=True))
    _randomize_on_device_(m, 2240 + int(os.environ.get('VM_PARITY_SEED', '0')))
    MyPrecision().convert_module(m)
    assert m.lm_head.weight.dtype == torch.bfloat16 and m.sam.image_encoder.norm.weight.dtype == torch.float32
    assert m.config.hidden_size == 4096 and m.config.vision_config['hidden_size'] == 1792 and m.config.vocab_size == 32008
    m.train()
    m.sam.eval(), m.isam_model.eval()
    m.on_fit_start()
    yield m, tok
    import gc
    del m
    gc.unfreeze()
    gc.collect()
    torch.cuda.empty_cache()


def _step_cfg(m, tok):
    from oracle import vividmed as O
    return O.StepCfg(lm=oracle_cfg(m.config), sam=O.SamCfg(), isam=O.SamCfg(num_instances=6, instance=True),
                     mask_loss=dict(dice_weight=2, focal_weight=2, focal_gamma=2), isam_loss=O.ISamLossCfg(),
                     bop_token_id=tok.bop_token_id, eop_token_id=tok.eop_token_id)


WATCH = (
    'model.layers.0.self_attn.vision_expert_query_key_value.lora_A.default.weight',
    'model.layers.0.self_attn.language_expert_query_key_value.lora_B.default.weight',
    'model.layers.0.mlp.language_mlp.gate_proj.lora_A.default.weight',
    'model.layers.1.self_attn.language_expert_dense.lora_B.default.weight',
    'model.layers.1.mlp.vision_mlp.down_proj.lora_A.default.weight',
    'model.layers.1.mlp.language_mlp.up_proj.lora_B.default.weight',
    'model.layers.1.post_attention_layernorm.weight',
    'model.norm.weight',
    'model.embed_tokens.weight',
    'lm_head.weight',
    'model.vision.transformer.layers.0.attention.query_key_value.lora_A.default.weight',
    'model.vision.transformer.layers.1.mlp.fc1.lora_B.default.weight',
    'model.vision.transformer.layers.1.mlp.fc2.lora_A.default.weight',
    'model.vision.linear_proj.dense_h_to_4h.lora_B.default.weight',
    'vg_proj.0.weight',
    'vg_proj.2.bias',
)
WATCH_HEAD = {False: ('sam.image_encoder.blocks.11.mlp.linear2.weight', 'sam.mask_decoder.transformer.layers.1.mlp.lin1.weight',
                      'sam.image_encoder.blocks.0.attn.qkv.weight'),
              True: ('isam_model.image_encoder.blocks.11.mlp.linear2.weight', 'isam_model.box_head.0.weight',
                     'isam_model.image_encoder.blocks.0.attn.qkv.weight')}


@pytest.mark.parametrize('instance,drop_p', [(False, 0.0), (True, 0.0), (False, 0.05)],
                         ids=['semantic->SAM', 'instance->iSAM', 'semantic->SAM, lora_dropout 0.05 (conf/lora.yaml:3)'])
def test_config0_true_width_reduced_depth_end_to_end_vs_oracle(dev, cfg0, instance, drop_p):
    from oracle import vividmed as O
    from mmmm_amd.data.synthetic import make_batch
    from mmmm_amd.models.mmmm import MyPrecision
    m, tok = cfg0
    tag = ('iSAM' if instance else 'SAM') + (f' p={drop_p}' if drop_p else '')
    batch = make_batch([(3, 1, 224, 224)], [(1, 16, 16)], [(1, 2, 2)], [64], tok=tok, seed=224 + instance, grounding=True, n_pairs=3,
                       instance=[instance], device=dev)
    vi = batch['vlm_inputs']
    assert vi['input_ids'].shape == (1, 117) and int(vi['token_type_ids'].sum()) == 51          # L = 117, Np + 2 = 51 (configs[0])
    batch = MyPrecision().convert_input(batch) | {'host': batch['host']}
    assert batch['image'][0].dtype == torch.bfloat16 and batch['grounding_image'][0].dtype == torch.float32
    # (drop_p > 0: every LoRA linear drops, and the oracle is handed the masks the HIP kernels drew — tests/_gpu_common.LoraMasks)
    with (lora_dropout_on(m, vi, drop_p) if drop_p > 0 else contextlib.nullcontext()) as lmasks:
        # ---- the step
        for p in m.parameters():
            if p.grad is not None:
                p.grad = None
        m.logged.clear()
        loss = m.training_step(batch)
        loss.backward()
        # ---- the same forward once more for the intermediate results (the masks are a function of (step, site, element): the same ones)
        with torch.no_grad():
            out = m(**vi, image=batch['image'], patch_size=batch['patch_size'], pool_size=batch['pool_size'], return_dict=True,
                    output_hidden_states=True, materialize_logits=True)
            prompts = m._get_vg_prompts(vi['input_ids'][:, 1:], out.hidden_states[-1][:, :-1].float(), [None])
            masks, boxes, disc = m.visual_grounding(vi['input_ids'][:, 1:], out.hidden_states[-1][:, :-1].float(), batch['grounding_image'],
                                                    batch['patch_size'], [None], batch['instance_mask'])
        assert len(out.hidden_states) == 3 and out.logits.shape == (1, 117, 32008)
        # ---- the oracle on the host cores: fp32 (exact on the model's weights) and bf16-true
        scfg = _step_cfg(m, tok)
        cb = cpu({k: v for k, v in batch.items() if k != 'host'})
        res = {}
        for mode in ('fp32', 'bf16'):
            sd = oracle_state(m)
            b = dict(cb)
            if mode == 'bf16':
                sd = _bf16_true(sd)
            else:
                b['image'] = [x.float() for x in b['image']]
                b['vlm_inputs'] = dict(b['vlm_inputs'], weight=b['vlm_inputs']['weight'].float())
            watch = WATCH + WATCH_HEAD[instance]
            sd = {k: (v.requires_grad_(True) if k in watch else v) for k, v in sd.items()}
            cap: dict = {}
            if lmasks is not None:
                lmasks.begin()
            l_, log = O.training_step(sd, scfg, b, rope_dtype=torch.bfloat16, capture=cap)
            l_.backward()
            res[mode] = dict(loss=l_.detach(), log=log, cap=cap, grads={k: sd[k].grad.float() for k in watch})
    r32, r16 = res['fp32'], res['bf16']
    am = cb['vlm_inputs']['attention_mask'].bool()
    # ---- loss and logged scalars: exactly the oracle's keys (= the reference's, mmmm.py:333-351)
    assert set(m.logged) == set(r32['log']), (sorted(m.logged), sorted(r32['log']))
    REPORT[f'{tag} loss'] = dict(hip=loss.item(), fp32=r32['loss'].item(), bf16=r16['loss'].item())
    for k, v32 in r32['log'].items():
        v16, got = float(r16['log'][k]), float(m.logged[k])
        e_ref = abs(v16 - float(v32)) / max(abs(float(v32)), 1e-12)
        REPORT[f'{tag} scalar {k}'] = dict(e_ref=e_ref, e_hip=abs(got - float(v32)) / max(abs(float(v32)), 1e-12))
        # one number, not a norm over many: its error does not average — absolute floor as in test_f13_training_step_bf16_true_vs_reference
        assert abs(got - float(v32)) <= (3 * e_ref + 5e-3) * max(abs(float(v32)), 1e-3), (k, got, v16, float(v32))
    # ---- forward: logits on valid rows, every hidden state, prompts, head outputs
    lm32, lm16 = r32['cap']['lm'], r16['cap']['lm']
    _bounds(f'{tag} logits (valid rows)', out.logits.cpu()[am], lm16.logits[am], lm32.logits[am])
    for i in range(1, 3):
        _bounds(f'{tag} hidden state {i}', out.hidden_states[i].float().cpu()[am], lm16.hidden_states[i][am], lm32.hidden_states[i][am])
    assert rel(out.hidden_states[0].float().cpu()[am], lm16.hidden_states[0].float()[am]) <= max(
        1.6 * rel(lm16.hidden_states[0].float()[am], lm32.hidden_states[0][am]), 1e-6)          # embedding + image scatter: the vision tower's error only
    _bounds(f'{tag} vg prompts', prompts[0].cpu(), r16['cap']['prompts'][0], r32['cap']['prompts'][0])
    if instance:
        _bounds(f'{tag} boxes', boxes[0].cpu(), r16['cap']['boxes'][0], r32['cap']['boxes'][0], floor=2e-4)
        _bounds(f'{tag} disc logits', disc[0].cpu(), r16['cap']['disc_logit'][0], r32['cap']['disc_logit'][0], floor=2e-4)
    else:
        assert masks[0].shape == r32['cap']['masks_logits'][0].shape == (3, 1, 224, 224)
        _bounds(f'{tag} mask logits', masks[0].cpu(), r16['cap']['masks_logits'][0], r32['cap']['masks_logits'][0], floor=2e-4)
    # ---- the fp32 islands fed the ORACLE's fp32 prompts: 1e-4 (their own error, without the bf16 language model in front)
    with torch.no_grad():
        p32 = [r32['cap']['prompts'][0].to(dev)]
        if instance:
            o = m.isam_model(batch['grounding_image'], batch['patch_size'], p32)
            e = dict(boxes=rel(o.boxes[0], r32['cap']['boxes'][0]), disc=rel(o.disc_logit[0], r32['cap']['disc_logit'][0]))
        else:
            e = dict(masks=rel(m.sam(batch['grounding_image'], batch['patch_size'], p32)[0], r32['cap']['masks_logits'][0]))
    REPORT[f'{tag} fp32 island on identical prompts'] = e
    assert all(v < 1e-4 for v in e.values()), e
    # ---- gradients
    ps = dict(m.named_parameters())
    for n in WATCH + WATCH_HEAD[instance]:
        assert ps[n].grad is not None, n
        # Gradients that reach their parameter only through the grounding head's loss of ONE sample (vg_proj, the head's own weights;
        # for iSAM behind a discrete Hungarian assignment and a focal loss) are dominated by WHICH bf16 rounding pattern the prompts
        # carry, not by how large it is: over three weight seeds e_hip / e_ref ranges 0.76 .. 2.02 for round 3's kernels and 0.76 .. 1.84
        # for round 4's, tensor by tensor uncorrelated between the two builds (profiles/r4_parity_seed_sweep.txt, tools/parity_seeds.sh).
        # The language-model-side gradients average over thousands of rows and stay within 0.74 .. 1.38: they keep the 1.5 / 1.8 bound.
        head_side = n.startswith(('vg_proj.', 'sam.', 'isam_model.'))
        if head_side:
            # recorded, not bounded (e_hip / e_ref 0.76 .. 2.02 over seeds: a bound wide enough to hold cannot fail) — the deterministic form
            # of this check follows below
            REPORT[f'{tag} grad {n} (recorded)'] = dict(e_ref=rel(r16['grads'][n], r32['grads'][n]), e_hip=rel(ps[n].grad.float().cpu(), r32['grads'][n]))
            continue
        _bounds(f'{tag} grad {n}', ps[n].grad.float().cpu(), r16['grads'][n], r32['grads'][n], a=1.5, b=1.8, floor=2e-4)
    # ---- head-side gradients, deterministically: the fp32 islands (vg_proj -> SAM / iSAM -> loss) are fed the ORACLE's bf16 last hidden
    # state, i.e. bit for bit what the oracle's own islands saw in its bf16-true run; their parameter gradients of the grounding loss
    # (matching, Dice / focal, box and discriminator terms included) must then agree with that run's to 1e-4 — the language model's
    # rounding pattern, which made the end-to-end comparison of these tensors a lottery, is the same on both sides.
    head_names = [n for n in WATCH + WATCH_HEAD[instance] if n.startswith(('vg_proj.', 'sam.', 'isam_model.'))]
    for n in head_names:
        ps[n].grad = None
    h16 = r16['cap']['lm'].hidden_states[-1].detach().to(dev)
    assert h16.dtype == torch.bfloat16
    ml, bx, dl = m.visual_grounding(vi['input_ids'][:, 1:], h16[:, :-1].float(), batch['grounding_image'], batch['patch_size'], [None],
                                    batch['instance_mask'])
    vg_loss, _ = m._compute_vg_loss(ml, bx, dl, batch['masks'], batch['boxes'], batch['index_offsets'])
    vg_loss.backward()
    hg = {n: rel(ps[n].grad.float().cpu(), r16['grads'][n]) for n in head_names}
    REPORT[f'{tag} head-side gradients on the oracle\'s bf16 hidden state'] = dict(hg, vg_loss=vg_loss.item(), oracle=float(r16['log']['train/vg_loss']))
    assert abs(vg_loss.item() - float(r16['log']['train/vg_loss'])) <= 1e-4 * max(abs(float(r16['log']['train/vg_loss'])), 1e-3)
    assert all(v < 1e-4 for v in hg.values()), hg
    # ... and their PARAMETER gradients on identical prompts, true width (the image encoder's blocks run the three-product
    # split-bf16 arithmetic, image_encoder.ENCODER_F32_SPLIT = 2: this is the measurement behind that default)
    head, pre = (m.isam_model, 'isam_model') if instance else (m.sam, 'sam')
    names = [n[len(pre) + 1:] for n in WATCH_HEAD[instance]] + ['image_encoder.blocks.5.mlp.linear1.weight', 'image_encoder.blocks.5.attn.out_proj.weight',
                                                               'image_encoder.norm.weight', 'image_encoder.patch_embedding.proj.weight']
    hp = dict(head.named_parameters())
    names = [n for n in names if n in hp]
    for n in names:
        hp[n].grad = None
    hsd = {f'{pre}.{k}': v.detach().float().cpu() for k, v in head.state_dict().items()}
    for n in names:
        hsd[f'{pre}.{n}'].requires_grad_(True)
    pin = [r32['cap']['prompts'][0].detach().to(dev)]
    pin_c = [r32['cap']['prompts'][0].detach().clone()]
    gim = [x.cpu() for x in batch['grounding_image']]
    if instance:
        o = head(batch['grounding_image'], batch['patch_size'], pin)
        (o.boxes[0].sum() + o.disc_logit[0].square().sum()).backward()
        _, _, bx, dl = O.isam_forward(hsd, scfg.isam, pre, gim, batch['patch_size'], pin_c)
        (bx[0].sum() + dl[0].square().sum()).backward()
    else:
        head(batch['grounding_image'], batch['patch_size'], pin)[0].square().mean().backward()
        O.sam_forward(hsd, scfg.sam, pre, gim, batch['patch_size'], pin_c)[0].square().mean().backward()
    ge = {n: rel(hp[n].grad, hsd[f'{pre}.{n}'].grad) for n in names}
    REPORT[f'{tag} fp32 island parameter gradients on identical prompts (true width)'] = ge
    assert len(ge) >= 5 and all(v < 1e-4 for v in ge.values()), ge
    for n in names:
        hp[n].grad = None


def test_depth_sweep_true_width_decoder(dev):
    """1, 2, 4, 8 true-width decoder layers (+ 2 ViT layers, vocabulary 32 008) on the configs[0] input: error of the last hidden state
    and of the logits against the fp32 oracle, HIP next to the oracle's own bf16-true evaluation"""
    from oracle import vividmed as O
    from mmmm_amd.data.synthetic import SpecialTokens, make_batch
    from mmmm_amd.models.cogvlm.configuration_cogvlm import CogVLMConfig
    from mmmm_amd.models.lora import LoraConfig
    from mmmm_amd.models.mmmm import MMMMForCausalLM, VisionArgs
    from mmmm_amd.utils import apply_lora
    DEPTHS = (1, 2, 4, 8)
    cfg = CogVLMConfig(num_hidden_layers=DEPTHS[-1])
    cfg.vision_config['num_hidden_layers'] = 2
    tok = SpecialTokens(base_vocab=32000)
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device(dev):
            m = MMMMForCausalLM.build(None, vision_override=VisionArgs(pos_embed_shape=(8, 32, 32), pt_pos_embed_shape=(35, 35), patch_size=16),
                                      tokenizer=tok, config=cfg)
    finally:
        torch.set_default_dtype(torch.float32)
    apply_lora(m, LoraConfig(r=64, lora_alpha=8, lora_dropout=0.0, use_rslora=True))
    _randomize_on_device_(m, 88)
    m.to(torch.bfloat16).eval()
    batch = make_batch([(3, 1, 224, 224)], [(1, 16, 16)], [(1, 2, 2)], [64], tok=tok, seed=7, grounding=False, device=dev)
    vi = batch['vlm_inputs']
    am = cpu(vi['attention_mask']).bool()
    # the oracle once per mode at full depth; hidden state k = what a k-layer model hands its final norm
    ocfg = oracle_cfg(m.config)
    orc = {}
    for mode, dt in (('fp32', torch.float32), ('bf16', torch.bfloat16)):
        sd = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in oracle_state(m).items()}
        cv = cpu(vi)
        with torch.no_grad():
            o = O.causal_lm_forward(sd, ocfg, cv['input_ids'], image=[x.to(dt) for x in cpu(batch['image'])], patch_size=batch['patch_size'],
                                    pool_size=batch['pool_size'], token_type_ids=cv['token_type_ids'], attention_mask=cv['attention_mask'],
                                    position_ids=cv['position_ids'], rope_dtype=torch.bfloat16)
            per = {}
            for k in DEPTHS:
                x = o.hidden_states[k] if k < DEPTHS[-1] else None
                if x is None:
                    per[k] = (o.hidden_states[-1], o.logits)
                else:
                    y = torch.zeros_like(x)
                    y[am] = O.rms_norm(x[am], O.P(sd, 'model.norm.weight'), ocfg.rms_norm_eps)
                    per[k] = (y, O.linear(sd, 'lm_head', y, ocfg).float())
        orc[mode] = per
        del sd
    layers = m.model.layers
    curve = {}
    try:
        for k in DEPTHS:
            m.model.layers = layers[:k]
            m.config.num_hidden_layers = k
            with torch.no_grad():
                out = m(**vi, image=batch['image'], patch_size=batch['patch_size'], pool_size=batch['pool_size'], return_dict=True,
                        output_hidden_states=True, materialize_logits=True)
            h32, l32 = orc['fp32'][k]
            h16, l16 = orc['bf16'][k]
            curve[k] = dict(
                hidden_e_ref=rel(h16.float()[am], h32[am]), hidden_e_hip=rel(out.hidden_states[-1].float().cpu()[am], h32[am]),
                hidden_hip_vs_bf16=rel(out.hidden_states[-1].float().cpu()[am], h16.float()[am]),
                logits_e_ref=rel(l16[am], l32[am]), logits_e_hip=rel(out.logits.cpu()[am], l32[am]),
                logits_hip_vs_bf16=rel(out.logits.cpu()[am], l16[am]))
    finally:
        m.model.layers = layers
        m.config.num_hidden_layers = DEPTHS[-1]
    REPORT['depth sweep (true-width decoder layers -> relative L2 error vs the fp32 oracle)'] = curve
    for k, c in curve.items():
        # the HIP path must track the oracle's own bf16 evaluation at every depth: as close to exact, and closer to it than to exact
        assert c['logits_e_hip'] <= 1.3 * c['logits_e_ref'] + 1e-4 and c['logits_hip_vs_bf16'] <= 1.6 * c['logits_e_ref'] + 1e-4, (k, c)
        assert c['hidden_e_hip'] <= 1.3 * c['hidden_e_ref'] + 1e-4 and c['hidden_hip_vs_bf16'] <= 1.6 * c['hidden_e_ref'] + 1e-4, (k, c)
    # growth law: fit e(depth) = e1 * depth^p on the oracle's own bf16 curve and on the HIP curve; extrapolate to 32 layers
    import numpy as np
    xs = np.log(np.array(DEPTHS, dtype=float))
    fit = {}
    for key in ('logits_e_ref', 'logits_e_hip'):
        ys = np.log(np.array([curve[k][key] for k in DEPTHS]))
        p, c0 = np.polyfit(xs, ys, 1)
        fit[key] = dict(exponent=float(p), e_at_1=float(np.exp(c0)), extrapolated_32_layers=float(np.exp(c0) * 32 ** p))
    REPORT['depth sweep fit'] = fit


def test_zz_report():
    """not a check: writes what the bounds above measured (profiles/r3_parity_report.json)"""
    print('\n' + json.dumps(REPORT, indent=1))
    out = Path(os.environ.get('GRAFT_REPO_ROOT', Path(__file__).resolve().parents[1])) / 'gpurun_out'
    if out.is_dir() and REPORT:
        (out / os.environ.get('VM_PARITY_REPORT_NAME', 'r5_parity_config0.json')).write_text(json.dumps(REPORT, indent=1))
