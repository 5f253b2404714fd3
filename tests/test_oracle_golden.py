"""Pins the CPU oracle (oracle/vividmed.py) against outputs of the REFERENCE itself.

The fixtures under tests/golden/ were produced by oracle/make_golden.py, which imports the unmodified
reference files from /root/reference (development container only) and runs them on explicit tensors.
fp32 tolerance 1e-5 relative (summation order differs); integer / boolean outputs are bit-exact."""
from pathlib import Path

import pytest
import torch

from oracle import vividmed as O
from tests import _tiny

G = Path(__file__).parent / 'golden'


def load(name):
    return torch.load(G / name, weights_only=False)


def close(a, b, tol=1e-5):
    a, b = a.double(), b.double()
    denom = b.norm().clamp_min(1e-12)
    assert ((a - b).norm() / denom).item() < tol, ((a - b).norm() / denom).item()
    assert (a - b).abs().max().item() <= tol * 50 * max(1.0, b.abs().max().item())


def test_f1_expert_mask_and_position_ids_bit_exact():
    for c in load('f1_expert_mask.pt'):
        v, l = O.get_expert_mask(c['token_type_ids'], c['attention_mask'].bool())
        assert torch.equal(v, c['vision']) and torch.equal(l, c['language'])
        assert torch.equal(O.build_position_ids(c['token_type_ids'], c['attention_mask']), c['position_ids'])


def test_f3_rmsnorm_rope_ce():
    f = load('f3_units.pt')
    close(O.rms_norm(f['rms']['x'], f['rms']['w'], 1e-6), f['rms']['y'], 1e-6)
    r = f['rope']
    cos, sin = O.rope_tables(32, int(r['pos'].max()) + 1, torch.float32)
    assert torch.equal(cos, r['cos']) and torch.equal(sin, r['sin'])
    q, k = O.apply_rope(r['q'], r['k'], cos, sin, r['pos'])
    close(q, r['q_out'], 1e-6)
    close(k, r['k_out'], 1e-6)
    c = f['ce']
    close(O.weighted_ce(c['logits'], c['labels'], c['weight']), c['loss'], 1e-6)


def test_f3_rope_table_bf16_quirk_is_reproduced():
    # under bf16-true the reference's inv_freq AND position range are bf16
    t = load('f3_units.pt')['rope_bf16_table']
    cos, sin = O.rope_tables(32, 600, torch.bfloat16)
    assert torch.equal(cos.float(), t['cos']) and torch.equal(sin.float(), t['sin'])
    cos32, _ = O.rope_tables(32, 600, torch.float32)
    assert (cos32[300:] - cos.float()[300:]).abs().max() > 0.1   # the quirk is material


def test_f4_vision_tower_and_zfold_conv():
    f = load('f4_vit.pt')
    cfg = _tiny.lm_cfg()
    feats = O.vision_forward(f['state_dict'], cfg, f['images'], f['patch_size'], f['pool_size'])
    for a, b in zip(feats, f['feats']):
        close(a, b)
    sd = f['state_dict']
    conv = O.downsample_conv(f['images'][1][None], sd['model.vision.patch_embedding.proj.weight'],
                             sd['model.vision.patch_embedding.proj.bias'], f['patch_size'][1])
    close(conv, f['conv_zfold'], 1e-6)


@pytest.fixture(scope='module')
def f5():
    return load('f5_tiny_lm.pt')


def test_f5_tiny_lm_forward(f5):
    sd = f5['state_dict']
    vi = f5['vlm_inputs']
    out = O.causal_lm_forward(sd, _tiny.lm_cfg(), vi['input_ids'], image=f5['images'], patch_size=f5['patch_size'],
                              pool_size=f5['pool_size'], token_type_ids=vi['token_type_ids'], attention_mask=vi['attention_mask'],
                              position_ids=vi['position_ids'], labels=vi['labels'], weight=vi['weight'])
    am = vi['attention_mask'].bool()
    close(out.loss, f5['loss'])
    close(out.logits[am], f5['logits'][am])
    assert len(out.hidden_states) == len(f5['hidden_states']) == 3
    for a, b in zip(out.hidden_states, f5['hidden_states']):
        close(a * am[..., None], b)


def test_f5_tiny_lm_gradients(f5):
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in f5['state_dict'].items()}
    vi = f5['vlm_inputs']
    out = O.causal_lm_forward(sd, _tiny.lm_cfg(), vi['input_ids'], image=f5['images'], patch_size=f5['patch_size'],
                              pool_size=f5['pool_size'], token_type_ids=vi['token_type_ids'], attention_mask=vi['attention_mask'],
                              position_ids=vi['position_ids'], labels=vi['labels'], weight=vi['weight'])
    out.loss.backward()
    for name, g in f5['grads'].items():
        close(sd[name].grad, g, 2e-5)


def test_f6_sam_and_instance_sam():
    f = load('f6_sam.pt')
    prompts = [p.clone().requires_grad_() for p in f['prompts']]
    masks = O.sam_forward(f['sam_state'], _tiny.sam_cfg(False), '', f['images'], f['patch_size'], prompts) \
        if False else O.sam_forward({f'sam.{k}': v for k, v in f['sam_state'].items()}, _tiny.sam_cfg(False), 'sam', f['images'], f['patch_size'], prompts)
    for a, b in zip(masks, f['sam_masks']):
        close(a, b)
    sum(m.square().mean() for m in masks).backward()
    for p, g in zip(prompts, f['sam_prompt_grads']):
        close(p.grad, g, 2e-5)
    prompts = [p.clone().requires_grad_() for p in f['prompts']]
    sd = {f'isam_model.{k}': v for k, v in f['isam_state'].items()}
    full, low, boxes, disc = O.isam_forward(sd, _tiny.sam_cfg(True), 'isam_model', f['images'], f['patch_size'], prompts)
    for a, b in zip(low, f['isam_low']):
        close(a, b)
    close(full[1], f['isam_masks_2d'])
    for a, b in zip(boxes, f['isam_boxes']):
        close(a, b)
    for a, b in zip(disc, f['isam_disc']):
        close(a, b)
    (sum(b.sum() for b in boxes) + sum(d.square().sum() for d in disc) + sum(m.mean() for m in full)).backward()
    for p, g in zip(prompts, f['isam_prompt_grads']):
        close(p.grad, g, 2e-5)


def test_f7_losses_and_hungarian_assignment():
    f = load('f7_losses.pt')
    d = f['dice_focal']
    kw = dict(dice_weight=2, focal_weight=2, focal_gamma=2)
    for key, rb, tgt in (('out', True, d['t']), ('out_nobatch', False, d['t']), ('out_none', True, None)):
        got = O.dice_focal_loss(d['x'], tgt, reduce_batch=rb, **kw)
        assert set(got) == set(d[key])
        for k in got:
            close(got[k], d[key][k], 1e-6)
    gi = f['giou']
    close(O.box_pair_giou(O.box_cs_to_cc(gi['a']), O.box_cs_to_cc(gi['b'])), gi['out'], 1e-6)
    i = f['isam']
    br, dl = i['boxes_reg'].clone().requires_grad_(), i['disc'].clone().requires_grad_()
    loss, log, match = O.isam_compute_loss(O.ISamLossCfg(), br, dl, i['boxes_label'], i['index_offsets'])
    assert torch.equal(match, i['match'])          # Hungarian assignment: bit-exact
    close(loss, i['loss'], 1e-6)
    assert set(log) == set(i['log'])
    for k in log:
        close(log[k], i['log'][k], 1e-6)
    loss.backward()
    close(br.grad, i['d_boxes'], 1e-5)
    close(dl.grad, i['d_disc'], 1e-5)


def test_f8_training_step_end_to_end():
    f = load('f8_training_step.pt')
    sd = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in f['state_dict'].items()}
    loss, log = O.training_step(sd, _tiny.step_cfg(), f['batch'])
    close(loss, f['loss'])
    assert set(log) == set(f['logged']), (sorted(log), sorted(f['logged']))
    for k, v in f['logged'].items():
        close(log[k], v)
    loss.backward()
    for name, g in f['grads'].items():
        close(sd[name].grad, g, 5e-5)


def test_f12_generation_path_matches_the_reference():
    """KV-cache decode, the <p>/</p> position rule and the single-query attention branch, against the
    reference's own prefill + teacher-forced decode steps (oracle/make_golden.py::f12_generation)."""
    f = load('f12_generation.pt')
    a = f['attn']
    close(O.decode_attention(a['q'], a['k'], a['v'], a['padding_mask']), a['out'], 1e-6)
    sd, cfg = f['state_dict'], _tiny.lm_cfg()
    for b, rec in enumerate(f['samples']):
        pre = rec['prefill']
        logits, _ = O.causal_lm_cached(sd, cfg, pre['input_ids'], token_type_ids=pre['token_type_ids'],
                                       attention_mask=pre['attention_mask'], position_ids=pre['position_ids'],
                                       image=f['images'][b:b + 1], patch_size=f['patch_size'][b:b + 1], pool_size=f['pool_size'][b:b + 1])
        close(logits, rec['prefill_logits'])
        forced = f['forced'][b:b + 1]
        seq, step_logits, pos = O.generate(sd, cfg, pre['input_ids'], token_type_ids=pre['token_type_ids'],
                                           position_ids=pre['position_ids'], image=f['images'][b:b + 1],
                                           patch_size=f['patch_size'][b:b + 1], pool_size=f['pool_size'][b:b + 1],
                                           bop_token_id=f['bop_token_id'], eop_token_id=f['eop_token_id'],
                                           max_new_tokens=forced.shape[1], forced=forced)
        assert torch.equal(seq[:, pre['input_ids'].shape[1]:], forced)
        assert torch.equal(pos, rec['final_position_ids'])                        # integer rule: bit-exact
        n0 = pre['input_ids'].shape[1]
        for t, st in enumerate(rec['steps']):
            assert int(pos[0, n0 + t]) == st['position_id']
            close(step_logits[t], st['logits'])
        # size-independent property: a cached decode step equals the last row of an uncached forward over the whole sequence
        full = O.causal_lm_forward(sd, cfg, seq, image=f['images'][b:b + 1], patch_size=f['patch_size'][b:b + 1],
                                   pool_size=f['pool_size'][b:b + 1], token_type_ids=torch.cat([pre['token_type_ids'], torch.zeros_like(forced)], 1),
                                   attention_mask=torch.ones_like(seq), position_ids=pos)
        close(full.logits[:, -1], step_logits[-1], 2e-5)
