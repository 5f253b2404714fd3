"""TEST INFRASTRUCTURE — generates tests/golden/*.pt by running the REFERENCE (imported from
/root/reference through oracle/ref_shims.py) on explicit tensors. Runs only in the development container:

    python -m oracle.make_golden

Fixtures hold inputs, explicit weights and the reference's outputs (data only, no reference source).
"""
from __future__ import annotations

import math
import os
from pathlib import Path
import types

import torch

from . import ref_shims

OUT = Path(__file__).resolve().parents[1] / 'tests' / 'golden'


import contextlib


@contextlib.contextmanager
def zero_filled_empty():
    """The reference leaves the padded rows of several activations UNINITIALISED (`torch.empty` at modeling_cogvlm.py:277 and in
    attention_fn; rows outside the padding mask are never written). Their content never reaches a valid row, but 0 x garbage
    does reach weight gradients (a NaN bit pattern in a padded row of the last hidden state turns the whole lm_head gradient
    into NaN: seen while generating f13). Inside this context `torch.empty*` returns zeros, which is one legal outcome of
    "uninitialised" and makes the fixtures reproducible."""
    e, el = torch.empty, torch.empty_like
    torch.empty = lambda *a, **k: e(*a, **k).zero_()
    torch.empty_like = lambda *a, **k: el(*a, **k).zero_()
    try:
        yield
    finally:
        torch.empty, torch.empty_like = e, el


def randomize_(module: torch.nn.Module, seed: int):
    """explicit, non-degenerate values for every parameter/buffer (norm weights near 1, nothing at zero)"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            leaf = name.rsplit('.', 1)[-1]
            if p.ndim == 1 and ('norm' in name or 'layernorm' in name or name.endswith('output_upscaling.1.weight')) and leaf == 'weight':
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g))
            elif leaf == 'bias':
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
            else:
                fan_in = p.shape[1] if p.ndim >= 2 else p.shape[0]
                if p.ndim > 2:
                    fan_in = math.prod(p.shape[1:])
                std = 0.5 if p.ndim < 2 or 'embed' in name or name.endswith(('boi', 'eoi')) else 1.0 / math.sqrt(fan_in)
                p.copy_(std * torch.randn(p.shape, generator=g))
        for name, b in module.named_buffers():
            if 'positional_encoding_gaussian_matrix' in name:
                b.copy_(torch.randn(b.shape, generator=g))


def tiny_cfg(R, n_lm=2, n_vit=2):
    cfg = R.config.CogVLMConfig(
        vocab_size=160, hidden_size=64, intermediate_size=128, num_hidden_layers=n_lm, num_attention_heads=2,
        vision_config=dict(in_channels=3, hidden_size=32, patch_size=(4, 8, 8), num_heads=2, num_hidden_layers=n_vit,
                           intermediate_size=64, hidden_act='gelu', dropout_prob=0.0, layer_norm_eps=1e-6,
                           pos_embed_shape=(2, 2, 4), pt_pos_embed_shape=(2, 4)),
    )
    va = R.mmmm.VisionArgs(pos_embed_shape=(2, 2, 4), pt_pos_embed_shape=(2, 4), patch_size=(4, 8, 8))
    return cfg, va


def make_vlm_inputs(samples, L, g):
    """samples: list of (n_image_tokens Np, n_text T). Returns right-padded reference-style inputs
    (layout of mmmm/data/utils.py:104-140)."""
    B = len(samples)
    ids = torch.zeros(B, L, dtype=torch.long)
    tt = torch.zeros(B, L, dtype=torch.long)
    pos = torch.zeros(B, L, dtype=torch.long)
    am = torch.zeros(B, L, dtype=torch.long)
    labels = torch.full((B, L), -100, dtype=torch.long)
    weight = torch.zeros(B, L)
    BOP, EOP, GRD = 150, 151, 152
    for b, (Np, T) in enumerate(samples):
        n = 1 + (Np + 2) + 1 + T
        text = torch.randint(3, 140, (T,), generator=g)
        # two <p> ... </p> pairs
        text[3], text[6], text[10], text[14] = BOP, EOP, BOP, EOP
        ids[b, :n] = torch.cat([torch.tensor([1]), torch.zeros(Np + 2, dtype=torch.long), torch.tensor([GRD]), text])
        tt[b, 1:1 + Np + 2] = 1
        tp = torch.empty(T, dtype=torch.long)
        tp[0] = 5
        for i in range(1, T):
            tp[i] = tp[i - 1] if (text[i - 1] == BOP or text[i] == EOP) else tp[i - 1] + 1
        pos[b, :n] = torch.cat([torch.tensor([0, 1]), torch.full((Np,), 2), torch.tensor([3, 4]), tp])
        am[b, :n] = 1
        lab = torch.cat([text[1:], torch.tensor([2])])
        labels[b, n - T:n] = lab
        weight[b, n - T:n] = 1.0
        weight[b, n - T:n][lab == BOP] = 5.0
    return dict(input_ids=ids, token_type_ids=tt, position_ids=pos, attention_mask=am, labels=labels, weight=weight)


def f1_masks(R):
    g = torch.Generator().manual_seed(1)
    cases = []
    pats = [
        ([0, 1, 1, 1, 1, 0, 0, 0, 0, 0], [1] * 10),
        ([0, 1, 1, 1, 0, 0, 0, 0, 0, 0], [1, 1, 1, 1, 1, 1, 1, 0, 0, 0]),
        ([1, 1, 1, 0, 0, 1, 1, 0, 1, 1], [1] * 10),
        ([0, 0, 0, 0, 0, 0, 0, 0, 0, 0], [1, 1, 1, 1, 0, 0, 0, 0, 0, 0]),
        ([1], [1]),
    ]
    for tt, am in pats:
        tt_t, am_t = torch.tensor([tt]), torch.tensor([am])
        v, l = R.modeling_cogvlm.get_expert_mask(tt_t, am_t.bool())
        p = R.modeling_cogvlm.build_position_ids(tt_t, am_t)
        cases.append(dict(token_type_ids=tt_t, attention_mask=am_t, vision=v, language=l, position_ids=p))
    tt = torch.randint(0, 2, (4, 33), generator=g)
    am = torch.ones(4, 33, dtype=torch.long)
    am[1, 20:] = 0
    am[3, 5:] = 0
    v, l = R.modeling_cogvlm.get_expert_mask(tt, am.bool())
    cases.append(dict(token_type_ids=tt, attention_mask=am, vision=v, language=l, position_ids=R.modeling_cogvlm.build_position_ids(tt, am)))
    torch.save(cases, OUT / 'f1_expert_mask.pt')


def f3_units(R):
    g = torch.Generator().manual_seed(3)
    mc = R.modeling_cogvlm
    x = torch.randn(5, 64, generator=g)
    norm = mc.RMSNorm(64, eps=1e-6)
    with torch.no_grad():
        norm.weight.copy_(1 + 0.1 * torch.randn(64, generator=g))
    rot = mc.RotaryEmbedding(32)
    q, k = torch.randn(2, 2, 7, 32, generator=g), torch.randn(2, 2, 7, 32, generator=g)
    pos = torch.tensor([[0, 1, 2, 2, 2, 3, 4], [0, 1, 2, 3, 3, 4, 9]])
    cos, sin = rot(q, seq_len=int(pos.max()) + 1)
    qr, kr = mc.apply_rotary_pos_emb_index_bhs(q, k, cos, sin, pos)
    logits = torch.randn(3, 6, 50, generator=g)
    labels = torch.randint(0, 50, (3, 6), generator=g)
    labels[0, :2] = -100
    w = torch.rand(3, 6, generator=g) * 5
    # bf16 rope table quirk: module converted to bf16 -> inv_freq bf16
    rot16 = mc.RotaryEmbedding(32).to(torch.bfloat16)
    c16, s16 = rot16(q.bfloat16(), seq_len=600)
    torch.save(dict(
        rms=dict(x=x, w=norm.weight.detach().clone(), y=norm(x).detach()),
        rope=dict(q=q, k=k, pos=pos, cos=cos[:, 0].clone(), sin=sin[:, 0].clone(), q_out=qr, k_out=kr),
        rope_bf16_table=dict(cos=c16[:, 0].float().clone(), sin=s16[:, 0].float().clone()),
        ce=dict(logits=logits, labels=labels, weight=w, loss=mc._sample_weighted_ce(logits, labels, w)),
    ), OUT / 'f3_units.pt')


def build_tiny_model(R, with_sam: bool, seed: int, n_lm=2, n_vit=2):
    cfg, va = tiny_cfg(R, n_lm, n_vit)
    m = R.mmmm.MMMMForCausalLM(cfg, vision_override=va)
    tok = types.SimpleNamespace(bop_token_id=150, eop_token_id=151)
    m.tokenizer = tok
    m.lm_loss_weight = 1.0
    m.sam = None
    if with_sam:
        sam = R.build_sam._build_sam(embed_dim=32, encoder_num_layers=2, num_heads=2, patch_size=(4, 8, 8), pos_embed_shape=(2, 2, 4))
        isam = R.build_sam._build_instance_sam(embed_dim=32, encoder_num_layers=2, num_heads=2, patch_size=(4, 8, 8),
                                               pos_embed_shape=(2, 2, 4), num_instances=6)
        m.sam, m.isam_model = sam, isam
        m.mask_loss = R.loss.DiceFocalLoss(dice_weight=2, focal_weight=2, focal_gamma=2)
        m.isam_loss = R.sam.InstanceSamLoss(use_neg_mask=False, box_l1_weight=5, box_giou_weight=2, disc_weight=2,
                                            disc_focal_gamma=2, disc_focal_alpha=0.85)
        m.isam_loss.mask_loss = m.mask_loss
        m.vg_proj = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.ReLU(inplace=True), torch.nn.Linear(64, 32))
    m.trainer = types.SimpleNamespace(is_parallel=False)
    randomize_(m, seed)
    return m, cfg


def f5_tiny_lm(R):
    """tiny CogVLM (2+2 layers) forward/backward on a mixed 2D/3D right-padded batch"""
    g = torch.Generator().manual_seed(5)
    m, cfg = build_tiny_model(R, False, seed=50)
    m.train()
    images = [torch.randn(3, 1, 16, 32, generator=g), torch.randn(3, 8, 16, 32, generator=g), torch.randn(3, 4, 32, 16, generator=g)]
    patch = [(1, 8, 8), (4, 8, 8), (2, 8, 8)]
    pool = [(1, 2, 2), (2, 2, 2), (1, 1, 1)]
    # tokens after pooling: (1,2,4)->(1,1,2)=2 ; (2,2,4)->(1,1,2)=2 ; (2,4,2)->16
    vi = make_vlm_inputs([(2, 20), (2, 17), (16, 22)], L=44, g=g)
    out = m(**vi, image=images, patch_size=patch, pool_size=pool, return_dict=True, output_hidden_states=True)
    out.loss.backward()
    names = ['model.layers.0.self_attn.vision_expert_query_key_value.weight', 'model.layers.1.mlp.language_mlp.down_proj.weight',
             'model.layers.0.input_layernorm.weight', 'model.embed_tokens.weight', 'lm_head.weight',
             'model.vision.transformer.layers.0.attention.query_key_value.weight', 'model.vision.patch_embedding.proj.weight',
             'model.vision.patch_embedding.position_embedding.weight', 'model.vision.linear_proj.gate_proj.weight',
             'model.vision.transformer.layers.1.post_attention_layernorm.weight', 'model.vision.patch_embedding.cls_embedding.weight']
    params = dict(m.named_parameters())
    am = vi['attention_mask'].bool()
    torch.save(dict(
        state_dict={k: v.detach().clone() for k, v in m.state_dict().items()},
        images=images, patch_size=patch, pool_size=pool, vlm_inputs=vi,
        logits=out.logits.detach(), loss=out.loss.detach(),
        hidden_states=[h.detach() * am[..., None] for h in out.hidden_states],   # padded rows are undefined in the reference
        grads={n: params[n].grad.clone() for n in names},
        meta=dict(note='sample 0 and 2 resample the position embedding (UNPINNED luolib semantics, shim = trilinear); sample 1 is the identity case'),
    ), OUT / 'f5_tiny_lm.pt')


def f4_vit_identity(R):
    """vision tower alone on an identity-resample input (pins everything except luolib.resample) + z-fold conv"""
    g = torch.Generator().manual_seed(4)
    m, cfg = build_tiny_model(R, False, seed=40)
    m.eval()
    images = [torch.randn(3, 8, 16, 32, generator=g), torch.randn(3, 4, 16, 32, generator=g)]
    patch = [(4, 8, 8), (2, 8, 8)]        # second: kernel z 4 folded to 2 (resample.py:55-62)
    pool = [(2, 2, 2), (1, 2, 2)]
    with torch.no_grad():
        feats = m.model.vision(images, patch, pool)
        conv = m.model.vision.patch_embedding.proj(images[1][None], patch[1])
    torch.save(dict(state_dict={k: v.detach().clone() for k, v in m.state_dict().items() if k.startswith('model.vision')},
                    images=images, patch_size=patch, pool_size=pool, feats=[f[0] for f in feats], conv_zfold=conv), OUT / 'f4_vit.pt')


def f6_sam(R):
    g = torch.Generator().manual_seed(6)
    sam = R.build_sam._build_sam(embed_dim=32, encoder_num_layers=2, num_heads=2, patch_size=(4, 8, 8), pos_embed_shape=(2, 2, 4))
    isam = R.build_sam._build_instance_sam(embed_dim=32, encoder_num_layers=2, num_heads=2, patch_size=(4, 8, 8),
                                           pos_embed_shape=(2, 2, 4), num_instances=6)
    randomize_(sam, 60)
    randomize_(isam, 61)
    images = [torch.rand(3, 8, 16, 32, generator=g), torch.rand(3, 1, 16, 32, generator=g), torch.rand(3, 4, 32, 16, generator=g)]
    patch = [(4, 8, 8), (1, 8, 8), (2, 8, 8)]
    prompts = [torch.randn(1, 32, generator=g).requires_grad_(), torch.randn(3, 32, generator=g).requires_grad_(),
               torch.randn(2, 32, generator=g).requires_grad_()]
    masks = sam(images, patch, prompts)
    sum(mk.square().mean() for mk in masks).backward()
    gp = [p.grad.clone() for p in prompts]
    for p in prompts:
        p.grad = None
    out = isam(images, patch, prompts)
    (sum(b.sum() for b in out.boxes) + sum(d.square().sum() for d in out.disc_logit) + sum(mm.mean() for mm in out.masks_logits)).backward()
    torch.save(dict(
        sam_state={k: v.detach().clone() for k, v in sam.state_dict().items()},
        isam_state={k: v.detach().clone() for k, v in isam.state_dict().items()},
        images=images, patch_size=patch, prompts=[p.detach().clone() for p in prompts],
        sam_masks=[m_.detach() for m_ in masks], sam_prompt_grads=gp,
        isam_masks_2d=out.masks_logits[1].detach(), isam_low=[m_.detach() for m_ in out.masks_logits_low_res],
        isam_boxes=[b.detach() for b in out.boxes], isam_disc=[d.detach() for d in out.disc_logit],
        isam_prompt_grads=[p.grad.clone() for p in prompts],
        meta=dict(note='image 0 is the identity-resample case; images 1,2 resample position embeddings (UNPINNED)'),
    ), OUT / 'f6_sam.pt')


def f7_losses(R):
    g = torch.Generator().manual_seed(7)
    dfl = R.loss.DiceFocalLoss(dice_weight=2, focal_weight=2, focal_gamma=2)
    x = torch.randn(3, 1, 4, 16, 16, generator=g) * 2
    t = torch.rand(3, 1, 4, 16, 16, generator=g) < 0.2
    d = dfl(x, t, return_dict=True)
    d_nb = dfl(x, t, reduce_batch=False, return_dict=True)
    d_none = dfl(x, None, return_dict=True)
    il = R.sam.InstanceSamLoss(use_neg_mask=False, box_l1_weight=5, box_giou_weight=2, disc_weight=2, disc_focal_gamma=2,
                               disc_focal_alpha=0.85)
    il.mask_loss = dfl
    nt, M = 3, 7
    boxes_reg = torch.rand(nt, M, 6, generator=g) * 0.5 + 0.2
    disc = torch.randn(nt, M - 1, generator=g)
    boxes_label = torch.rand(9, 6, generator=g) * 0.4 + 0.3
    index_offsets = torch.tensor([[0, 2], [2, 2], [2, 9]])     # 2 instances, none, 7 (> num queries)
    dummy = boxes_reg.new_empty((nt, M, 0, 0, 0))
    br = boxes_reg.clone().requires_grad_()
    dl = disc.clone().requires_grad_()
    loss, log = il.compute_loss(dummy, dummy, br, dl, None, boxes_label, index_offsets)
    loss.backward()
    # recover the assignment the reference used
    matches = []
    for i in range(nt):
        s, e = index_offsets[i].tolist()
        mt = il._match_instances(dummy[i, 1:, None], boxes_reg[i, 1:], disc[i].float(), None, boxes_label[s:e], 0, int(index_offsets[i, 0]))
        matches.append(mt if torch.is_tensor(mt) else torch.full((M - 1,), mt))
    torch.save(dict(
        dice_focal=dict(x=x, t=t, out={k: v.detach() for k, v in d.items()}, out_nobatch={k: v.detach() for k, v in d_nb.items()},
                        out_none={k: v.detach() for k, v in d_none.items()}),
        isam=dict(boxes_reg=boxes_reg, disc=disc, boxes_label=boxes_label, index_offsets=index_offsets, loss=loss.detach(),
                  log={k: v.detach() for k, v in log.items()}, match=torch.stack(matches), d_boxes=br.grad.clone(), d_disc=dl.grad.clone()),
        giou=dict(a=boxes_reg[0], b=boxes_label[:7],
                  out=ref_shims.box_pair_giou(ref_shims.convert_box_mode(boxes_reg[0], src_mode=ref_shims.CenterSizeMode),
                                              ref_shims.convert_box_mode(boxes_label[:7], src_mode=ref_shims.CenterSizeMode))),
    ), OUT / 'f7_losses.pt')


def f8_training_step(R):
    """MMMMForCausalLM.training_step with sam (mask loss) and isam (box/disc loss + Hungarian) branches live"""
    g = torch.Generator().manual_seed(8)
    m, cfg = build_tiny_model(R, True, seed=80)
    m.train()
    images = [torch.randn(3, 1, 16, 32, generator=g), torch.randn(3, 8, 16, 32, generator=g)]
    gimages = [torch.rand(3, 1, 16, 32, generator=g), torch.rand(3, 8, 16, 32, generator=g)]
    patch = [(1, 8, 8), (4, 8, 8)]
    pool = [(1, 2, 2), (2, 2, 2)]
    vi = make_vlm_inputs([(2, 20), (2, 18)], L=28, g=g)
    masks = [torch.rand(2, 1, 16, 32, generator=g) < 0.15, None]
    boxes = [None, torch.rand(5, 6, generator=g) * 0.4 + 0.3]
    index_offsets = [None, torch.tensor([[0, 2], [2, 5]])]
    batch = dict(vlm_inputs=vi, image=images, grounding_image=gimages, patch_size=patch, pool_size=pool, masks=masks, boxes=boxes,
                 index_offsets=index_offsets, instance_mask=[False, True], vg_label_mask=[None, None])
    loss = m.training_step(batch)
    loss.backward()
    params = dict(m.named_parameters())
    names = ['vg_proj.0.weight', 'vg_proj.2.bias', 'model.layers.1.self_attn.language_expert_dense.weight', 'model.norm.weight',
             'sam.mask_decoder.transformer.layers.0.cross_attn_token_to_image.q_proj.weight', 'sam.image_encoder.blocks.0.attn.qkv.weight',
             'isam_model.box_head.4.weight', 'isam_model.mask_decoder.mask_tokens.weight', 'lm_head.weight']
    torch.save(dict(
        state_dict={k: v.detach().clone() for k, v in m.state_dict().items()},
        batch=batch, loss=loss.detach(), logged={k: (v.detach() if torch.is_tensor(v) else v) for k, v in m._logged.items()},
        grads={n: params[n].grad.clone() for n in names if params[n].grad is not None},
    ), OUT / 'f8_training_step.pt')


def f9_lora_targets(R):
    """LoRA target / modules_to_save discovery (mmmm/utils.py:19-61, mmmm.py:157-165) at 1+1 layers"""
    res = {}
    for freeze in (True, False):
        m, cfg = build_tiny_model(R, True, seed=90, n_lm=1, n_vit=1)
        if freeze:
            m.sam.requires_grad_(False)
            m.isam_model.requires_grad_(False)
        m._freeze_sam_unused()
        m.model.config.lora_lang = True
        target, save = m.get_lora_modules(prefix='')
        res[f'freeze_sam={freeze}'] = dict(target_modules=target, modules_to_save=save)
    torch.save(res, OUT / 'f9_lora_targets.pt')


def f10_state_dict_adapters(R):
    """load-time adapters: visual.py:37-57, resample.py:31-53, image_encoder.py:82-119, mask_decoder.py:76-87,
    on synthetic checkpoints in identity-resample geometry (luolib.resample is unpinned)"""
    import sys
    from argparse import Namespace
    g = torch.Generator().manual_seed(10)
    res = {}
    # 1. CogVLM patch embedding: 2-D position table + 2-D conv kernel + unwrapped parameter names
    cfg = Namespace(in_channels=3, hidden_size=8, patch_size=(4, 2, 2), pos_embed_shape=(2, 3, 4), pt_pos_embed_shape=(3, 4))
    pe = R.visual.PatchEmbedding(cfg)
    ckpt = {'position_embedding.weight': torch.randn(1 + 12, 8, generator=g), 'proj.weight': torch.randn(8, 3, 2, 2, generator=g),
            'proj.bias': torch.randn(8, generator=g), 'cls_embedding': torch.randn(1, 8, generator=g)}
    pe.load_state_dict({k: v.clone() for k, v in ckpt.items()}, strict=False)
    res['patch_embedding'] = dict(cfg=vars(cfg), ckpt=ckpt, loaded={k: v.detach().clone() for k, v in pe.state_dict().items()})
    # 2. Downsample 'center' inflation, odd and even depth
    for kz in (3, 4):
        ds = R.resample.Downsample(2, 5, (kz, 2, 2), inflation='center')
        w2 = torch.randn(5, 2, 2, 2, generator=g)
        ds.load_state_dict({'weight': w2.clone(), 'bias': torch.zeros(5)})
        res[f'downsample_center_{kz}'] = dict(weight_2d=w2, loaded=ds.weight.detach().clone())
    # 3. SAM patch embedding block from a SegVol-style checkpoint (Linear over flattened patches, flat position table)
    ie = sys.modules['mmmm.models.segvol.modeling.image_encoder']
    blk = ie.PatchEmbeddingBlock(in_channels=3, patch_size=(2, 4, 4), pos_embed_shape=(2, 2, 3), hidden_size=8, num_heads=2, dropout_rate=0.0,
                                 pt_in_channels=1, pt_patch_size=(2, 4, 4), pt_pos_embed_shape=(2, 2, 3))
    ckpt = {'patch_embeddings.1.weight': torch.randn(8, 2 * 4 * 4 * 1, generator=g), 'patch_embeddings.1.bias': torch.randn(8, generator=g),
            'position_embeddings': torch.randn(1, 12, 8, generator=g)}
    blk.load_state_dict({k: v.clone() for k, v in ckpt.items()}, strict=False)
    res['sam_patch_embedding'] = dict(ckpt=ckpt, loaded={k: v.detach().clone() for k, v in blk.state_dict().items()})
    # 4. mask decoder: per-voxel LayerNorm affine + shorter mask-token table
    md_mod = sys.modules['mmmm.models.segvol.modeling.mask_decoder']
    tr = sys.modules['mmmm.models.segvol.modeling.transformer']
    torch.manual_seed(1010)          # the embedding tables' own init is part of the fixture (init_tokens): make it reproducible
    dec = md_mod.MaskDecoder(transformer_dim=16, transformer=tr.TwoWayTransformer(depth=1, embedding_dim=16, mlp_dim=32, num_heads=2),
                             num_instances=6)
    init_tokens = dec.mask_tokens.weight.detach().clone()
    ckpt = {'output_upscaling.1.weight': torch.randn(4, 2, 3, 3, generator=g), 'output_upscaling.1.bias': torch.randn(4, 2, 3, 3, generator=g),
            'mask_tokens.weight': torch.randn(4, 16, generator=g)}
    dec.load_state_dict({k: v.clone() for k, v in ckpt.items()}, strict=False)
    res['mask_decoder'] = dict(ckpt=ckpt, init_tokens=init_tokens, ln_weight=dec.output_upscaling[1].weight.detach().clone(),
                               ln_bias=dec.output_upscaling[1].bias.detach().clone(), mask_tokens=dec.mask_tokens.weight.detach().clone())
    torch.save(res, OUT / 'f10_adapters.pt')


F11_CASES = [
    dict(conv=[('Where is the <p>liver</p>?', 'The <p>liver</p> is here, no <np>tumor</np> found.')], n_img=49, inference=False, grounding=True,
         bop_weight=5.0, max_seq_len=None),
    dict(conv=[('Describe the image.', 'A chest radiograph with <p>cardiomegaly</p> and <p>pleural effusion</p>.'),
               ('Is there <np>pneumothorax</np>?', 'No <np>pneumothorax</np>.')], n_img=196, inference=False, grounding=True, bop_weight=2.5,
         max_seq_len=None),
    dict(conv=[('What modality is this?', 'CT.')], n_img=16, inference=False, grounding=False, bop_weight=1.0, max_seq_len=None),
    dict(conv=[('Find <p>the left kidney</p> and <p>the spleen</p> please', 'ok <p>the left kidney</p></p> done')], n_img=9, inference=False,
         grounding=True, bop_weight=5.0, max_seq_len=24),
    dict(conv=[('First question', 'first <p>answer</p>'), ('Second question about <p>the lesion</p>', '')], n_img=49, inference=True,
         grounding=True, bop_weight=None, max_seq_len=None),
]


def f11_vlm_inputs(R):
    """prepare_vlm_inputs / get_text_position_ids / _collate_fn padding (mmmm/data/utils.py:20-145, datamodule.py:20-39) driven
    by oracle/fake_tokenizer.py"""
    from .fake_tokenizer import FakeTokenizer
    from torch.nn.utils.rnn import pad_sequence
    U = ref_shims.load_data_utils()
    import sys
    ConvTurn = sys.modules['mmmm.data.defs'].ConvTurn
    tok = FakeTokenizer()
    outs = []
    for c in F11_CASES:
        inputs, text = U.prepare_vlm_inputs([ConvTurn(q, a) for q, a in c['conv']], tok, c['n_img'], inference=c['inference'],
                                            grounding=c['grounding'], max_seq_len=c['max_seq_len'], bop_weight=c['bop_weight'])
        outs.append(dict(inputs={k: v.clone() for k, v in inputs.items()}, text=text))
    train = [o['inputs'] for o, c in zip(outs, F11_CASES) if not c['inference']]
    padded = {k: pad_sequence([x[k] for x in train], batch_first=True, padding_value=-100 if k == 'labels' else 0) for k in train[0]}
    torch.save(dict(cases=F11_CASES, outputs=outs, collated=padded), OUT / 'f11_vlm_inputs.pt')


def f12_generation(R):
    """generation path: prefill with use_cache, then teacher-forced single-token decode steps through the
    reference's own `prepare_inputs_for_generation` (mmmm.py:368-406, the <p>/</p> position rule) and its KV-cache
    attention (modeling_cogvlm.py:129-141, 253-262). The two kwargs
    updates that live in transformers' GenerationMixin (absent from the transformers version of this image) are restated
    here: attention_mask gets a column of ones, past_key_values is taken from the output; the reference's own parts
    (token_type_ids += LANGUAGE, modeling_cogvlm.py:764-780; position_ids += last+1, mmmm.py:354-366) follow its code."""
    g = torch.Generator().manual_seed(12)
    m, cfg = build_tiny_model(R, False, seed=120)
    m.eval()
    images = [torch.randn(3, 1, 16, 32, generator=g), torch.randn(3, 8, 16, 32, generator=g), torch.randn(3, 4, 32, 16, generator=g)]
    patch = [(1, 8, 8), (4, 8, 8), (2, 8, 8)]
    pool = [(1, 2, 2), (2, 2, 2), (1, 1, 1)]
    right = make_vlm_inputs([(2, 20), (2, 17), (16, 16)], L=44, g=g)
    keys = ('input_ids', 'token_type_ids', 'position_ids', 'attention_mask')
    BOP, EOP = m.tokenizer.bop_token_id, m.tokenizer.eop_token_id
    forced = torch.tensor([[17, BOP, 33, 34, EOP, 9, 71],
                           [BOP, 5, EOP, 88, 89, BOP, 90],
                           [40, 41, 42, BOP, 43, EOP, 44]])
    n_steps = forced.shape[1]
    # The reference scatters image features at columns [1, 1+n) of every row (modeling_cogvlm.py:455-466), i.e. it assumes
    # the prompt starts at column 0: a left-padded batch (what HF generate wants) cannot carry images. Each sample is
    # therefore prefetched and decoded alone (batch of 1, no padding) -- what `_inference_path` (mmmm.py:408-424) does.
    per_sample = []
    with torch.no_grad():
        for b in range(3):
            n = int(right['attention_mask'][b].sum())
            kw = {k: right[k][b:b + 1, :n].clone() for k in keys}
            out = m(**kw, image=images[b:b + 1], patch_size=patch[b:b + 1], pool_size=pool[b:b + 1], use_cache=True, return_dict=True)
            rec = dict(prefill=dict(**{k: v.clone() for k, v in kw.items()}), prefill_logits=out.logits.clone(), steps=[])
            input_ids, token_type_ids, position_ids, attention_mask = (kw[k] for k in keys)
            past = out.past_key_values
            for t in range(n_steps):
                # --- what generate() does between steps
                input_ids = torch.cat([input_ids, forced[b:b + 1, t:t + 1]], dim=1)
                attention_mask = torch.cat([attention_mask, torch.ones(1, 1, dtype=torch.long)], dim=1)              # GenerationMixin
                token_type_ids = torch.cat([token_type_ids, torch.zeros(1, 1, dtype=torch.long)], dim=1)              # :764-780
                position_ids = torch.cat([position_ids, position_ids[:, -1:] + 1], dim=1)                             # mmmm.py:354-366
                inputs = m.prepare_inputs_for_generation(input_ids, token_type_ids=token_type_ids, position_ids=position_ids,
                                                         image=None, past_key_values=past, attention_mask=attention_mask,
                                                         patch_size=None, pool_size=None, use_cache=True)
                out = m(**inputs, return_dict=True)
                past = out.past_key_values
                rec['steps'].append(dict(token=int(forced[b, t]), position_id=int(inputs['position_ids'][0, -1]),
                                         logits=out.logits[:, -1].clone()))
            rec['final_position_ids'] = position_ids.clone()
            per_sample.append(rec)
    # the decode branch of attention_fn on its own, batched with padding inside the cache (modeling_cogvlm.py:129-141)
    Bq, H, Lk, hd = 3, 2, 11, 32
    q = torch.randn(Bq, H, 1, hd, generator=g)
    k = torch.randn(Bq, H, Lk, hd, generator=g)
    v = torch.randn(Bq, H, Lk, hd, generator=g)
    pm = torch.ones(Bq, Lk, dtype=torch.bool)
    pm[0, :4] = False
    pm[2, 2:5] = False
    attn_out = R.modeling_cogvlm.attention_fn(q.clone(), k.clone(), v.clone(), pm)
    torch.save(dict(
        state_dict={kk: vv.detach().clone() for kk, vv in m.state_dict().items()},
        images=images, patch_size=patch, pool_size=pool, forced=forced, samples=per_sample,
        bop_token_id=BOP, eop_token_id=EOP,
        attn=dict(q=q, k=k, v=v, padding_mask=pm, out=attn_out),
    ), OUT / 'f12_generation.pt')


# ------------------------------------------------------------------------------------------------------------------
# bf16-true fixtures: the reference's own training precision (conf/phase-vg/fit.yaml:7 `precision: bf16-true`,
# mmmm.py:468-492 MyPrecision) run on the CPU. They separate "rounds where the reference rounds" (testable, tight) from
# "differs from an fp32 evaluation by bf16 rounding" (physics), and give the bf16 tolerance of the GPU tests its evidence.
def f13_bf16_true(R):
    g = torch.Generator().manual_seed(13)
    res = {}
    # (a) tiny CogVLM of F5 (same weights, same inputs) converted with .bfloat16()
    f5 = torch.load(OUT / 'f5_tiny_lm.pt', weights_only=False)
    m, cfg = build_tiny_model(R, False, seed=50)
    m.load_state_dict(f5['state_dict'])
    m.train()
    m = m.bfloat16()
    images = [im.bfloat16() for im in f5['images']]
    vi = dict(f5['vlm_inputs'])
    vi['weight'] = vi['weight'].bfloat16()          # HalfPrecision.convert_input casts every floating-point input
    out = m(**vi, image=images, patch_size=f5['patch_size'], pool_size=f5['pool_size'], return_dict=True, output_hidden_states=True)
    out.loss.backward()
    params = dict(m.named_parameters())
    am = vi['attention_mask'].bool()
    res['tiny_lm'] = dict(
        logits=out.logits.detach(), loss=out.loss.detach(),
        hidden_states=[torch.where(am[..., None], h.detach(), torch.zeros((), dtype=h.dtype)) for h in out.hidden_states],   # padded rows: uninitialised
        grads={n: params[n].grad.clone() for n in f5['grads']},
        note='weights / inputs = f5_tiny_lm.pt rounded to bf16; logits are fp32 (the reference casts them, modeling_cogvlm.py:701)')
    # (b) training_step of F8 under MyPrecision (sam / isam_model / vg_proj stay fp32; grounding_image / boxes stay fp32)
    f8 = torch.load(OUT / 'f8_training_step.pt', weights_only=False)
    m, cfg = build_tiny_model(R, True, seed=80)
    m.load_state_dict(f8['state_dict'])
    m.train()
    prec = R.mmmm.MyPrecision()
    m = prec.convert_module(m)
    assert m.sam.image_encoder.blocks[0].attn.qkv.weight.dtype == torch.float32 and m.lm_head.weight.dtype == torch.bfloat16
    batch = prec.convert_input(f8['batch'])
    assert batch['grounding_image'][0].dtype == torch.float32 and batch['image'][0].dtype == torch.bfloat16
    loss = m.training_step(batch)
    loss.backward()
    params = dict(m.named_parameters())
    res['training_step'] = dict(loss=loss.detach(), logged={k: (v.detach() if torch.is_tensor(v) else v) for k, v in m._logged.items()},
                                grads={n: params[n].grad.clone() for n in f8['grads'] if params[n].grad is not None})
    torch.save(res, OUT / 'f13_bf16_true.pt')


def spec_tensor(name: str, shape, scale: float, dtype=torch.float32) -> torch.Tensor:
    """Seed-free weights for the true-width fixtures: element i of tensor `name` is a closed-form function of (name, i), so the
    400 M parameters of a true-width layer need not be stored: u = frac(sin(i * 12.9898 + salt) * 43758.5453) in float64,
    value = (2u - 1) * scale * sqrt(3) (variance scale^2). Restated identically in tests/_tiny.py:spec_tensor."""
    import zlib
    n = math.prod(shape)
    salt = float(zlib.crc32(name.encode()) % 10007)
    i = torch.arange(n, dtype=torch.float64)
    u = torch.sin(i * 12.9898 + salt * 78.233) * 43758.5453
    u = u - torch.floor(u)
    return ((2 * u - 1) * (scale * math.sqrt(3.0))).to(dtype).reshape(shape)


def grad_rows(g: torch.Tensor) -> torch.Tensor:
    """the part of a weight gradient a fixture keeps: the first 8 rows of a matrix, all of a vector"""
    g = g.float()
    return (g if g.ndim == 1 else g.reshape(g.shape[0], -1)[:8]).clone()


def f2_true_width(R):
    """ONE decoder layer, ONE ViT layer at the true widths (h=4096 / i=11008 / 32 heads of 128; d=1792 / f=15360 / 16
    heads of 112) and the 768-wide two-way transformer block of SAM (F6), run by the reference in fp32 AND in bf16. Weights come from
    `spec_tensor` (the fixture stores the spec + checksums, inputs, outputs, the input gradient and slices of weight gradients)."""
    mc, vis = R.modeling_cogvlm, R.visual
    import sys
    tr = sys.modules['mmmm.models.segvol.modeling.transformer']
    res = {}
    g = torch.Generator().manual_seed(2)

    def fill(module, prefix, scales):
        sums = {}
        with torch.no_grad():
            for n, p_ in module.named_parameters():
                leaf = n.rsplit('.', 1)[-1]
                if leaf == 'weight' and p_.ndim == 1:
                    t = 1 + spec_tensor(prefix + n, p_.shape, 0.1)
                elif leaf == 'bias':
                    t = spec_tensor(prefix + n, p_.shape, 0.02)
                else:
                    t = spec_tensor(prefix + n, p_.shape, scales(n, p_))
                p_.copy_(t)
                sums[n] = float(t.double().sum())
        return sums

    # ---- decoder layer: L = 40 (14 vision-expert tokens), batch 2 with right padding on sample 1; inputs from spec_tensor as well
    cfg = R.config.CogVLMConfig()          # true widths are the defaults: 4096 / 11008 / 32 heads
    cfg.num_hidden_layers = 1
    layer = mc.CogVLMDecoderLayer(cfg)
    sums = fill(layer, 'dec.', lambda n, p_: 1.0 / math.sqrt(p_.shape[1]))
    B, L, h = 2, 40, cfg.hidden_size
    x = spec_tensor('dec.x', (B, L, h), 0.5)
    tt = torch.zeros(B, L, dtype=torch.long)
    tt[:, 1:15] = 1
    am = torch.ones(B, L, dtype=torch.long)
    am[1, 33:] = 0
    pos = torch.cat([torch.tensor([0, 1]), torch.full((12,), 2), torch.tensor([3, 4]), torch.arange(5, 5 + L - 16)])[None].repeat(B, 1)
    gy = spec_tensor('dec.gy', (B, L, h), 1.0) * am[..., None]
    watch = ['self_attn.vision_expert_query_key_value.weight', 'mlp.language_mlp.down_proj.weight', 'input_layernorm.weight']

    def run_dec(dtype):
        lay = layer.to(dtype)
        lay.self_attn.rotary_emb.max_seq_len_cached = 0      # as in training: the table is first built AFTER the precision conversion
        for p_ in lay.parameters():
            p_.grad = None
        xx = x.detach().to(dtype).clone().requires_grad_()
        y = lay(xx, tt, pos, am.bool())[0]
        y = torch.where(am.bool()[..., None], y, torch.zeros((), dtype=dtype))      # padded rows are uninitialised memory in the reference
        (y.float() * gy).sum().backward()
        ps = dict(lay.named_parameters())
        return dict(y=y.detach().clone(), dx=torch.where(am.bool()[..., None], xx.grad, torch.zeros((), dtype=dtype)),
                    wgrad_rows={n: grad_rows(ps[n].grad) for n in watch},
                    wgrad_norm={n: float(ps[n].grad.double().norm()) for n in watch})
    res['decoder_layer'] = dict(shape=(B, L, h), token_type_ids=tt, attention_mask=am, position_ids=pos, param_sums=sums,
                                input_sums=dict(x=float(x.double().sum()), gy=float(gy.double().sum())),
                                fp32=run_dec(torch.float32))
    res['decoder_layer']['bf16'] = run_dec(torch.bfloat16)
    del layer

    # ---- ViT layer: two packed sequences (one 3-D-sized 2049-token sequence would make the fixture 15 MB; 97 + 40 tokens cover the
    # block-diagonal mask at hd = 112; the Nv = 2049 case is an oracle comparison on the GPU, tests/test_model_gpu.py)
    from argparse import Namespace
    # EVA2-CLIP-E widths (THUDM/cogvlm-chat-hf config.json: the reference reads them from the checkpoint's config)
    vc = Namespace(in_channels=3, hidden_size=1792, num_heads=16, num_hidden_layers=1, intermediate_size=15360, hidden_act='gelu',
                   dropout_prob=0.0, layer_norm_eps=1e-6)
    vl = vis.TransformerLayer(vc)
    vsums = fill(vl, 'vit.', lambda n, p_: 1.0 / math.sqrt(p_.shape[1]))
    lens = [97, 40]
    xv = spec_tensor('vit.x', (1, sum(lens), vc.hidden_size), 0.5)
    gv = spec_tensor('vit.gy', (1, sum(lens), vc.hidden_size), 1.0)
    from xformers.ops.fmha import BlockDiagonalMask
    vwatch = ['attention.query_key_value.weight', 'mlp.fc2.weight', 'post_attention_layernorm.weight']

    def run_vit(dtype):
        lay = vl.to(dtype)
        for p_ in lay.parameters():
            p_.grad = None
        xx = xv.detach().to(dtype).clone().requires_grad_()
        y = lay(xx, BlockDiagonalMask(lens))
        (y.float() * gv).sum().backward()
        ps = dict(lay.named_parameters())
        return dict(y=y.detach()[0].clone(), dx=xx.grad[0].clone(),
                    wgrad_rows={n: grad_rows(ps[n].grad) for n in vwatch},
                    wgrad_norm={n: float(ps[n].grad.double().norm()) for n in vwatch})
    res['vit_layer'] = dict(lens=lens, param_sums=vsums, input_sums=dict(x=float(xv.double().sum()), gy=float(gv.double().sum())),
                            fp32=run_vit(torch.float32))
    res['vit_layer']['bf16'] = run_vit(torch.bfloat16)
    del vl

    # ---- SAM two-way block at 768 (8 heads, mlp 2048) on an [8,16,16]-sized key set is 2048 keys x 768; fixture uses the 3-D grid
    # 128 keys to stay small, 7 query tokens, 2 prompts. fp32 only (the heads are fp32 islands).
    blk = tr.TwoWayAttentionBlock(embedding_dim=768, num_heads=8, mlp_dim=2048, skip_first_layer_pe=False)
    bsums = fill(blk, 'twoway.', lambda n, p_: 1.0 / math.sqrt(p_.shape[1]))
    nq, nk, P_ = 7, 128, 2
    queries = spec_tensor('twoway.queries', (P_, nq, 768), 1.0).requires_grad_()
    keys = spec_tensor('twoway.keys', (P_, nk, 768), 1.0).requires_grad_()
    qpe = spec_tensor('twoway.qpe', (P_, nq, 768), 1.0)
    kpe = spec_tensor('twoway.kpe', (P_, nk, 768), 1.0)
    q2, k2 = blk(queries, keys, qpe, kpe)
    (q2.square().mean() + k2.square().mean()).backward()
    res['two_way_block'] = dict(shape=(P_, nq, nk, 768), param_sums=bsums,
                                input_sums=dict(queries=float(queries.double().sum()), keys=float(keys.double().sum())),
                                q_out=q2.detach(), k_out=k2.detach(), dq=queries.grad.clone(), dk=keys.grad.clone())
    torch.save(res, OUT / 'f2_true_width.pt')


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    R = ref_shims.load()
    only = os.environ.get('GOLDEN_ONLY')
    for fn in (f1_masks, f3_units, f4_vit_identity, f5_tiny_lm, f6_sam, f7_losses, f8_training_step, f9_lora_targets, f10_state_dict_adapters, f11_vlm_inputs, f12_generation,
               f13_bf16_true, f2_true_width):
        if only and only not in fn.__name__:
            continue
        if fn in (f13_bf16_true, f2_true_width):
            with zero_filled_empty():
                fn(R)
        else:
            fn(R)
        print('wrote', fn.__name__)
    for p in sorted(OUT.glob('*.pt')):
        print(p.name, os.path.getsize(p) // 1024, 'KiB')


if __name__ == '__main__':
    main()
