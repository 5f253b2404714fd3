"""fp32 islands on the MI355X: f32 attention kernel, Sam / InstanceSam heads, losses and the full training_step,
against the CPU oracle. The fp32 kernels are exact-f32 MFMA, so tolerances are fp32 summation-order level (1e-4
relative on end-to-end outputs, 1e-5 on single kernels); the bf16 LM in front of the heads bounds the end-to-end
loss at bf16 level (5e-3)."""
import pytest
import torch

from tests._gpu_common import cpu, oracle_cfg, oracle_state, randomize_, rel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('hd,H', [(64, 2), (96, 8), (48, 8), (16, 8), (8, 8)])
@pytest.mark.parametrize('Lq,Lk,Bn', [(9, 9, 3), (9, 784, 2), (300, 9, 2), (130, 70, 1)])
def test_attention_f32_dense(dev, hd, H, Lq, Lk, Bn):
    from mmmm_amd import kernels as K
    q = torch.randn(Bn, Lq, H * hd, device=dev)
    k = torch.randn(Bn, Lk, H * hd, device=dev)
    v = torch.randn(Bn, Lk, H * hd, device=dev)
    scale = hd ** -0.5
    out, lse = K.attn_f32_fwd(q, k, v, H, hd, scale)
    qf, kf, vf = (t.clone().requires_grad_() for t in (q, k, v))
    qh, kh, vh = (t.view(Bn, -1, H, hd).transpose(1, 2) for t in (qf, kf, vf))
    ref = ((qh @ kh.transpose(2, 3) * scale).softmax(-1) @ vh).transpose(1, 2).reshape(Bn, Lq, H * hd)
    assert rel(out, ref) < 1e-5
    dout = torch.randn_like(out)
    ref.backward(dout)
    dq, dk, dv = K.attn_f32_bwd(q, k, v, out, lse, dout, H, hd, scale)
    assert rel(dq, qf.grad) < 2e-5 and rel(dk, kf.grad) < 2e-5 and rel(dv, vf.grad) < 2e-5


def test_attention_f32_varlen(dev):
    from mmmm_amd import kernels as K
    H, hd, lens = 2, 64, [8, 16, 17, 130]
    T = sum(lens)
    qkv = torch.randn(T, 3 * H * hd, device=dev)
    C = H * hd
    cu = torch.tensor([0, 8, 24, 41, 171], dtype=torch.int32, device=dev)
    out, lse = K.attn_f32_fwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], H, hd, hd ** -0.5, cu, max(lens))
    ref = torch.zeros_like(out)
    s = 0
    for n in lens:
        q, k, v = (qkv[s:s + n, i * C:(i + 1) * C].view(n, H, hd).transpose(0, 1) for i in range(3))
        ref[s:s + n] = ((q @ k.transpose(1, 2) * hd ** -0.5).softmax(-1) @ v).transpose(0, 1).reshape(n, C)
        s += n
    assert rel(out, ref) < 1e-5


@pytest.mark.parametrize('split,tol', [(2, 3e-5), (3, 3e-6)])
@pytest.mark.parametrize('lens', [[784, 784, 784, 784], [8, 16, 17, 130, 2048]])
def test_attention_f32_split_products(dev, split, tol, lens):
    """head_dim 64 (SAM ViT-B) with the four products of attention on split-bf16 MFMAs (`f32_split` 2: three products, ~2^-17 each;
    3: six products = fp32 products) against an fp64 reference, forward and all three gradients, packed var-len; and the exact
    form's error on the same problem for scale (the bound is relative to the largest magnitude, as everywhere in this file)"""
    from mmmm_amd import kernels as K
    H, hd = 12, 64
    T, C = sum(lens), H * hd
    torch.manual_seed(11)
    qkv = torch.randn(T, 3 * C, device=dev)
    qkv[:, :2 * C] *= 1.5                                     # scores with a spread of a few units, like trained attention
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=dev)
    dout = torch.randn(T, C, device=dev)
    x = qkv.double().requires_grad_()
    ref = torch.zeros(T, C, dtype=torch.float64, device=dev)
    s = 0
    outs = []
    for n in lens:
        q, k, v = (x[s:s + n, i * C:(i + 1) * C].view(n, H, hd).transpose(0, 1) for i in range(3))
        outs.append(((q @ k.transpose(1, 2) * hd ** -0.5).softmax(-1) @ v).transpose(0, 1).reshape(n, C))
        s += n
    ref = torch.cat(outs)
    ref.backward(dout.double())
    res = {}
    for mode in (0, split):
        out, lse = K.attn_f32_fwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], H, hd, hd ** -0.5, cu, max(lens), f32_split=mode)
        dqkv = torch.empty_like(qkv)
        K.attn_f32_bwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, lse, dout, H, hd, hd ** -0.5, cu, max(lens),
                       grads=(dqkv[:, :C], dqkv[:, C:2 * C], dqkv[:, 2 * C:]), f32_split=mode)
        res[mode] = (rel(out, ref), rel(dqkv[:, :C], x.grad[:, :C]), rel(dqkv[:, C:2 * C], x.grad[:, C:2 * C]), rel(dqkv[:, 2 * C:], x.grad[:, 2 * C:]))
    assert max(res[0]) < 3e-6, res
    assert max(res[split]) < tol, res


def _tiny_sams(dev):
    from mmmm_amd.models import build_instance_sam, build_sam
    sam = build_sam(embed_dim=128, encoder_num_layers=2, num_heads=2, patch_size=(4, 8, 8), pos_embed_shape=(2, 2, 4))
    isam = build_instance_sam(embed_dim=128, encoder_num_layers=2, num_heads=2, patch_size=(4, 8, 8), pos_embed_shape=(2, 2, 4), num_instances=6)
    randomize_(sam, 60)
    randomize_(isam, 61)
    return sam.to(dev), isam.to(dev)


def _sam_cfg(instance):
    from oracle import vividmed as O
    return O.SamCfg(embed_dim=128, num_layers=2, num_heads=2, mlp_dim=512, patch_size=(4, 8, 8), pos_embed_shape=(2, 2, 4),
                    num_instances=6 if instance else 3, instance=instance)


def test_sam_and_isam_match_oracle(dev):
    from oracle import vividmed as O
    sam, isam = _tiny_sams(dev)
    g = torch.Generator().manual_seed(6)
    images = [torch.rand(3, 8, 16, 32, generator=g), torch.rand(3, 1, 16, 32, generator=g), torch.rand(3, 4, 32, 16, generator=g)]
    patch = [(4, 8, 8), (1, 8, 8), (2, 8, 8)]
    prompts = [torch.randn(1, 128, generator=g), torch.randn(3, 128, generator=g), torch.randn(2, 128, generator=g)]
    pg = [p.to(dev).requires_grad_() for p in prompts]
    masks = sam([x.to(dev) for x in images], patch, pg)
    sum(m.square().mean() for m in masks).backward()
    pc = [p.clone().requires_grad_() for p in prompts]
    ref = O.sam_forward({f'sam.{k}': v for k, v in oracle_state(sam).items()}, _sam_cfg(False), 'sam', images, patch, pc)
    sum(m.square().mean() for m in ref).backward()
    for a, b in zip(masks, ref):
        assert a.shape == b.shape and rel(a, b) < 1e-4
    for a, b in zip(pg, pc):
        assert rel(a.grad, b.grad) < 2e-4
    pg = [p.to(dev).requires_grad_() for p in prompts]
    out = isam([x.to(dev) for x in images], patch, pg, need_masks=True)
    (sum(b.sum() for b in out.boxes) + sum(d.square().sum() for d in out.disc_logit) + sum(m.mean() for m in out.masks_logits)).backward()
    pc = [p.clone().requires_grad_() for p in prompts]
    full, low, boxes, disc = O.isam_forward({f'isam_model.{k}': v for k, v in oracle_state(isam).items()}, _sam_cfg(True), 'isam_model',
                                            images, patch, pc)
    (sum(b.sum() for b in boxes) + sum(d.square().sum() for d in disc) + sum(m.mean() for m in full)).backward()
    for a, b in zip(out.boxes, boxes):
        assert rel(a, b) < 1e-4
    for a, b in zip(out.disc_logit, disc):
        assert rel(a, b) < 1e-4
    for a, b in zip(out.masks_logits, full):
        assert rel(a, b) < 1e-4
    for a, b in zip(pg, pc):
        assert rel(a.grad, b.grad) < 2e-4


def test_sam_parameter_gradients_through_the_bucket_path_match_oracle(dev):
    """Unfrozen SAM (README Stage 1): with a BucketedGradAllReduce attached, the fp32 weight / bias gradients are written by
    the GEMM residual path and the column-sum kernel straight into the bucket views from a side stream
    (functional._off_critical_path). They must equal the oracle's parameter gradients, and a second accumulation must add."""
    from oracle import vividmed as O
    from mmmm_amd.ddp import BucketedGradAllReduce
    sam, _ = _tiny_sams(dev)
    sam.train()
    g = torch.Generator().manual_seed(16)
    images = [torch.rand(3, 8, 16, 32, generator=g), torch.rand(3, 4, 32, 16, generator=g)]
    patch = [(4, 8, 8), (2, 8, 8)]
    prompts = [torch.randn(2, 128, generator=g), torch.randn(3, 128, generator=g)]
    trainable = [p for p in sam.parameters() if p.requires_grad]
    ddp = BucketedGradAllReduce(trainable, world_size=1, bucket_bytes=1 << 20)
    try:
        ddp.zero_grad()
        for _ in range(2):          # two micro-batches: the second pass must ACCUMULATE
            masks = sam([x.to(dev) for x in images], patch, [p.to(dev) for p in prompts])
            sum(m.square().mean() for m in masks).backward()
        ddp.finish()
        torch.cuda.synchronize()
        sd = {f'sam.{k}': v.requires_grad_(v.is_floating_point()) for k, v in oracle_state(sam).items()}
        ref = O.sam_forward(sd, _sam_cfg(False), 'sam', images, patch, prompts)
        sum(m.square().mean() for m in ref).backward()
        checked, worst = 0, (0.0, '')
        for name, p in sam.named_parameters():
            gr = sd[f'sam.{name}'].grad
            if not p.requires_grad or gr is None or gr.norm() < 1e-7:     # (k_proj.bias: softmax is shift-invariant, gradient == 0)
                continue
            assert p.grad is not None, name
            assert rel(p.grad, 2 * gr) < 1e-4, (name, rel(p.grad, 2 * gr))          # measured worst 1.2e-5 (three-product encoder blocks)
            worst = max(worst, (rel(p.grad, 2 * gr), name))
            checked += 1
        assert checked > 40
        print(f'[unfrozen SAM parameter gradients vs oracle] worst relative error {worst[0]:.2e} at {worst[1]} over {checked} tensors')
    finally:
        ddp.remove()
        for p in sam.parameters():
            p.grad = None


def test_losses_match_oracle_and_hungarian_is_bit_exact(dev):
    from oracle import vividmed as O
    from mmmm_amd.models.loss import DiceFocalLoss
    from mmmm_amd.models.segvol.modeling.sam import InstanceSamLoss
    f = torch.load('tests/golden/f7_losses.pt', weights_only=False)   # reference outputs
    d = f['dice_focal']
    dfl = DiceFocalLoss(dice_weight=2, focal_weight=2, focal_gamma=2)
    got = dfl(d['x'].to(dev), d['t'].to(dev), return_dict=True)
    for k in got:
        assert rel(got[k], d['out'][k]) < 1e-5
    i = f['isam']
    il = InstanceSamLoss(use_neg_mask=False, box_l1_weight=5, box_giou_weight=2, disc_weight=2, disc_focal_gamma=2, disc_focal_alpha=0.85)
    br, dl = i['boxes_reg'].to(dev).requires_grad_(), i['disc'].to(dev).requires_grad_()
    dummy = br.new_empty((*br.shape[:2], 0, 0, 0))
    loss, log = il.compute_loss(dummy, dummy, br, dl, None, i['boxes_label'].to(dev), i['index_offsets'].to(dev))
    assert rel(loss, i['loss']) < 1e-5
    match = il._match_all(br.detach(), dl.detach(), i['boxes_label'].to(dev), i['index_offsets'])
    assert match.is_cuda                                     # solved on the device: no device->host synchronisation
    assert torch.equal(match.cpu(), i['match'])              # == the reference's SciPy assignment, bit-exact
    loss.backward()
    assert rel(br.grad, i['d_boxes']) < 1e-4 and rel(dl.grad, i['d_disc']) < 1e-4
    for k, v in i['log'].items():
        assert rel(log[k], v) < 1e-5, k


def test_training_step_end_to_end_matches_oracle(dev):
    from oracle import vividmed as O
    from mmmm_amd.data.synthetic import SpecialTokens, make_batch
    from mmmm_amd.models.loss import DiceFocalLoss
    from mmmm_amd.models.lora import LoraConfig
    from mmmm_amd.models.mmmm import MMMMForCausalLM, MyPrecision, VisionArgs
    from mmmm_amd.models.segvol.modeling.sam import InstanceSamLoss
    from mmmm_amd.utils import apply_lora
    from tests.test_model_gpu import tiny_config
    sam, isam = _tiny_sams('cpu')
    tok = SpecialTokens(base_vocab=184)
    m = MMMMForCausalLM.build(None, vision_override=VisionArgs(pos_embed_shape=(2, 2, 4), patch_size=(4, 8, 8)), tokenizer=tok,
                              sam=sam, isam=isam, mask_loss=DiceFocalLoss(dice_weight=2, focal_weight=2, focal_gamma=2),
                              isam_loss=InstanceSamLoss(use_neg_mask=False, box_l1_weight=5, box_giou_weight=2, disc_weight=2,
                                                        disc_focal_gamma=2, disc_focal_alpha=0.85), config=tiny_config())
    apply_lora(m, LoraConfig(r=64, lora_alpha=8, lora_dropout=0.0, use_rslora=True))
    randomize_(m, 80)
    m.to(dev)
    MyPrecision().convert_module(m)
    assert m.sam.image_encoder.norm.weight.dtype == torch.float32 and m.lm_head.weight.dtype == torch.bfloat16
    m.train()
    m.sam.eval(); m.isam_model.eval()
    m.on_fit_start()
    batch = make_batch([(3, 1, 16, 32), (3, 8, 16, 32), (3, 4, 32, 16)], [(1, 8, 8), (4, 8, 8), (2, 8, 8)], [(1, 2, 2), (2, 2, 2), (1, 1, 1)],
                       [30, 28, 33], tok=tok, seed=8, instance=[False, True, False], device=dev, n_pairs=3)
    loss = m.training_step(batch)
    loss.backward()
    sd = {k: v.requires_grad_(v.is_floating_point()) for k, v in oracle_state(m).items()}
    scfg = O.StepCfg(lm=oracle_cfg(m.config), sam=_sam_cfg(False), isam=_sam_cfg(True),
                     mask_loss=dict(dice_weight=2, focal_weight=2, focal_gamma=2), isam_loss=O.ISamLossCfg(),
                     bop_token_id=tok.bop_token_id, eop_token_id=tok.eop_token_id)
    cb = cpu(batch)
    cb['image'] = [x.float() for x in cb['image']]
    ref_loss, ref_log = O.training_step(sd, scfg, cb, rope_dtype=torch.bfloat16)
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) / abs(ref_loss.item()) < 5e-3, (loss.item(), ref_loss.item())
    for k, v in ref_log.items():
        assert k in m.logged, k
        assert abs(float(m.logged[k]) - float(v)) <= 2e-2 * max(1.0, abs(float(v))), (k, float(m.logged[k]), float(v))
    for name in ('vg_proj.0.weight', 'vg_proj.2.bias', 'model.layers.1.self_attn.language_expert_dense.lora_B.default.weight', 'model.norm.weight'):
        g = dict(m.named_parameters())[name].grad
        assert g is not None, name
        assert rel(g.float(), sd[name].grad) < 8e-2, (name, rel(g.float(), sd[name].grad))


def test_device_matching_equals_host_matching_for_every_target_count(dev):
    """`vm_lsap_f32` route (no synchronisation) against the reference's route (SciPy on host copies) on one sample whose targets
    have 0, 1, 3, 6 (= queries) and 9 (> queries) boxes: identical assignment, and the loss built from static-size device index
    tensors equals the loss built from the host assignment — values, logged terms and gradients"""
    from mmmm_amd.models.segvol.modeling.sam import InstanceSamLoss
    il = InstanceSamLoss(use_neg_mask=False, box_l1_weight=5, box_giou_weight=2, disc_weight=2, disc_focal_gamma=2, disc_focal_alpha=0.85)
    g = torch.Generator().manual_seed(3)
    counts = [0, 1, 3, 6, 9]
    nt, nq = len(counts), 6
    offs, s = [], 0
    for c in counts:
        offs.append((s, s + c))
        s += c
    index_offsets = torch.tensor(offs, dtype=torch.int64)
    centre = torch.rand(s, 3, generator=g) * 0.5 + 0.25
    size = torch.rand(s, 3, generator=g) * 0.3 + 0.05
    boxes_label = torch.cat([centre, size], 1).to(dev)
    br = torch.cat([torch.rand(nt, 1 + nq, 3, generator=g) * 0.5 + 0.25, torch.rand(nt, 1 + nq, 3, generator=g) * 0.3 + 0.05], -1).to(dev)
    dl = torch.randn(nt, nq, generator=g).to(dev)
    dev_match = il._match_all(br, dl, boxes_label, index_offsets)
    assert dev_match.is_cuda
    costs, metas = il._match_costs(br[:, 1:], dl.float(), boxes_label, offs)
    host_match = il._assign([c.cpu().numpy() for c in costs], metas, nt, nq)
    assert torch.equal(dev_match.cpu(), host_match)
    assert [(m >= 0).sum().item() for m in host_match] == [min(c, nq) for c in counts]
    dummy = br.new_empty((nt, 1 + nq, 0, 0, 0))
    res = []
    for match in (dev_match, host_match):
        b, d = br.clone().requires_grad_(), dl.clone().requires_grad_()
        loss, log = il.compute_loss(dummy, dummy, b, d, None, boxes_label, index_offsets, match=match)
        loss.backward()
        res.append((loss.detach(), {k: v.detach() for k, v in log.items()}, b.grad, d.grad))
    (l0, g0, b0, d0), (l1, g1, b1, d1) = res
    # (the device assignment takes the fused loss kernels, the host assignment the element-wise torch form: same terms, fp32
    # sums in a different order)
    assert rel(l0, l1) < 2e-6 and rel(b0, b1) < 1e-5 and rel(d0, d1) < 1e-5
    assert g0.keys() == g1.keys() and all(rel(g0[k], g1[k]) < 2e-6 for k in g0)
    # the element-wise form on the device assignment is bit-identical to the host route
    il.fused = False
    b, d = br.clone().requires_grad_(), dl.clone().requires_grad_()
    loss, log = il.compute_loss(dummy, dummy, b, d, None, boxes_label, index_offsets, match=dev_match)
    loss.backward()
    assert torch.equal(loss.detach(), l1) and torch.equal(b.grad, b1) and torch.equal(d.grad, d1)
    assert all(torch.equal(log[k].detach(), g1[k]) for k in g1)
