"""Model-level parity on the MI355X: the HIP product path (through the C ABI) against the CPU oracle on the
same explicit weights and inputs.

Tolerances (stated per BASELINE.json north_star: "logits within 1e-3 rel ... token-type indexing bit-exact"):
  * bf16 path, evidence-based (same rule as tests/test_truewidth_gpu.py, where the yardstick is the REFERENCE's own bf16-true run):
    the oracle is evaluated twice on the same bf16-rounded weights — in fp32 and in its bf16-true mode (pinned to the reference's
    bf16 fixtures by tests/test_oracle_truewidth_cpu.py) — which gives the bf16 error of the reference arithmetic on THIS input,
    e_or = |oracle_bf16 - oracle_fp32|. The HIP path must satisfy |hip - oracle_fp32| <= 1.3 e_or and |hip - oracle_bf16| <= 1.5 e_or
    (`bf16_ok`). The coarse absolute bounds of round 1 (2e-2 on logits / hidden states, 5e-3 on the loss, 6e-2 on gradients) stay
    as a backstop. 1e-3 is not reachable by any bf16-true evaluation, the reference's included (e_or is 5e-3 .. 1.5e-2 here).
  * integer routing (expert masks, row maps) is bit-exact.
"""
import pytest
import torch

from tests._gpu_common import cpu, lora_dropout_on, oracle_cfg, oracle_state, randomize_, rel

pytestmark = pytest.mark.gpu


def tiny_config(n_lm=2, n_vit=2):
    from mmmm_amd.models.cogvlm.configuration_cogvlm import CogVLMConfig
    return CogVLMConfig(vocab_size=192, hidden_size=128, intermediate_size=256, num_hidden_layers=n_lm, num_attention_heads=2,
                        vision_config=dict(in_channels=3, hidden_size=128, num_heads=2, num_hidden_layers=n_vit,
                                           intermediate_size=256, layer_norm_eps=1e-6, patch_size=(4, 8, 8),
                                           pos_embed_shape=(2, 2, 4)))


def make_inputs(dev, seed=0):
    from mmmm_amd.data.synthetic import SpecialTokens, make_batch
    tok = SpecialTokens(base_vocab=184)
    batch = make_batch([(3, 1, 16, 32), (3, 8, 16, 32), (3, 4, 32, 16)], [(1, 8, 8), (4, 8, 8), (2, 8, 8)],
                       [(1, 2, 2), (2, 2, 2), (1, 1, 1)], [24, 17, 30], tok=tok, seed=seed, device=dev)
    return batch, tok


@pytest.fixture(scope='module')
def lm(dev):
    from mmmm_amd.models.mmmm import MMMMForCausalLM, VisionArgs
    from mmmm_amd.models.lora import LoraConfig
    from mmmm_amd.utils import apply_lora
    cfg = tiny_config()
    m = MMMMForCausalLM(cfg, vision_override=VisionArgs(pos_embed_shape=(2, 2, 4), patch_size=(4, 8, 8)))
    apply_lora(m, LoraConfig(r=64, lora_alpha=8, lora_dropout=0.0, use_rslora=True))
    randomize_(m, 123)
    m.to(dev).to(torch.bfloat16)
    return m


def run_oracle(m, batch, need_grad=False, dtype=torch.float32):
    """dtype fp32: the exact result on the model's (bf16-rounded) weights; bf16: the oracle's bf16-true mode"""
    from oracle import vividmed as O
    sd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in oracle_state(m).items()}
    if need_grad:
        sd = {k: v.requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    vi = cpu(batch['vlm_inputs'])
    out = O.causal_lm_forward(sd, oracle_cfg(m.config), vi['input_ids'], image=[x.to(dtype) for x in cpu(batch['image'])],
                              patch_size=batch['patch_size'], pool_size=batch['pool_size'], token_type_ids=vi['token_type_ids'],
                              attention_mask=vi['attention_mask'], position_ids=vi['position_ids'], labels=vi['labels'],
                              weight=vi['weight'].to(dtype), rope_dtype=torch.bfloat16)
    return out, sd


def bf16_ok(what, hip, or16, or32, a=1.3, b=1.5, floor=1e-4):
    e_or, e_hip, d = rel(or16.float(), or32), rel(hip.float(), or32), rel(hip.float(), or16.float())
    assert e_hip <= a * e_or + floor and d <= b * e_or + floor, (what, dict(e_oracle_bf16=e_or, e_hip=e_hip, hip_vs_oracle_bf16=d))


def test_routing_is_bit_exact(dev, lm):
    from oracle import vividmed as O
    batch, _ = make_inputs(dev)
    vi = batch['vlm_inputs']
    rt = lm.model.build_routing(vi['token_type_ids'], vi['attention_mask'], vi['position_ids'])
    v, l = O.get_expert_mask(cpu(vi['token_type_ids']), cpu(vi['attention_mask']).bool())
    mask = rt.expert_mask.cpu()
    assert torch.equal((mask & 1).bool(), v) and torch.equal((mask & 2).bool(), l)
    n_vis, n_rows = rt.counts[:2].tolist()
    assert n_vis == int(v.sum()) and n_rows == int((v | l).sum())
    # row_pos carries the explicit position id of each packed row
    tor = rt.tok_of_row[:n_rows].long().cpu()
    assert torch.equal(rt.row_pos[:n_rows].cpu().long(), cpu(vi['position_ids']).reshape(-1)[tor])


def test_lm_forward_matches_oracle(dev, lm):
    lm.eval()
    batch, _ = make_inputs(dev)
    with torch.no_grad():
        out = lm(**batch['vlm_inputs'], image=batch['image'], patch_size=batch['patch_size'], pool_size=batch['pool_size'],
                 output_hidden_states=True, materialize_logits=True)
    ref, _ = run_oracle(lm, batch)
    with torch.no_grad():
        ref16, _ = run_oracle(lm, batch, dtype=torch.bfloat16)
    am = cpu(batch['vlm_inputs']['attention_mask']).bool()
    assert abs(out.loss.item() - ref.loss.item()) / abs(ref.loss.item()) < 5e-3, (out.loss.item(), ref.loss.item())
    assert rel(out.logits.cpu()[am], ref.logits[am]) < 2e-2
    bf16_ok('logits', out.logits.cpu()[am], ref16.logits[am], ref.logits[am])
    assert len(out.hidden_states) == len(ref.hidden_states)
    for i in range(len(ref.hidden_states)):
        assert rel(out.hidden_states[i].float().cpu()[am], ref.hidden_states[i][am]) < 2e-2, i
        if i:
            bf16_ok(f'hidden {i}', out.hidden_states[i].float().cpu()[am], ref16.hidden_states[i][am], ref.hidden_states[i][am])


def test_vision_tower_matches_oracle(dev, lm):
    from oracle import vividmed as O
    lm.eval()
    batch, _ = make_inputs(dev, seed=3)
    with torch.no_grad():
        feats = lm.model.vision(batch['image'], batch['patch_size'], batch['pool_size'])
    ref = O.vision_forward(oracle_state(lm), oracle_cfg(lm.config), [x.float() for x in cpu(batch['image'])], batch['patch_size'],
                           batch['pool_size'])
    for a, b in zip(feats, ref):
        assert a.shape == b.shape
        assert rel(a.float(), b) < 2e-2


def test_lm_backward_matches_oracle(dev, lm):
    lm.train()
    lm.gradient_checkpointing_enable()
    batch, _ = make_inputs(dev, seed=5)
    for p in lm.parameters():
        p.grad = None
    out = lm(**batch['vlm_inputs'], image=batch['image'], patch_size=batch['patch_size'], pool_size=batch['pool_size'])
    out.loss.backward()
    ref, sd = run_oracle(lm, batch, need_grad=True)
    ref.loss.backward()
    ref16, sd16 = run_oracle(lm, batch, need_grad=True, dtype=torch.bfloat16)
    ref16.loss.backward()
    assert abs(out.loss.item() - ref.loss.item()) / abs(ref.loss.item()) < 5e-3
    checked = 0
    worst = {}
    for name, p in lm.named_parameters():
        if not p.requires_grad:
            assert p.grad is None, name
            continue
        assert p.grad is not None, f'no gradient for trainable {name}'
        g_ref = sd[name].grad
        if g_ref is None or g_ref.norm() == 0:
            continue
        e = rel(p.grad.float(), g_ref)
        worst[name] = e
        # (a vector summed from a handful of bf16 rows — cls / boi / eoi embeddings — has no averaging: wider factor)
        few = p.numel() <= 2 * lm.config.hidden_size
        bf16_ok('d' + name, p.grad.float().cpu(), sd16[name].grad.float(), g_ref, 2.5 if few else 1.4, 2.5 if few else 1.6)
        checked += 1
    bad = {k: v for k, v in worst.items() if v > 6e-2}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:10]
    assert checked > 40
    lm.model.gradient_checkpointing = False
    lm.model.vision.transformer.gradient_checkpointing = False


def test_lm_backward_matches_oracle_with_lora_dropout_on(dev, lm):
    """conf/lora.yaml:3 lora_dropout 0.05 is what the benchmark runs, and until round 5 the one arithmetic without an oracle comparison
    (the oracle had p = 0 only). The oracle now takes the keep-masks as data (oracle.vividmed.LORA_DROPOUT): forward loss and every
    trainable gradient against the fp32 and the bf16-true oracle, with the layers checkpointed (the recompute must redraw the forward's
    masks) — ViT-E, the GLU adapter and both experts of every decoder linear drop what the HIP kernels dropped.

    THREE mask realisations, and a tensor must hold the p = 0 test's e_ref bound in at least two of them. On this 43-token model WHICH
    elements are dropped decides how well conditioned the backward is: for some realisations a 7 % error of the gradient entering one
    attention layer (normal bf16 noise here) leaves it as 40 % on dq / dk — the kernel agrees with fp32 autograd on the same inputs to
    0.5 %, it is the softmax backward's cancellation — and everything below that layer inherits it, while the oracle's own bf16 run, with
    a different rounding pattern, stays at 4 % (tools/debug_dropout_parity.py: 3 of 8 seeds; per-site forward projections agree to bf16
    rounding in all of them). A mask that does not reach a kernel is a different thing: that site's LoRA gradients are ~30 % off in EVERY
    realisation. The median over all tensors of e_hip / e_ref must stay below 1.15 in every realisation (measured 0.86 .. 0.89)."""
    import statistics
    from oracle import vividmed as O
    from mmmm_amd.models.lora import StepState
    lm.train()
    lm.gradient_checkpointing_enable()
    batch, _ = make_inputs(dev, seed=11)
    old_seed, old_step = StepState.seed, StepState.step
    # (lora_dropout_on numbers the sites itself and the step counter is pinned here: the three realisations do not depend on what the process
    # built or stepped before — with the counter left at whatever earlier tests had bumped it to, a `-k` selection of the suite drew other masks
    # than the full run and landed on a badly conditioned one)
    passes: dict = {}
    try:
        StepState.step = 0
        for seed in (1, 2, 3):
            StepState.seed = seed
            with lora_dropout_on(lm, batch['vlm_inputs'], 0.05) as masks:
                for p in lm.parameters():
                    p.grad = None
                out = lm(**batch['vlm_inputs'], image=batch['image'], patch_size=batch['patch_size'], pool_size=batch['pool_size'])
                out.loss.backward()
                masks.begin()
                ref, sd = run_oracle(lm, batch, need_grad=True)
                ref.loss.backward()
                masks.begin()
                ref16, sd16 = run_oracle(lm, batch, need_grad=True, dtype=torch.bfloat16)
                ref16.loss.backward()
                if seed == 1:
                    # every kind of site was served a mask, and about 5 % of each mask is zero
                    kinds = ('vision.transformer.layers.0.attention.query_key_value', 'vision.transformer.layers.1.mlp.fc2', 'vision.linear_proj.gate_proj',
                             'layers.0.self_attn.vision_expert_query_key_value', 'layers.1.self_attn.language_expert_dense',
                             'layers.1.mlp.language_mlp.down_proj')
                    assert all(any(u.endswith(k) for u in masks.used) for k in kinds), sorted(masks.used)
                    masks.begin()
                    probe = masks('model.vision.transformer.layers.0.mlp.fc1', torch.zeros(40, lm.config.vision_config['hidden_size']))
                    assert 0.02 < 1.0 - probe.float().mean().item() < 0.09
                    # ... and the masks matter: the same oracle without them is somewhere else
                    O.LORA_DROPOUT = None
                    with torch.no_grad():
                        nodrop, _ = run_oracle(lm, batch)
                    O.LORA_DROPOUT = masks
                    assert abs(nodrop.loss.item() - ref.loss.item()) > 20 * abs(out.loss.item() - ref.loss.item()) or abs(nodrop.loss.item() - ref.loss.item()) > 1e-3
            assert abs(out.loss.item() - ref.loss.item()) / abs(ref.loss.item()) < 5e-3, seed
            ratios = []
            for name, p in lm.named_parameters():
                if not p.requires_grad:
                    continue
                assert p.grad is not None, f'no gradient for trainable {name}'
                g_ref = sd[name].grad
                if g_ref is None or g_ref.norm() == 0:
                    continue
                e_or, e_hip, d = rel(sd16[name].grad.float(), g_ref), rel(p.grad.float(), g_ref), rel(p.grad.float(), sd16[name].grad.float())
                few = p.numel() <= 2 * lm.config.hidden_size         # cls / boi / eoi: three single rows, no averaging
                a_, b_ = (2.5, 2.5) if few else (1.4, 1.6)
                ok = e_hip <= a_ * e_or + 1e-4 and d <= b_ * e_or + 1e-4
                passes.setdefault(name, []).append((ok, round(e_hip / max(e_or, 1e-9), 2)))
                ratios.append(e_hip / max(e_or, 1e-9))
                assert e_hip < 0.6, (seed, name, e_hip)              # a gradient that is simply wrong
            assert statistics.median(ratios) < 1.15, (seed, statistics.median(ratios))
        assert len(passes) > 40
        bad = {n: v for n, v in passes.items() if sum(ok for ok, _ in v) < 2}
        assert not bad, sorted(bad.items())[:10]
    finally:
        StepState.seed, StepState.step = old_seed, old_step
        lm.model.gradient_checkpointing = False
        lm.model.vision.transformer.gradient_checkpointing = False


def assert_same_grads(a: dict, b: dict):
    """Bit for bit, except the gradients whose column sums go through fp32 ATOMICS (norm weights / biases: rowwise.hip; ATen's
    patch-embedding backward): three or more workgroups adding into one address may do so in a different order from launch to launch
    (seen once in ~10 runs of the suite). The same exception as tests/test_fullsize_gpu.py; anything actually wrong moves these far
    more than 1e-4."""
    assert a.keys() == b.keys()
    # listed by the module that owns the parameter, not by a substring of the path: the gains / biases of the norm layers (column sums with fp32
    # atomics, rowwise.hip) and the patch embedding's five tensors (ATen's convolution / index backward)
    NORMS = {'input_layernorm', 'post_attention_layernorm', 'norm', 'norm1'}

    def atomic(n):
        owner = n.split('.')[-2] if n.count('.') else ''
        return owner in NORMS or '.patch_embedding.' in n
    for n in a:
        if atomic(n):
            x, y = a[n].float(), b[n].float()
            assert torch.allclose(x, y, rtol=1e-4, atol=1e-6 * float(x.abs().max()) + 1e-12), n
        else:
            assert torch.equal(a[n], b[n]), n


def test_checkpointing_does_not_change_gradients(dev, lm):
    lm.train()
    batch, _ = make_inputs(dev, seed=7)
    grads = []
    for ck in (False, True):
        lm.model.gradient_checkpointing = ck
        lm.model.vision.transformer.gradient_checkpointing = ck
        for p in lm.parameters():
            p.grad = None
        out = lm(**batch['vlm_inputs'], image=batch['image'], patch_size=batch['patch_size'], pool_size=batch['pool_size'])
        out.loss.backward()
        grads.append({n: p.grad.clone() for n, p in lm.named_parameters() if p.grad is not None})
    assert_same_grads(grads[0], grads[1])
    lm.model.gradient_checkpointing = False
    lm.model.vision.transformer.gradient_checkpointing = False


def test_lora_dropout_mask_is_replayed(dev, lm):
    """with lora_dropout > 0 the forward, the checkpoint recompute and the backward must agree on the mask"""
    from mmmm_amd.models.lora import Linear
    lm.train()
    for mod in lm.modules():
        if isinstance(mod, Linear) and mod.lora_cfg is not None:
            mod.lora_cfg.lora_dropout = 0.05
    batch, _ = make_inputs(dev, seed=9)
    res = []
    for ck in (False, True):
        lm.model.gradient_checkpointing = ck
        lm.model.vision.transformer.gradient_checkpointing = ck
        for p in lm.parameters():
            p.grad = None
        out = lm(**batch['vlm_inputs'], image=batch['image'], patch_size=batch['patch_size'], pool_size=batch['pool_size'])
        out.loss.backward()
        res.append((out.loss.item(), {n: p.grad.clone() for n, p in lm.named_parameters() if p.grad is not None}))
    assert res[0][0] == res[1][0]
    assert_same_grads(res[0][1], res[1][1])
    for mod in lm.modules():
        if isinstance(mod, Linear) and mod.lora_cfg is not None:
            mod.lora_cfg.lora_dropout = 0.0
    lm.model.gradient_checkpointing = False
    lm.model.vision.transformer.gradient_checkpointing = False


def _grads(lm, batch):
    for p in lm.parameters():
        p.grad = None
    out = lm(**batch['vlm_inputs'], image=batch['image'], patch_size=batch['patch_size'], pool_size=batch['pool_size'])
    out.loss.backward()
    return out.loss.item(), {n: p.grad.clone() for n, p in lm.named_parameters() if p.grad is not None}


def test_activation_budget_keeps_some_layers_without_changing_gradients(dev, lm):
    """HBM-budgeted checkpointing (ActivationBudget): any split between kept and recomputed layers is the same step"""
    from mmmm_amd.models.lora import ActivationBudget
    lm.train()
    lm.gradient_checkpointing_enable()
    batch, _ = make_inputs(dev, seed=11)
    try:
        ActivationBudget.limit = None
        ActivationBudget.reset()
        ref = _grads(lm, batch)
        assert all(kept == 0 for _, kept in ActivationBudget.last_plan)
        plans = []
        for limit in (1 << 20, 1 << 40):
            ActivationBudget.limit = limit
            ActivationBudget.reset()
            got = _grads(lm, batch)
            plans.append(list(ActivationBudget.last_plan))
            assert got[0] == ref[0]
            assert_same_grads(got[1], ref[1])
        assert all(kept == total for total, kept in plans[1]) and len(plans[1]) == 2
        assert sum(kept for _, kept in plans[0]) < sum(kept for _, kept in plans[1])
    finally:
        ActivationBudget.limit = None
        lm.model.gradient_checkpointing = False
        lm.model.vision.transformer.gradient_checkpointing = False


def test_resident_lora_transposes_match_per_use_transposes(dev, lm):
    from mmmm_amd.models.lora import Linear, LoraTransposes
    lm.train()
    batch, _ = make_inputs(dev, seed=13)
    mods = [m for m in lm.modules() if isinstance(m, Linear) and m.lora_cfg is not None]
    for m in mods:
        m._lora_t = None
    ref = _grads(lm, batch)
    cache = LoraTransposes(lm)
    cache.refresh()
    assert len(cache.linears) == len(mods) and all(m.lora_t()[0] is not None for m in mods)
    for m in mods:
        At, Bt = m.lora_t()
        assert torch.equal(At, m.A.detach().t()) and torch.equal(Bt, m.B.detach().t())
    got = _grads(lm, batch)
    assert got[0] == ref[0]
    assert_same_grads(got[1], ref[1])
    # an in-place parameter update invalidates the cached copies until the next refresh()
    with torch.no_grad():
        mods[0].A.mul_(1.5)
    assert mods[0].lora_t() == (None, None) and mods[1].lora_t()[0] is not None
    cache.refresh()
    assert torch.equal(mods[0].lora_t()[0], mods[0].A.detach().t())
    with torch.no_grad():
        mods[0].A.div_(1.5)
    for m in mods:
        m._lora_t = None


# ------------------------------------------------------------------ true widths (one layer each)
def test_true_width_layers_match_oracle(dev):
    """One decoder layer at h=4096 / i=11008 / 32 heads of 128 and one ViT layer at d=1792 / f=15360 / 16 heads of 112
    (the widths of BASELINE configs[1]) on a 320x320 image + 96 text tokens: the shapes at which the 2-segment GEMM,
    the LoRA split-K / skinny kernels and the hd-128 / hd-112 attention run in production, against the fp32 oracle."""
    from mmmm_amd.models.cogvlm.configuration_cogvlm import CogVLMConfig
    from mmmm_amd.models.mmmm import MMMMForCausalLM, VisionArgs
    from mmmm_amd.models.lora import LoraConfig
    from mmmm_amd.utils import apply_lora
    from mmmm_amd.data.synthetic import SpecialTokens, make_batch
    cfg = CogVLMConfig(vocab_size=2056, hidden_size=4096, intermediate_size=11008, num_hidden_layers=1, num_attention_heads=32,
                       vision_config=dict(in_channels=3, hidden_size=1792, num_heads=16, num_hidden_layers=1, intermediate_size=15360,
                                          layer_norm_eps=1e-6, patch_size=(16, 16, 16), pos_embed_shape=(2, 20, 20)))
    m = MMMMForCausalLM(cfg, vision_override=VisionArgs(pos_embed_shape=(2, 20, 20), patch_size=16))
    apply_lora(m, LoraConfig(r=64, lora_alpha=8, lora_dropout=0.0, use_rslora=True))
    randomize_(m, 321)
    with torch.no_grad():
        # keep the first ViT layer out of softmax saturation: with unit-variance random kernels the z-folded 2-D patch
        # embedding has std ~33, q.k scores in the hundreds, one-hot attention, and dq / dk become pure cancellation noise
        # (P (dP - delta) ~ 0) that no bf16-vs-fp32 comparison can resolve
        m.model.vision.patch_embedding.proj.weight.mul_(0.03)
    m.to(dev).to(torch.bfloat16).train()
    tok = SpecialTokens(base_vocab=2048)
    batch = make_batch([(3, 1, 320, 320), (3, 1, 320, 320)], [(1, 16, 16)] * 2, [(1, 2, 2)] * 2, [96, 61], tok=tok, seed=3, device=dev)
    out = m(**batch['vlm_inputs'], image=batch['image'], patch_size=batch['patch_size'], pool_size=batch['pool_size'])
    out.loss.backward()
    ref, sd = run_oracle(m, batch, need_grad=True)
    ref.loss.backward()
    assert abs(out.loss.item() - ref.loss.item()) / abs(ref.loss.item()) < 5e-3
    worst = {}
    for name, p in m.named_parameters():
        if not p.requires_grad or p.grad is None:
            continue
        g_ref = sd[name].grad
        if g_ref is None or g_ref.norm() == 0:
            continue
        worst[name] = rel(p.grad.float(), g_ref)
    assert len(worst) > 30
    bad = {k: v for k, v in worst.items() if v > 6e-2}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:10]


def test_side_stream_factor_gradients_match_main_stream(dev, lm):
    """LoRA factor gradients written straight into the gradient buckets from a side stream (functional._off_critical_path)
    must equal the single-stream result bit for bit (the kernels are deterministic; only the ordering discipline differs)"""
    from mmmm_amd import functional as Fh
    from mmmm_amd.ddp import BucketedGradAllReduce
    lm.train()
    batch, _ = make_inputs(dev, seed=17)
    plain = _grads(lm, batch)[1]          # no buckets: every gradient goes through autograd's AccumulateGrad
    trainable = [p for p in lm.parameters() if p.requires_grad]
    ddp = BucketedGradAllReduce(trainable, world_size=1, bucket_bytes=1 << 20)
    res = []
    old = Fh.WGRAD_SIDE_STREAM
    try:
        for side in (False, True, True):
            Fh.WGRAD_SIDE_STREAM = side
            ddp.zero_grad()
            out = lm(**batch['vlm_inputs'], image=batch['image'], patch_size=batch['patch_size'], pool_size=batch['pool_size'])
            out.loss.backward()
            ddp.finish()
            torch.cuda.synchronize()
            res.append({n: p.grad.clone() for n, p in lm.named_parameters() if p.grad is not None})
    finally:
        Fh.WGRAD_SIDE_STREAM = old
        ddp.remove()
        for p in lm.parameters():
            p.grad = None
    assert any('lora_A' in n for n in res[0])
    # bucket path (direct accumulation: LoRA factors, norm weights / biases) vs the plain autograd path
    assert plain.keys() == res[0].keys()
    for n in plain:
        assert rel(res[0][n].float(), plain[n].float()) < 2e-3, (n, rel(res[0][n].float(), plain[n].float()))
    for other in res[1:]:
        assert_same_grads(res[0], other)


def test_lm_head_on_labelled_rows_equals_the_full_product(dev):
    """_LMHeadCE compacts the rows that carry a label (device-side, no sync) and runs lm_head + CE on those only. Against the full-size
    form (modeling_cogvlm.LM_HEAD_LABEL_ROWS = False): loss, per-row CE and the hidden-state gradient bit-identical (a row's products do not depend on
    the other rows), the weight gradient equal up to the order of its fp32 row sum."""
    from mmmm_amd.models.cogvlm import modeling_cogvlm as mc
    g = torch.Generator(device='cpu').manual_seed(11)
    M, Kd, V = 1500, 256, 1000
    h0 = torch.randn(M, Kd, generator=g).to(dev).bfloat16()
    W0 = (torch.randn(V, Kd, generator=g) / 16).to(dev).bfloat16()
    labels = torch.randint(0, V, (M,), generator=g)
    labels[torch.rand(M, generator=g) < 0.45] = -100
    labels[-37:] = -100
    labels = labels.to(dev)
    weight = (torch.rand(M, generator=g) + 0.5).to(dev)
    nrows = torch.tensor([M - 37], dtype=torch.int32, device=dev)
    res = {}
    for mode in (False, True):
        mc.LM_HEAD_LABEL_ROWS = mode
        h = h0.clone().requires_grad_(True)
        W = W0.clone().requires_grad_(True)
        loss, row_ce = mc._LMHeadCE.apply(h, W, labels, weight, nrows)
        loss.backward()
        res[mode] = (loss.detach(), row_ce, h.grad, W.grad)
    mc.LM_HEAD_LABEL_ROWS = True
    full, comp = res[False], res[True]
    assert torch.equal(full[0], comp[0])
    assert torch.equal(full[1], comp[1])
    assert torch.equal(full[2].view(torch.int16), comp[2].view(torch.int16))
    assert (comp[2][labels < 0] == 0).all()
    assert rel(comp[3], full[3]) < 2e-3
    # all rows unlabelled: a zero loss and zero gradients, not a division by zero
    h = h0.clone().requires_grad_(True)
    loss, _ = mc._LMHeadCE.apply(h, W0, torch.full_like(labels, -100), weight, nrows)
    loss.backward()
    assert loss.item() == 0.0 and not h.grad.any()
