"""Every BASELINE config at FULL size on the HIP path — configs[1] phase-vg 448x448, configs[2] phase-vlm mixed 2-D / 3-D batch with
variable text length, configs[3] phase-grg 32x256x256, configs[4] model-hr 896x896 and 64x384x384 (test_every_*_config_* below).
Description of the first and most detailed of them — BASELINE configs[1] at FULL size (CogVLM-7B, 32 decoder + 63 ViT-E layers, SAM-B + iSAM unfrozen, LoRA r64 with dropout, bf16;
batch 2 to keep the test short): the oracle cannot run this size, so parity is checked through size-independent properties:

  * replay: the same step twice from the same state gives the same loss to 2e-6 (usually bit-identical; dropout masks are a pure function of
    (seed, step, site, element); the only order-dependent sums of the forward are the fp32-atomic split-K products of the
    hyper-network MLPs in the heads). Gradients: the backward contains order-dependent fp32 atomics
    (the split-K weight gradients and the norm layers' column sums in the heads: 1e-7 relative; the trilinear up-sampling backward
    is a deterministic gather since round 2, upsample.hip), and a random-init network of this depth is
    chaotic in backward — every attention layer with saturated random-init softmax amplifies the noise ~100x (measured with
    tools/debug_replay.py: 0 at lm_head, 5e-7 at vg_proj, 6e-5 in the last decoder MLP, 2e-3 one attention layer further
    down, 1e-2 .. 5e-2 through the ViT, where |grad| reaches 1e10). The tight comparison is therefore made upstream of the
    first attention backward — lm_head, final norm, last decoder MLP, vg_proj, both grounding heads — and the other 1.7k
    parameters are only required to agree loosely. (This replay is also what exposed a real race: the factor-gradient kernels
    on the side stream read `dy` while autograd accumulated into it in place; see functional._off_critical_path.)
  * recompute: activation checkpointing of every layer (the reference's mode) versus keeping every activation (the HBM-budgeted
    mode of the benchmark) gives the same loss and the same gradients under the same comparison;
  * permutation: swapping the two samples of the batch leaves the loss unchanged up to bf16 summation order."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(dev, workload):
    import bench
    from mmmm_amd.ddp import BucketedGradAllReduce
    w = bench.WORKLOADS[workload]
    model, tok = bench.build(w, dev, 1.0)
    trainable = [p for p in model.parameters() if p.requires_grad]
    ddp = BucketedGradAllReduce(trainable, world_size=1)
    batch = bench.make_batch(w, tok, 2, dev, seed=11)
    model._test_tok = tok
    return model, ddp, batch


def _batch_for(model, workload, dev, seed=11):
    """a batch of two samples of another workload for an already built model (the architecture is the same for every config:
    only image / token shapes differ)"""
    import bench
    return bench.make_batch(bench.WORKLOADS[workload], model._test_tok, 2, dev, seed=seed)


@pytest.fixture
def full(dev):
    model, ddp, batch = _build(dev, 'phase-vg-448')
    yield model, ddp, batch
    ddp.remove()
    del model, ddp, batch
    torch.cuda.empty_cache()


@pytest.fixture
def full_vlm(dev):
    model, ddp, batch = _build(dev, 'phase-vlm-448')
    yield model, ddp, batch
    ddp.remove()
    del model, ddp, batch
    torch.cuda.empty_cache()


def run_step(model, ddp, batch, step_no: int, budget):
    from mmmm_amd.models.lora import ActivationBudget, StepState
    ActivationBudget.limit = budget
    StepState.step = step_no          # training_step advances it: pin it so that two runs draw the same dropout masks
    ddp.zero_grad()
    loss = model.training_step(batch)
    loss.backward()
    ddp.finish()
    torch.cuda.synchronize()
    return loss.detach().clone(), {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.requires_grad}


TIGHT = ('lm_head.', 'model.norm.', 'vg_proj.', 'sam.', 'isam_model.')
NEAR = ('model.layers.31.mlp.',)           # one bf16 MLP below the final norm: still before the first attention backward


def same_gradients(got, want):
    for n, w in want.items():
        nb = w.norm().item()
        if nb < 1e-6:
            continue            # analytically zero gradients (e.g. attention key biases): pure cancellation noise
        if n.startswith(TIGHT):
            d = got[n] - w
            err = (d.norm() / nb).item()
            if err >= 1e-3:
                # The heads' forward has fp32 split-K atomics (1e-7 noise between replays). A ReLU unit of a mask-decoder MLP whose
                # pre-activation sits within that noise of zero flips between two replays and takes its row of lin1.weight / column
                # of lin2.weight / bias entry with it (seen: model-hr-3d, unit 1972 of sam.mask_decoder.transformer.layers.1.mlp —
                # ONE row of 2048 off, everything else equal to 1e-6). Allow two such units; everything else stays tight.
                assert err < 3e-2, (n, err)
                d = d.clone()
                if d.dim() == 2:
                    d[d.norm(dim=1).topk(2).indices] = 0
                    d[:, d.norm(dim=0).topk(2).indices] = 0
                else:
                    d.view(-1)[d.abs().view(-1).topk(2).indices] = 0
                err = (d.norm() / nb).item()
            assert err < 1e-3, (n, err)
        elif n.startswith(NEAR):
            assert ((got[n] - w).norm() / nb).item() < 1e-2, n
        else:       # chaotic region: same scale, nothing stronger can be asserted about a replay
            assert 0.5 < got[n].norm().item() / nb < 2.0, n


def same_loss(a, b):
    """equal up to the order of the fp32 atomics of the small split-K products in the heads' forward (1e-7 relative)"""
    return abs(a.item() - b.item()) <= 2e-6 * abs(b.item())


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _permuted(batch):
    perm = {k: ([v[1], v[0]] if isinstance(v, list) else v) for k, v in batch.items()}
    perm['vlm_inputs'] = {k: v.flip(0) for k, v in batch['vlm_inputs'].items()}
    perm['host'] = dict(input_ids=batch['host']['input_ids'].flip(0), index_offsets=batch['host']['index_offsets'][::-1])
    return perm


def _without_dropout(model):
    import contextlib

    @contextlib.contextmanager
    def ctx():
        cfgs = {id(m.lora_cfg): m.lora_cfg for m in model.modules() if getattr(m, 'lora_cfg', None) is not None}
        saved = {k: c.lora_dropout for k, c in cfgs.items()}
        for c in cfgs.values():
            c.lora_dropout = 0.0
        try:
            yield
        finally:
            for k, c in cfgs.items():
                c.lora_dropout = saved[k]
    return ctx()


@pytest.mark.parametrize('workload', ['phase-grg-3d', 'model-hr-2d', 'model-hr-3d'])
def test_every_grounding_config_at_full_size(dev, full, workload):
    """BASELINE configs[3] (phase-grg: 3-D CT 32x256x256, SAM + iSAM unfrozen) and configs[4] (model-hr: 896x896 2-D and 64x384x384
    3-D; bf16 here) through the full-depth HIP path, batch 2 (one semantic, one instance sample): finite loss and gradients for
    every trainable parameter, replay, recompute-every-layer == keep-within-budget, sample permutation. What these shapes add
    over configs[1]: 2 049 / 3 137 / 4 609-token ViT sequences, the z-folded patch embedding (pz 4 / 8), 3-D SAM grids
    ([8,16,16] and [8,24,24]) with the z-collapsing up-sampler, position tables resampled to non-cubic grids."""
    model, ddp, _ = full
    batch = _batch_for(model, workload, dev)
    budget = 120 << 30
    l0, g0 = run_step(model, ddp, batch, 7, budget)
    assert torch.isfinite(l0) and all(torch.isfinite(g).all() for g in g0.values())
    assert all(g0[n].norm() > 0 for n in g0 if n.startswith(('lm_head.', 'vg_proj.')))
    l1, g1 = run_step(model, ddp, batch, 7, budget)
    assert same_loss(l0, l1)
    same_gradients(g1, g0)
    l3, g3 = run_step(model, ddp, batch, 7, None)            # the reference's mode: every layer recomputed
    assert same_loss(l0, l3)
    same_gradients(g3, g0)
    with _without_dropout(model):
        a, _ = run_step(model, ddp, batch, 7, budget)
        b, _ = run_step(model, ddp, _permuted(batch), 7, budget)
    assert abs(a.item() - b.item()) / abs(a.item()) < 2e-3, (a.item(), b.item())


def test_mixed_2d_3d_config_at_full_size(dev, full_vlm):
    """BASELINE configs[2] (phase-vlm: one 2-D 448x448 and one 3-D 32x256x256 sample in the same batch, text lengths drawn from
    128..512, sam = None): ragged ViT sequences (785 + 2 049 tokens) in one block-diagonal attention, two patch-embedding
    geometries and two pooling shapes in one forward, right-padded decoder rows. Without the grounding heads the step is
    bit-reproducible: replay and recompute-every-layer are checked to the last bit (norm / patch-embedding atomics excepted)."""
    model, ddp, _ = full_vlm
    batch = _batch_for(model, 'phase-vlm-mixed', dev)
    vi = batch['vlm_inputs']
    assert not bool(vi['attention_mask'].all()) and batch['image'][0].shape[1] == 1 and batch['image'][1].shape[1] == 32
    keep_all = 200 << 30
    l0, g0 = run_step(model, ddp, batch, 3, keep_all)
    assert torch.isfinite(l0) and all(torch.isfinite(g).all() for g in g0.values()) and len(g0) > 1400
    for what, budget in (('replay', keep_all), ('every layer recomputed', None)):
        l, g = run_step(model, ddp, batch, 3, budget)
        assert torch.equal(l, l0), what
        diff = [n for n in g0 if not torch.equal(g[n], g0[n])]
        atomics = ('norm', 'patch_embedding.')
        assert all(any(a in n for a in atomics) for n in diff), (what, [n for n in diff if not any(a in n for a in atomics)][:5])
        for n in diff:
            assert rel(g[n], g0[n]) < 2e-3, (what, n)
    with _without_dropout(model):
        a, _ = run_step(model, ddp, batch, 3, keep_all)
        perm = _permuted(batch)
        b, _ = run_step(model, ddp, perm, 3, keep_all)
    assert abs(a.item() - b.item()) / abs(a.item()) < 2e-3, (a.item(), b.item())


def test_full_batch_of_eight_matches_four_pairs(dev, full_vlm):
    """configs[1] at the benchmark's batch size (8 per GPU; the other tests of this file use 2): the weighted CE is a mean over the
    valid target tokens, so the loss of the batch of eight equals the token-weighted mean of its four pairs (dropout off)."""
    import bench
    model, ddp, _ = full_vlm
    big = bench.make_batch(bench.WORKLOADS['phase-vlm-448'], model._test_tok, 8, dev, seed=21)
    with _without_dropout(model):
        l8, _ = run_step(model, ddp, big, 5, 200 << 30)
        num = den = 0.0
        for i in range(0, 8, 2):
            sub = {k: (v[i:i + 2] if isinstance(v, list) else v) for k, v in big.items()}
            sub['vlm_inputs'] = {k: v[i:i + 2] for k, v in big['vlm_inputs'].items()}
            sub['host'] = dict(input_ids=big['host']['input_ids'][i:i + 2], index_offsets=big['host']['index_offsets'][i:i + 2])
            l2, _ = run_step(model, ddp, sub, 5, 200 << 30)
            n = int((sub['vlm_inputs']['labels'] >= 0).sum())
            num, den = num + l2.item() * n, den + n
    print('batch of 8:', l8.item(), 'token-weighted mean of its pairs:', num / den)
    assert abs(l8.item() - num / den) / abs(l8.item()) < 1e-2, (l8.item(), num / den)


def test_full_size_step_properties(dev, full):
    model, ddp, batch = full
    keep_all = 200 << 30
    l0, g0 = run_step(model, ddp, batch, 7, keep_all)
    assert torch.isfinite(l0) and all(torch.isfinite(g).all() for g in g0.values())
    assert sum(1 for n in g0 if n.startswith(TIGHT)) >= 20 and all(g0[n].norm() > 0 for n in g0 if n.startswith('lm_head.'))
    # replay
    l1, g1 = run_step(model, ddp, batch, 7, keep_all)
    assert same_loss(l0, l1)
    same_gradients(g1, g0)
    # a different step number draws different dropout masks
    l2, _ = run_step(model, ddp, batch, 8, keep_all)
    assert not same_loss(l0, l2) and abs(l2.item() - l0.item()) / abs(l0.item()) < 0.05
    # recompute everything (reference mode) vs keep everything
    l3, g3 = run_step(model, ddp, batch, 7, None)
    assert same_loss(l0, l3)
    same_gradients(g3, g0)
    # sample permutation
    perm = {k: ([v[1], v[0]] if isinstance(v, list) else v) for k, v in batch.items()}
    perm['vlm_inputs'] = {k: v.flip(0) for k, v in batch['vlm_inputs'].items()}
    perm['host'] = dict(input_ids=batch['host']['input_ids'].flip(0), index_offsets=batch['host']['index_offsets'][::-1])
    # (dropout masks depend on the element index, so the two orders draw different masks: compare with dropout off)
    cfgs = {id(m.lora_cfg): m.lora_cfg for m in model.modules() if getattr(m, 'lora_cfg', None) is not None}
    saved = {k: c.lora_dropout for k, c in cfgs.items()}
    for c in cfgs.values():
        c.lora_dropout = 0.0
    try:
        a, _ = run_step(model, ddp, batch, 7, keep_all)
        b, _ = run_step(model, ddp, perm, 7, keep_all)
    finally:
        for k, c in cfgs.items():
            c.lora_dropout = saved[k]
    assert abs(a.item() - b.item()) / abs(a.item()) < 2e-3, (a.item(), b.item())


def test_full_size_backward_is_bit_reproducible_without_the_heads(dev, full_vlm):
    """phase-vlm (no grounding heads, hence none of their atomics) at full size: the backward of the 7B decoder + ViT-E is
    bit-reproducible, so three things that must not change results can be checked to the last bit on 1.4k parameter gradients —
    a replay, the weight-gradient kernels moved between the main stream and the side stream, and every layer recomputed instead of
    kept. Only the norm weights / biases (their column sums use fp32 atomics: 1e-5 .. 1e-4 after bf16 rounding) and the patch
    embedding (ATen's atomic conv-weight / interpolate backward) may differ."""
    import mmmm_amd.functional as Fh
    model, ddp, batch = full_vlm
    keep_all = 200 << 30
    l0, g0 = run_step(model, ddp, batch, 3, keep_all)

    def check(l, g, what):
        assert torch.equal(l, l0), what
        diff = [n for n in g0 if not torch.equal(g[n], g0[n])]
        atomics = ('norm', 'patch_embedding.')      # norm column sums (ours, fp32 atomics); ATen conv / interpolate backward of the patch embedding
        assert all(any(a in n for a in atomics) for n in diff), (what, [n for n in diff if not any(a in n for a in atomics)][:5])
        for n in diff:
            assert rel(g[n], g0[n]) < 2e-3, (what, n)
        return len(diff)

    check(*run_step(model, ddp, batch, 3, keep_all), 'replay')
    # the weight-gradient side stream (off by default since round 3) and the grouped factor-gradient launches, each toggled
    prev = Fh.WGRAD_SIDE_STREAM
    Fh.WGRAD_SIDE_STREAM = not prev
    try:
        check(*run_step(model, ddp, batch, 3, keep_all), f'weight-gradient side stream {"on" if not prev else "off"}')
    finally:
        Fh.WGRAD_SIDE_STREAM = prev
    check(*run_step(model, ddp, batch, 3, None), 'every layer recomputed')
    assert len(g0) > 1400


def test_full_size_generation_properties(dev, full_vlm):
    """generation path at full size (7B decoder, 456-token prompts with images, batch 2): the hipGraph-replayed decode loop
    reproduces the eager loop bit for bit, and the logits of every cached decode step equal the last row of an uncached forward
    over the grown sequence up to bf16 evaluation-order drift (flat in t: 4 % at this depth, measured)"""
    model, ddp, batch = full_vlm
    vi = batch['vlm_inputs']
    kw = dict(token_type_ids=vi['token_type_ids'], position_ids=vi['position_ids'], attention_mask=vi['attention_mask'],
              image=batch['image'], patch_size=batch['patch_size'], pool_size=batch['pool_size'])
    steps = 6
    eager = model.generate(vi['input_ids'], **kw, max_new_tokens=steps, return_logits=True)
    graphed = model.generate(vi['input_ids'], **kw, max_new_tokens=steps, use_graph=True)
    assert torch.equal(eager.new_tokens, graphed.new_tokens) and torch.equal(eager.new_position_ids, graphed.new_position_ids)
    assert model.training                                    # generate() restores the mode it found
    seq = eager.sequences(vi['input_ids'])
    n = vi['input_ids'].shape[1]
    assert bool(vi['attention_mask'].all())                  # the synthetic prompts of this workload are unpadded
    model.eval()
    try:
        with torch.no_grad():
            full = model(seq[:, :-1], token_type_ids=torch.cat([vi['token_type_ids'], torch.zeros_like(eager.new_tokens[:, :-1])], 1),
                         position_ids=torch.cat([vi['position_ids'], eager.new_position_ids[:, :-1]], 1),
                         image=batch['image'], patch_size=batch['patch_size'], pool_size=batch['pool_size'])
    finally:
        model.train()
    errs = [rel(eager.logits[t], full.logits[:, n - 1 + t]) for t in range(steps)]
    agree = [bool((eager.logits[t].argmax(-1) == full.logits[:, n - 1 + t].argmax(-1)).all()) for t in range(steps)]
    # 32 layers of bf16 with random-init (saturated) attention: two evaluation orders of the same math (skinny streaming linears
    # + single-query attention vs tiled GEMMs + flash attention) drift apart ~3x further than at the 2-layer depth of
    # tests/test_generate_gpu.py (1.2e-2); a cache-indexing error would show as O(1) and grow with t
    assert errs[0] < 1e-4 and max(errs) < 8e-2 and errs[-1] < 2 * errs[1] + 1e-2, errs
    assert sum(agree) >= steps - 2, agree
