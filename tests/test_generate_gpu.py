"""Generation path on the MI355X: decode attention kernel, KV cache, the <p>/</p> position rule and the
greedy loop, through the C ABI, against the CPU oracle (oracle/vividmed.py::generate, itself pinned to the reference by
fixture f12). Tolerances as in test_model_gpu.py: bf16 product vs fp32 oracle on the same bf16-rounded weights, logits
<= 2e-2 relative L2; position ids and forced-token bookkeeping are integer work and bit-exact."""
import pytest
import torch

from tests._gpu_common import cpu, oracle_cfg, oracle_state, rel
from tests.test_model_gpu import lm, make_inputs  # noqa: F401  (module fixture)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def K():
    from mmmm_amd import kernels
    return kernels


def ref_decode_attention(q, kc, vc, lens, H, hd, scale):
    """fp32 restatement of modeling_cogvlm.py:129-141 on the valid cache rows, with the reference's rounding of q and scores"""
    B = q.shape[0]
    out = torch.zeros(B, H * hd)
    for b in range(B):
        n = int(lens[b])
        qb = (q[b].float().view(H, hd) * scale).bfloat16().float()
        kb = kc[b, :n].float().view(n, H, hd)
        vb = vc[b, :n].float().view(n, H, hd)
        s = torch.einsum('hd,nhd->nh', qb, kb).bfloat16().float()
        p = s.softmax(0)
        out[b] = torch.einsum('nhd,nh->hd', vb, p).reshape(-1)
    return out


@pytest.mark.parametrize('hd', [32, 64, 128])
def test_attn_decode_kernel(dev, K, hd):
    g = torch.Generator().manual_seed(hd)
    B, H, Lmax = 5, 3, 400
    lens = torch.tensor([1, 32, 33, 129, 400], dtype=torch.int32)      # chunk edges (32 keys per wave)
    q = torch.randn(B, 3 * H * hd, generator=g).bfloat16()[:, :H * hd]          # strided rows, like the q third of a packed qkv
    kc = torch.randn(B, Lmax, H * hd, generator=g).bfloat16()
    vc = torch.randn(B, Lmax, H * hd, generator=g).bfloat16()
    scale = hd ** -0.5
    want = ref_decode_attention(q, kc, vc, lens, H, hd, scale)
    kd, vd = kc.to(dev), vc.to(dev)
    # rows past the length must never be read as data: poison them
    for b in range(B):
        kd[b, int(lens[b]):] = float('nan')
        vd[b, int(lens[b]):] = float('nan')
    got = K.attn_decode(q.to(dev), kd, vd, lens.to(dev), H, hd, scale, Lmax)
    assert torch.isfinite(got).all()
    assert rel(got.float(), want) < 6e-3
    # a tighter launch bound gives the same bits (chunk partials beyond the length are neutral)
    got2 = K.attn_decode(q.to(dev)[:4], kd[:4], vd[:4], lens[:4].to(dev), H, hd, scale, 129)
    assert torch.equal(got2, got[:4])


@pytest.mark.parametrize('M,N,Kd', [(1, 4096, 4096), (8, 1000, 704), (16, 264, 1152), (3, 16, 64)])
def test_gemv_kernel(dev, K, M, N, Kd):
    """skinny-M linear of the decode step vs fp32 torch: plain, with LoRA extension, bias and residual (torch's rounding:
    the linear is rounded to bf16 before the residual add); N tails, odd block counts, M = 1 .. 16; bit-reproducible"""
    g = torch.Generator().manual_seed(M * 1000 + N)
    x = torch.randn(M, Kd, generator=g).bfloat16().to(dev)
    w = (torch.randn(N, Kd, generator=g) / Kd ** 0.5).bfloat16().to(dev)
    t = torch.randn(M, 64, generator=g).bfloat16().to(dev)
    b2 = (torch.randn(N, 64, generator=g) / 8).bfloat16().to(dev)
    bias = torch.randn(N, generator=g).bfloat16().to(dev)
    res = torch.randn(M, N, generator=g).bfloat16().to(dev)
    base = x.float() @ w.float().T
    assert rel(K.gemv(x, w).float(), base) < 4e-3
    full = base + 0.5 * (t.float() @ b2.float().T) + bias.float()
    want = full.bfloat16().float() + res.float()
    got = K.gemv(x, w, a2=t, b2=b2, alpha2=0.5, bias=bias, residual=res)
    assert rel(got.float(), want) < 4e-3
    assert torch.equal(got, K.gemv(x, w, a2=t, b2=b2, alpha2=0.5, bias=bias, residual=res))
    # same contraction as the tiled GEMM (different summation order only)
    assert rel(got.float(), K.gemm(x, w, a2=t, b2=b2, alpha2=0.5, bias=bias, residual=res).float()) < 4e-3
    # strided output / input rows
    big = torch.zeros(M, N + 24, dtype=torch.bfloat16, device=dev)
    K.gemv(x, w, out=big[:, 8:8 + N])
    assert rel(big[:, 8:8 + N].float(), base) < 4e-3 and big[:, :8].abs().sum() == 0 and big[:, 8 + N:].abs().sum() == 0


def _sample(batch, b):
    """sample b alone, cut to its valid length (the reference generates one unpadded sample at a time)"""
    vi = batch['vlm_inputs']
    n = int(vi['attention_mask'][b].sum())
    kw = {k: vi[k][b:b + 1, :n] for k in ('input_ids', 'token_type_ids', 'position_ids')}
    return kw, dict(image=batch['image'][b:b + 1], patch_size=batch['patch_size'][b:b + 1], pool_size=batch['pool_size'][b:b + 1])


def _forced(tok, steps, B, seed):
    g = torch.Generator().manual_seed(seed)
    f = torch.randint(3, 150, (B, steps), generator=g)
    f[:, 2] = tok.bop_token_id
    f[0, min(5, steps - 1)] = tok.eop_token_id
    f[1 % B, 3] = tok.eop_token_id          # </p> right after <p>: both rules at once
    return f


def test_generate_matches_oracle_per_sample(dev, lm):
    from oracle import vividmed as O
    batch, tok = make_inputs(dev)
    lm.tokenizer = tok
    sd, cfg = oracle_state(lm), oracle_cfg(lm.config)
    steps = 8
    forced = _forced(tok, steps, 3, 1)
    for b in range(3):
        kw, img = _sample(batch, b)
        out = lm.generate(**kw, **img, forced_tokens=forced[b:b + 1].to(dev), return_logits=True)
        seq, step_logits, pos = O.generate(sd, cfg, cpu(kw['input_ids']), token_type_ids=cpu(kw['token_type_ids']),
                                           position_ids=cpu(kw['position_ids']), image=[x.float() for x in cpu(img['image'])],
                                           patch_size=img['patch_size'], pool_size=img['pool_size'], bop_token_id=tok.bop_token_id,
                                           eop_token_id=tok.eop_token_id, max_new_tokens=steps, forced=forced[b:b + 1],
                                           rope_dtype=torch.bfloat16)
        n = kw['input_ids'].shape[1]
        assert torch.equal(out.new_position_ids.cpu(), pos[:, n:])                       # integer rule: bit-exact
        assert torch.equal(out.sequences(kw['input_ids']).cpu(), seq)
        assert len(out.logits) == steps and len(step_logits) == steps    # the oracle also decodes the last token
        for t in range(1, steps):       # logits[t] follow the decode step of token t-1
            assert rel(out.logits[t].cpu(), step_logits[t - 1]) < 2e-2, (b, t)


def test_batched_right_padded_generate_equals_per_sample(dev, lm):
    batch, tok = make_inputs(dev)
    lm.tokenizer = tok
    steps = 6
    forced = _forced(tok, steps, 3, 2).to(dev)
    vi = batch['vlm_inputs']
    both = lm.generate(vi['input_ids'], token_type_ids=vi['token_type_ids'], position_ids=vi['position_ids'],
                       attention_mask=vi['attention_mask'], image=batch['image'], patch_size=batch['patch_size'],
                       pool_size=batch['pool_size'], forced_tokens=forced, return_logits=True)
    assert both.prompt_lengths.tolist() == vi['attention_mask'].sum(1).tolist()
    for b in range(3):
        kw, img = _sample(batch, b)
        one = lm.generate(**kw, **img, forced_tokens=forced[b:b + 1], return_logits=True)
        assert torch.equal(one.new_position_ids[0], both.new_position_ids[b])
        for t in range(steps):
            assert rel(both.logits[t][b], one.logits[t][0]) < 1e-2, (b, t)
    # the cache holds valid tokens only
    assert both.past_key_values.lens.tolist() == [int(n) + steps - 1 for n in vi['attention_mask'].sum(1).tolist()]


def test_cached_decode_equals_uncached_forward(dev, lm):
    """size-independent property: the logits of a cached decode step equal the last row of a full forward over the sequence"""
    batch, tok = make_inputs(dev)
    lm.tokenizer = tok
    kw, img = _sample(batch, 2)
    steps = 5
    forced = _forced(tok, steps, 1, 3).to(dev)
    out = lm.generate(**kw, **img, forced_tokens=forced, return_logits=True)
    seq = out.sequences(kw['input_ids'])
    n = kw['input_ids'].shape[1]
    lm.eval()
    with torch.no_grad():
        full = lm(seq[:, :-1], token_type_ids=torch.cat([kw['token_type_ids'], torch.zeros_like(forced[:, :-1])], 1),
                  position_ids=torch.cat([kw['position_ids'], out.new_position_ids[:, :-1]], 1), **img)
    for t in range(steps):
        assert rel(out.logits[t][0], full.logits[0, n - 1 + t]) < 2e-2, t      # two bf16 evaluations: the bf16 bound of the suite


def test_greedy_generate_and_eos(dev, lm):
    batch, tok = make_inputs(dev)
    lm.tokenizer = tok
    kw, img = _sample(batch, 0)
    free = lm.generate(**kw, **img, max_new_tokens=6, return_logits=True)
    assert free.new_tokens.shape == (1, 6)
    for t in range(6):
        assert int(free.logits[t].argmax(-1)) == int(free.new_tokens[0, t])
    eos = int(free.new_tokens[0, 1])
    stopped = lm.generate(**kw, **img, max_new_tokens=6, eos_token_id=eos, eos_check_every=2)
    got = stopped.new_tokens[0].tolist()
    assert got[:2] == free.new_tokens[0, :2].tolist() and all(t == eos for t in got[1:])


def test_forward_with_past_key_values_like_the_reference_calls_it(dev, lm):
    """the reference's calling convention: forward(use_cache=True) -> past_key_values -> prepare_inputs_for_generation ->
    forward(past_key_values=...) with [B,1] inputs (mmmm.py:368-406)"""
    batch, tok = make_inputs(dev)
    lm.tokenizer = tok
    lm.eval()
    kw, img = _sample(batch, 1)
    with torch.no_grad():
        out = lm(**kw, **img, use_cache=True)
        assert out.logits.shape[:2] == kw['input_ids'].shape and len(out.past_key_values) > 0
        new = torch.tensor([[tok.bop_token_id]], device=dev)
        ids = torch.cat([kw['input_ids'], new], 1)
        tt = torch.cat([kw['token_type_ids'], torch.zeros_like(new)], 1)
        pos = torch.cat([kw['position_ids'], kw['position_ids'][:, -1:] + 1], 1)
        inputs = lm.prepare_inputs_for_generation(ids, token_type_ids=tt, position_ids=pos, past_key_values=out.past_key_values,
                                                  attention_mask=torch.ones_like(ids), patch_size=None, pool_size=None, use_cache=True)
        step = lm(**inputs)
    assert step.logits.shape == (1, 1, lm.config.vocab_size)
    ref = lm.generate(**kw, **img, forced_tokens=torch.tensor([[tok.bop_token_id, 5]], device=dev), return_logits=True)
    assert torch.equal(step.logits[:, 0], ref.logits[1])


def test_graph_replayed_decode_equals_eager(dev, lm):
    """the hipGraph-captured decode step (static buffers, device-side step counter) reproduces the eager loop bit for bit"""
    batch, tok = make_inputs(dev)
    lm.tokenizer = tok
    vi = batch['vlm_inputs']
    kw = dict(token_type_ids=vi['token_type_ids'], position_ids=vi['position_ids'], attention_mask=vi['attention_mask'],
              image=batch['image'], patch_size=batch['patch_size'], pool_size=batch['pool_size'])
    eager = lm.generate(vi['input_ids'], **kw, max_new_tokens=12)
    graphed = lm.generate(vi['input_ids'], **kw, max_new_tokens=12, use_graph=True)
    assert torch.equal(eager.new_tokens, graphed.new_tokens) and torch.equal(eager.new_position_ids, graphed.new_position_ids)
    assert graphed.past_key_values.lens.tolist() == eager.past_key_values.lens.tolist()
    eos = int(eager.new_tokens[0, 2])
    a = lm.generate(vi['input_ids'], **kw, max_new_tokens=12, eos_token_id=eos, eos_check_every=4)
    b = lm.generate(vi['input_ids'], **kw, max_new_tokens=12, eos_token_id=eos, eos_check_every=4, use_graph=True)
    assert torch.equal(a.new_tokens, b.new_tokens)
    assert (a.new_tokens[0, 2:] == eos).all()
