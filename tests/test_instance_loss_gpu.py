"""`vm_box_match_cost` / `vm_instance_loss_fwd,_bwd` against the element-wise torch form of the same reference functions
(InstanceSamLoss.box_loss / disc_loss / _match_instances / compute_loss, reference segvol/modeling/sam.py:148-361), which
tests/test_sam_gpu.py pins against the reference's own outputs (tests/golden/f7_losses.pt)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda', 0)


def _sample(counts, nq, seed, dev):
    g = torch.Generator().manual_seed(seed)
    offs, s = [], 0
    for c in counts:
        offs.append((s, s + c))
        s += c
    nt = len(counts)
    label = torch.cat([torch.rand(s, 3, generator=g) * 0.5 + 0.25, torch.rand(s, 3, generator=g) * 0.3 + 0.05], 1).to(dev)
    reg = torch.cat([torch.rand(nt, 1 + nq, 3, generator=g) * 0.5 + 0.25, torch.rand(nt, 1 + nq, 3, generator=g) * 0.3 + 0.05], -1).to(dev)
    logit = (torch.randn(nt, nq, generator=g) * 2).to(dev)
    return reg, logit, label, torch.tensor(offs, dtype=torch.int64).reshape(-1, 2)


def _loss(**kw):
    from mmmm_amd.models.segvol.modeling.sam import InstanceSamLoss
    args = dict(use_neg_mask=False, box_l1_weight=5, box_giou_weight=2, disc_weight=2, disc_focal_gamma=2, disc_focal_alpha=0.85)
    args.update(kw)
    return InstanceSamLoss(**args)


@pytest.mark.parametrize('match_ce', [True, False])
@pytest.mark.parametrize('alpha', [0.85, None])
def test_cost_matrices_of_several_samples_in_one_launch(dev, match_ce, alpha):
    from mmmm_amd import kernels as K
    il = _loss(match_ce=match_ce, disc_focal_alpha=alpha)
    samples = [_sample([0, 1, 3, 6, 9], 6, 1, dev), _sample([2, 0, 7], 6, 2, dev), _sample([0], 6, 3, dev), _sample([12], 6, 4, dev)]
    # the kernel's matrices, through the same descriptor table match_samples builds
    nq, desc, ref = 6, [], []
    for reg, logit, label, offs in samples:
        o = [tuple(x) for x in offs.tolist()]
        costs, metas = il._match_costs(reg[:, 1:], logit, label, o)
        for i, (s, e) in enumerate(o):
            if e > s:
                desc.append((reg.data_ptr() + (i * (nq + 1) + 1) * 24, logit.data_ptr() + i * nq * 4, label.data_ptr() + s * 24, e - s, max(nq, e - s), nq))
                ref.append(costs[metas[i][0]])
    width = max(d[4] for d in desc)
    cost = K.box_match_cost(torch.tensor(desc, dtype=torch.int64, device=dev), len(desc), nq, width, il.box_l1_weight, il.box_giou_weight,
                            il.disc_weight, il.match_ce, il.disc_focal_gamma, il.disc_focal_alpha)
    for c, r in zip(cost, ref):
        assert rel(c[:, :r.shape[1]], r) < 2e-6
        assert torch.all(c[:, r.shape[1]:] == 0)
    # and the assignment built from them equals the one built from the element-wise matrices
    fused = il.match_samples(samples)
    il.fused = False
    plain = il.match_samples(samples)
    assert all(a.is_cuda and torch.equal(a, b) for a, b in zip(fused, plain))


@pytest.mark.parametrize('counts,nq', [([0, 1, 3, 6, 9], 6), ([0, 0], 4), ([5, 7], 3), ([2], 40)])
@pytest.mark.parametrize('gamma,alpha', [(2.0, 0.85), (2.0, None), (0.0, 0.25), (1.5, 0.5)])
def test_fused_instance_loss_equals_the_elementwise_form(dev, counts, nq, gamma, alpha):
    il = _loss(disc_focal_gamma=gamma, disc_focal_alpha=alpha)
    reg, logit, label, offs = _sample(counts, nq, 7, dev)
    match = il._match_all(reg, logit, label, offs)
    dummy = reg.new_empty((len(counts), 1 + nq, 0, 0, 0))
    res = []
    for fused in (True, False):
        il.fused = fused                                    # (set per call)
        b, d = reg.clone().requires_grad_(), logit.clone().requires_grad_()
        loss, log = il.compute_loss(dummy, dummy, b, d, None, label, offs, match=match)
        (loss * 1.7).backward()
        res.append((loss.detach(), log, b.grad, d.grad))
    (l0, g0, b0, d0), (l1, g1, b1, d1) = res
    assert b0 is not None                               # the fused form writes every row of the box gradient
    assert rel(l0, l1) < 2e-6
    assert g0.keys() == g1.keys()
    for k in g0:
        assert rel(g0[k], g1[k]) < 3e-6, k
    assert rel(d0, d1) < 1e-5
    if b1 is not None and b1.abs().max() > 0:
        assert rel(b0, b1) < 1e-5
    else:                                               # nothing matched: the element-wise form never touches the boxes
        assert bool(torch.all(b0 == 0))
    assert bool(torch.all(b0[:, 0] == 0))                     # the semantic-box rows are not part of the loss


def test_subgradient_conventions_at_ties_follow_torch(dev):
    """predicted box == label box in some coordinates (min / max ties: torch splits the gradient), disjoint boxes (clamp at 0),
    equal entries (sign(0) = 0 in the l1 term)"""
    from mmmm_amd import functional as Fh
    il = _loss()
    label = torch.tensor([[0.5, 0.5, 0.5, 0.2, 0.2, 0.2], [0.25, 0.25, 0.25, 0.1, 0.1, 0.1]], device=dev)
    reg = torch.tensor([[[0.5, 0.5, 0.5, 0.5, 0.5, 0.5],      # semantic row
                         [0.5, 0.5, 0.5, 0.2, 0.2, 0.2],      # identical to label 0: ties everywhere
                         [0.5, 0.5, 0.75, 0.2, 0.2, 0.2],     # shares faces in x, y; disjoint in z
                         [0.25, 0.3, 0.25, 0.1, 0.2, 0.1]]], device=dev)   # vs label 1: contains it in y, equal faces in x, z
    match = torch.tensor([[0, 0, 1]], device=dev)
    logit = torch.tensor([[0.3, -1.2, 2.0]], device=dev)
    b = reg.clone().requires_grad_()
    out = Fh.instance_loss(logit, b, label, match, 2.0, 0.85)
    (5 * out[3] + 2 * out[4]).backward()
    b2 = reg.clone().requires_grad_()
    ref = il.box_loss(b2[:, 1:].reshape(-1, 6), label[match.flatten()], return_dict=True)
    ref['total'].backward()
    assert rel(out[3], ref['l1']) < 1e-6 and rel(out[4], ref['giou']) < 1e-6
    assert rel(b.grad, b2.grad) < 1e-5, (b.grad, b2.grad)
