"""N>1 path on CPU: 2 processes, gloo backend — the bucketed all-reduce must produce the mean of the per-rank
gradients for every trainable parameter, including parameters one rank never touched (unused-parameter hazard)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(16, 32)
        self.frozen = torch.nn.Linear(32, 32)
        self.b = torch.nn.Linear(32, 8)
        self.head_sometimes = torch.nn.Linear(8, 4)      # only rank 0 uses it
        self.frozen.requires_grad_(False)

    def forward(self, x, use_head):
        y = self.b(torch.relu(self.frozen(torch.relu(self.a(x)))))
        return self.head_sometimes(y).sum() if use_head else y.sum()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from mmmm_amd.ddp import BucketedGradAllReduce
    torch.manual_seed(0)
    net = Net()
    ddp = BucketedGradAllReduce(net.parameters(), bucket_bytes=1024)     # force several buckets
    assert len(ddp.buckets) >= 3
    out = {}
    for step in range(2):
        ddp.zero_grad()
        torch.manual_seed(100 + rank + 10 * step)
        x = torch.randn(5, 16)
        net(x, use_head=(rank == 0)).backward()
        ddp.finish()
        out[step] = {n: p.grad.detach().numpy().copy() for n, p in net.named_parameters() if p.requires_grad}
    q.put((rank, out))   # numpy payload: no torch shared-memory FD passing across processes
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_bucketed_allreduce_world2_matches_mean_of_rank_grads():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=100) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    # single-process reference: mean over ranks of the local gradients
    for step in range(2):
        ref = None
        for rank in range(world):
            torch.manual_seed(0)
            net = Net()
            torch.manual_seed(100 + rank + 10 * step)
            x = torch.randn(5, 16)
            net(x, use_head=(rank == 0)).backward()
            g = {n: (p.grad if p.grad is not None else torch.zeros_like(p)) for n, p in net.named_parameters() if p.requires_grad}
            ref = g if ref is None else {n: ref[n] + g[n] for n in g}
        ref = {n: v / world for n, v in ref.items()}
        for rank in range(world):
            for n, v in ref.items():
                torch.testing.assert_close(torch.from_numpy(res[rank][step][n]), v, rtol=1e-6, atol=1e-7)
        # both ranks hold identical reduced gradients
        for n in ref:
            assert (res[0][step][n] == res[1][step][n]).all()


def test_world1_keeps_grads_in_flat_buckets():
    from mmmm_amd.ddp import BucketedGradAllReduce
    net = Net()
    ddp = BucketedGradAllReduce(net.parameters(), world_size=1)
    ddp.zero_grad()
    net(torch.randn(3, 16), True).backward()
    ddp.finish()
    for p in net.parameters():
        if p.requires_grad:
            b = ddp.buckets[ddp._bucket_of[id(p)]].buffer
            assert b.data_ptr() <= p.grad.data_ptr() < b.data_ptr() + b.numel() * b.element_size()
    for p in net.parameters():
        p.grad = None          # optimizer.zero_grad(set_to_none=True)
    ddp.zero_grad()
    assert all(p.grad is not None for p in net.parameters() if p.requires_grad)


def test_clip_grad_norm_on_buckets_matches_torch():
    from mmmm_amd.ddp import BucketedGradAllReduce
    torch.manual_seed(1)
    net = Net()
    ddp = BucketedGradAllReduce(net.parameters(), world_size=1, bucket_bytes=2048)
    ddp.zero_grad()
    (net(torch.randn(7, 16), True) * 30).backward()
    ddp.finish()
    ref = [p.grad.clone() for p in net.parameters() if p.requires_grad]
    ref_norm = torch.nn.utils.clip_grad_norm_(ref_params := [torch.nn.Parameter(torch.zeros_like(g)) for g in ref], 1e9)
    for rp, g in zip(ref_params, ref):
        rp.grad = g.clone()
    ref_norm = torch.nn.utils.clip_grad_norm_(ref_params, 1.0)
    got = ddp.clip_grad_norm_(1.0)
    torch.testing.assert_close(got, ref_norm, rtol=1e-5, atol=1e-6)
    for p, rp in zip([p for p in net.parameters() if p.requires_grad], ref_params):
        torch.testing.assert_close(p.grad, rp.grad, rtol=1e-5, atol=1e-7)


# ------------------------------------------------------------------------------------------------------------------
# Direct accumulation (functional._lora_wgrad and friends): the op adds its weight gradient straight into the bucket view,
# hands autograd None and calls p._vm_grad_ready(p) — once per USE of the parameter. A parameter used several times per step
# (iSAM's box / discriminator heads: once per sample) must be counted once, after its LAST use.
class _DirectLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x)
        ctx.w = w
        return x @ w.t()

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        w = ctx.w
        w.grad += dy.t() @ x                 # straight into the flat bucket
        w._vm_grad_ready(w)                  # "this use is done" — NOT "the slot is final"
        return dy @ w.detach(), None


class MultiUseNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.early = torch.nn.Linear(8, 8)
        self.shared = torch.nn.Parameter(torch.randn(8, 8) * 0.3)      # used 4x per step through the direct path
        self.late = torch.nn.Linear(8, 8)

    def forward(self, xs):
        out = 0
        for x in xs:                         # "one call per sample"
            out = out + self.late(_DirectLinear.apply(self.early(x), self.shared)).sum()
        return out

    def forward_ref(self, xs):
        out = 0
        for x in xs:
            out = out + self.late(self.early(x) @ self.shared.t()).sum()
        return out


def _multiuse_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from mmmm_amd.ddp import BucketedGradAllReduce
    torch.manual_seed(0)
    net = MultiUseNet()
    ddp = BucketedGradAllReduce(net.parameters(), bucket_bytes=64)       # every parameter its own bucket
    launches = []
    orig = ddp._launch
    ddp._launch = lambda b: (launches.append((ddp.buckets.index(b), b.pending, len(ddp._ready))), orig(b))[1]
    out = {}
    for step in range(2):
        ddp.zero_grad()
        torch.manual_seed(7 + rank + 10 * step)
        xs = [torch.randn(3, 8) for _ in range(4)]
        net(xs).backward()
        ddp.finish()
        out[step] = {n: p.grad.detach().numpy().copy() for n, p in net.named_parameters()}
    q.put((rank, out, launches))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_parameter_reporting_ready_several_times_is_reduced_after_its_last_use():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_multiuse_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=100) for _ in range(world)]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    res = {r: o for r, o, _ in got}
    for _, _, launches in got:
        assert all(pending <= 0 for _, pending, _ in launches), 'a bucket was launched with unreported slots'
    for step in range(2):
        ref = None
        for rank in range(world):
            torch.manual_seed(0)
            net = MultiUseNet()
            torch.manual_seed(7 + rank + 10 * step)
            xs = [torch.randn(3, 8) for _ in range(4)]
            net.forward_ref(xs).backward()
            g = {n: p.grad for n, p in net.named_parameters()}
            ref = g if ref is None else {n: ref[n] + g[n] for n in g}
        ref = {n: v / world for n, v in ref.items()}
        for rank in range(world):
            for n, v in ref.items():
                torch.testing.assert_close(torch.from_numpy(res[rank][step][n]), v, rtol=1e-5, atol=1e-6)
        for n in ref:
            assert (res[0][step][n] == res[1][step][n]).all()


def test_multiuse_parameter_counts_once_single_rank():
    from mmmm_amd.ddp import BucketedGradAllReduce
    torch.manual_seed(0)
    net = MultiUseNet()
    ddp = BucketedGradAllReduce(net.parameters(), world_size=1, bucket_bytes=64)
    seen = []
    orig = ddp._mark_ready
    ddp._mark_ready = lambda p: (seen.append(id(p)), orig(p))[1]
    for h in ddp._hooks:
        h.remove()
    ddp._hooks = [p.register_post_accumulate_grad_hook(ddp._mark_ready) for p in ddp.params]
    ddp.zero_grad()
    net([torch.randn(3, 8) for _ in range(4)]).backward()
    b = ddp.buckets[ddp._bucket_of[id(net.shared)]]
    assert b.pending == 0 and id(net.shared) in ddp._ready
    ddp.finish()
    assert all(bk.pending == 0 for bk in ddp.buckets)


def test_no_weight_decay_parameters_get_their_own_buckets():
    from mmmm_amd.ddp import BucketedGradAllReduce
    from mmmm_amd.param import NoWeightDecayParameter
    m = torch.nn.Module()
    m.a = torch.nn.Parameter(torch.zeros(10))
    m.gain = NoWeightDecayParameter(torch.ones(10))
    m.b = torch.nn.Parameter(torch.zeros(10))
    ddp = BucketedGradAllReduce(m.parameters(), world_size=1)
    assert sorted((b.decay, len(b.params)) for b in ddp.buckets) == [(False, 1), (True, 2)]
    assert ddp.buckets[ddp._bucket_of[id(m.gain)]].decay is False


# ------------------------------------------------------------------------------------------------------------------
# replica consistency at construction (what DDPStrategy's module wrap does, conf/phase-vg/fit.yaml:11-15)
def _sync_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from mmmm_amd.ddp import BucketedGradAllReduce
    torch.manual_seed(rank)                 # every rank draws DIFFERENT initial weights
    net = Net()
    net.b.weight.data = net.b.weight.data.to(torch.bfloat16)       # two dtypes -> two broadcast groups
    before = net.a.weight.detach().clone()
    ddp = BucketedGradAllReduce(net.parameters(), bucket_bytes=1024)
    state = {n: p.detach().float().numpy().copy() for n, p in net.named_parameters() if p.requires_grad}
    changed = not torch.equal(before, net.a.weight.detach())
    ddp.assert_replicas_equal()
    # a rank that drifts is caught
    caught = False
    if rank == 1:
        with torch.no_grad():
            net.a.bias[3] += 1e-3
    try:
        ddp.assert_replicas_equal()
    except RuntimeError as e:
        caught = 'diverged' in str(e)
    # a second backward pass after a bucket was reduced must raise instead of corrupting the sum
    net2 = Net()
    ddp2 = BucketedGradAllReduce(net2.parameters(), bucket_bytes=1 << 20)
    ddp2.zero_grad()
    net2(torch.randn(3, 16), True).backward()
    second = False
    try:
        net2(torch.randn(3, 16), True).backward()
    except RuntimeError as e:
        second = 'one backward pass per finish' in str(e)
    ddp2.finish()
    q.put((rank, state, changed, caught, second))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_construction_broadcasts_rank0_parameters_and_checks_replicas():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_sync_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {r: rest for r, *rest in (q.get(timeout=100) for _ in range(world))}
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    torch.manual_seed(0)
    ref = Net()
    for n, p in ref.named_parameters():
        if p.requires_grad:
            want = p.detach().to(torch.bfloat16).float().numpy() if n == 'b.weight' else p.detach().numpy()
            for r in range(world):
                assert (got[r][0][n] == want).all(), (r, n)          # everybody holds rank 0's draw, bit for bit
    assert got[0][1] is False and got[1][1] is True                   # rank 1's weights were overwritten
    assert all(got[r][2] for r in range(world)), 'a drifted replica must be detected on every rank'
    assert all(got[r][3] for r in range(world)), 'a second backward pass with collectives on must raise'


def test_abort_step_clears_side_accumulators_and_parked_tensors():
    import mmmm_amd.functional as Fh
    from mmmm_amd.ddp import BucketedGradAllReduce
    net = Net()
    net.a.weight.data = net.a.weight.data.to(torch.bfloat16)
    net.a.bias.data = net.a.bias.data.to(torch.bfloat16)
    ddp = BucketedGradAllReduce(net.parameters(), world_size=1)
    acc = ddp.f32_accumulator(net.a.bias)
    acc += 3.0                                   # partial sums of a pass that then raised
    Fh._HELD.append(torch.zeros(4))
    Fh._HELD_TASK[0] = 12345                     # ... inside graph task 12345, whose final callbacks were dropped
    for b in ddp.buckets:
        b.buffer.fill_(1)
    ddp._ready.add(id(net.a.bias))
    ddp.abort_step()
    assert float(acc.abs().sum()) == 0 and not Fh._HELD and Fh._HELD_TASK[0] is None
    assert all(float(b.buffer.float().abs().sum()) == 0 and b.pending == len(b.params) for b in ddp.buckets) and not ddp._ready


def test_parked_tensors_of_a_failed_backward_do_not_leak_into_the_next_pass():
    """the engine drops its final callbacks when a backward pass raises; the next pass must start from an empty list"""
    import mmmm_amd.functional as Fh

    class Park(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, fail):
            ctx.fail = fail
            return x * 2

        @staticmethod
        def backward(ctx, g):
            Fh._hold_until_backward_ends(g)
            if ctx.fail:
                raise RuntimeError('boom')
            return g * 2, None

    Fh._release_held()
    x = torch.ones(3, requires_grad=True)
    with pytest.raises(RuntimeError, match='boom'):
        Park.apply(Park.apply(x, True), False).sum().backward()
    assert len(Fh._HELD) >= 1                    # the raised pass never ran its callback
    Park.apply(Park.apply(x, False), False).sum().backward()
    assert not Fh._HELD and Fh._HELD_TASK[0] is None
