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
