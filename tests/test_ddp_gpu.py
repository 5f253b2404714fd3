"""N > 1 on the real model: two processes (gloo, both on the one GPU of the test box — RCCL needs one device per rank) run the tiny
MMMMForCausalLM through BucketedGradAllReduce + FlatAdamW on DIFFERENT batches. After every step both ranks must hold bit-identical
parameters (identical averaged gradients, identical updates), and the parameters must equal those of a single process that averages the
two ranks' gradients itself."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(dev):
    from tests.test_model_gpu import tiny_config
    from tests._gpu_common import randomize_
    from mmmm_amd.models.mmmm import MMMMForCausalLM, VisionArgs
    from mmmm_amd.models.lora import LoraConfig
    from mmmm_amd.utils import apply_lora
    torch.manual_seed(0)
    m = MMMMForCausalLM(tiny_config(), vision_override=VisionArgs(pos_embed_shape=(2, 2, 4), patch_size=(4, 8, 8)))
    apply_lora(m, LoraConfig(r=64, lora_alpha=8, lora_dropout=0.0, use_rslora=True))
    randomize_(m, 123)
    return m.to(dev).to(torch.bfloat16).train()


def _batch(dev, seed):
    from tests.test_model_gpu import make_inputs
    return make_inputs(dev, seed=seed)[0]


def _grads_of(m, batch):
    out = m(**batch['vlm_inputs'], image=batch['image'], patch_size=batch['patch_size'], pool_size=batch['pool_size'])
    out.loss.backward()
    return out.loss.detach()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from mmmm_amd.ddp import BucketedGradAllReduce
    from mmmm_amd.optim import FlatAdamW
    dev = torch.device('cuda:0')
    m = _model(dev)
    ddp = BucketedGradAllReduce([p for p in m.parameters() if p.requires_grad], bucket_bytes=1 << 20)
    assert ddp.world_size == 2 and ddp.collectives and len(ddp.buckets) > 2
    opt = FlatAdamW(ddp, lr=1e-3, weight_decay=0.01, max_grad_norm=1.0)
    losses = []
    for step in range(2):
        ddp.zero_grad()
        losses.append(float(_grads_of(m, _batch(dev, 100 + 10 * step + rank))))
        ddp.finish()
        opt.step()
    torch.cuda.synchronize()
    q.put((rank, losses, {n: p.detach().float().cpu().numpy() for n, p in m.named_parameters() if p.requires_grad}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_end_with_identical_parameters_equal_to_averaged_gradients(dev):
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict((r, (l, prm)) for r, l, prm in (q.get(timeout=500) for _ in range(world)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    p0, p1 = got[0][1], got[1][1]
    assert p0.keys() == p1.keys() and len(p0) > 40
    for n in p0:
        assert (p0[n] == p1[n]).all(), n                    # bit-identical on both ranks
    # single-process reference: average the two ranks' gradients by hand, same optimizer
    from mmmm_amd.ddp import BucketedGradAllReduce
    from mmmm_amd.optim import FlatAdamW
    m = _model(dev)
    ddp = BucketedGradAllReduce([p for p in m.parameters() if p.requires_grad], world_size=1, bucket_bytes=1 << 20)
    opt = FlatAdamW(ddp, lr=1e-3, weight_decay=0.01, max_grad_norm=1.0)
    for step in range(2):
        ddp.zero_grad()
        _grads_of(m, _batch(dev, 100 + 10 * step + 0))
        ddp.finish()
        g0 = [b.buffer.clone() for b in ddp.buckets]
        ddp.zero_grad()
        _grads_of(m, _batch(dev, 100 + 10 * step + 1))
        ddp.finish()
        for b, a in zip(ddp.buckets, g0):                    # sum in the bucket dtype, then halve: what the collective path does
            b.buffer.add_(a).mul_(0.5)
        opt.step()
    ddp.remove()
    worst = 0.0
    for n, p in m.named_parameters():
        if p.requires_grad:
            ref = p.detach().float().cpu()
            d = float((torch.from_numpy(p0[n]) - ref).norm() / ref.norm().clamp_min(1e-12))
            worst = max(worst, d)
    assert worst < 2e-3, worst           # (bf16 parameters: an ulp here and there from the order of the two-rank sum)


def _rccl_worker(port, q):
    """one rank on RCCL (torch 'nccl' backend on ROCm): the bucketed all-reduce is a real collective launch on RCCL's stream, overlapped
    with backward, and must change nothing on one rank"""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    from mmmm_amd.ddp import BucketedGradAllReduce
    from mmmm_amd.optim import FlatAdamW
    res = {}
    for mode in ('plain', 'rccl'):
        m = _model(dev)
        ddp = BucketedGradAllReduce([p for p in m.parameters() if p.requires_grad], bucket_bytes=1 << 20, world_size=1,
                                    force_collectives=(mode == 'rccl'))
        assert ddp.collectives == (mode == 'rccl') and len(ddp.buckets) > 2
        opt = FlatAdamW(ddp, lr=1e-3, weight_decay=0.01, max_grad_norm=1.0)
        launched = []
        if mode == 'rccl':
            orig = dist.all_reduce
            dist.all_reduce = lambda *a, **k: (launched.append(1), orig(*a, **k))[1]
        try:
            for step in range(3):
                ddp.zero_grad()
                _grads_of(m, _batch(dev, 300 + step))
                ddp.finish()
                opt.step()
        finally:
            if mode == 'rccl':
                dist.all_reduce = orig
        torch.cuda.synchronize()
        res[mode] = ({n: p.detach().float().cpu().numpy() for n, p in m.named_parameters() if p.requires_grad}, len(launched), len(ddp.buckets))
        ddp.remove()
    q.put(res)
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_single_rank_rccl_path_equals_the_collective_free_path_bit_for_bit(dev):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=500)
    p.join(timeout=60)
    assert p.exitcode == 0
    (plain, n0, _), (rccl, n1, nb) = res['plain'], res['rccl']
    assert n0 == 0 and n1 == 3 * nb                      # every bucket went through RCCL in every step
    assert plain.keys() == rccl.keys() and len(plain) > 20
    for n in plain:
        assert (plain[n] == rccl[n]).all(), n
