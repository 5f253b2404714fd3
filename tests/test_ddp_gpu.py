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


def test_structural_production_order_matches_what_backward_does_on_the_real_model(dev):
    """`ddp.production_stages` claims an order (heads, lm_head, norm, decoder L-1..0, embed_tokens, GLU adapter, ViT L-1..0, patch embedding);
    the REAL tiny model (LM + ViT + LoRA + SAM / iSAM unfrozen, full training_step) records the order in which autograd finished the
    slots. No bucket may wait for a LATER stage's gradients (then its all-reduce would not overlap with that stage's backward), and
    only the tail buckets may be issued after the last gradient."""
    from mmmm_amd.data.synthetic import SpecialTokens, make_batch
    from mmmm_amd.ddp import BucketedGradAllReduce, grad_production_order, production_stages
    from mmmm_amd.models.loss import DiceFocalLoss
    from mmmm_amd.models.lora import LoraConfig
    from mmmm_amd.models.mmmm import MMMMForCausalLM, MyPrecision, VisionArgs
    from mmmm_amd.models.segvol.modeling.sam import InstanceSamLoss
    from mmmm_amd.utils import apply_lora
    from tests._gpu_common import randomize_
    from tests.test_model_gpu import tiny_config
    from tests.test_sam_gpu import _tiny_sams
    sam, isam = _tiny_sams('cpu')
    tok = SpecialTokens(base_vocab=184)
    m = MMMMForCausalLM.build(None, vision_override=VisionArgs(pos_embed_shape=(2, 2, 4), patch_size=(4, 8, 8)), tokenizer=tok,
                              sam=sam, isam=isam, mask_loss=DiceFocalLoss(dice_weight=2, focal_weight=2, focal_gamma=2),
                              isam_loss=InstanceSamLoss(use_neg_mask=False, box_l1_weight=5, box_giou_weight=2, disc_weight=2,
                                                        disc_focal_gamma=2, disc_focal_alpha=0.85), config=tiny_config(),
                              freeze_sam=False, freeze_isam=False)
    apply_lora(m, LoraConfig(r=64, lora_alpha=8, lora_dropout=0.0, use_rslora=True))
    randomize_(m, 80)
    m.to(dev)
    MyPrecision().convert_module(m)
    m.train()
    m.sam.eval(); m.isam_model.eval()
    m.on_fit_start()
    names = {id(p): n for n, p in m.named_parameters()}
    stages = production_stages(m)
    stage = {id(p): i for i, st in enumerate(stages) for p in st}
    seq = grad_production_order(m)
    assert len(seq) == sum(p.requires_grad for p in m.parameters()) and len(stages) >= 8
    ddp = BucketedGradAllReduce(seq, world_size=1, order='given', bucket_bytes=96 << 10, tail_bytes=32 << 10)
    try:
        assert len(ddp.buckets) >= 10
        batch = make_batch([(3, 1, 16, 32), (3, 8, 16, 32), (3, 4, 32, 16)], [(1, 8, 8), (4, 8, 8), (2, 8, 8)], [(1, 2, 2), (2, 2, 2), (1, 1, 1)],
                           [30, 28, 33], tok=tok, seed=8, instance=[False, True, False], device=dev, n_pairs=3)
        ddp.zero_grad()
        m.training_step(batch).backward()
        ddp.finish()
        torch.cuda.synchronize()
        pos = {pid: i for i, pid in enumerate(ddp.ready_log)}
        assert len(pos) == len(seq), 'a trainable parameter never reported ready'
        # stage order as backward really produced it (first readiness of each stage must be non-decreasing in the claimed order)
        first = [min(pos[id(p)] for p in st) for st in stages if st]
        assert first == sorted(first), [(i, f) for i, f in enumerate(first)]
        late = []
        for (bi, at, by_finish), nxt in zip(ddp.launch_log, ddp.launch_log[1:]):
            last_group = max(stage[id(p)] for p in ddp.buckets[bi].params)
            later = [pos[id(p)] for p in ddp.buckets[nxt[0]].params if stage[id(p)] > last_group]
            if later and (by_finish or at > min(later) + 1):
                late.append((bi, names[id(ddp.buckets[bi].params[-1])]))
        assert not late, late
        rep = ddp.exposed_report()
        # issued after the last gradient: at most the tail bucket of each (dtype, decay) stream that ends with the step — a tail holds
        # up to tail_bytes, or ONE parameter larger than that (the patch-embedding kernel, the position table)
        biggest = max(p.numel() * p.element_size() for p in seq)
        assert rep['exposed_bytes'] <= 4 * (32 << 10) + 2 * biggest, (rep['exposed_bytes'], biggest)
        assert rep['exposed_bytes'] <= 0.2 * rep['total_bytes'], (rep['exposed_bytes'], rep['total_bytes'])
        print(f"[bucket order] {len(ddp.buckets)} buckets, {rep['exposed_bytes']} of {rep['total_bytes']} bytes issued after the last gradient, "
              f"{rep['inversions']} buckets waited behind an earlier-indexed one")
    finally:
        ddp.remove()
        for p in m.parameters():
            p.grad = None
