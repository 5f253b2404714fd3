"""end-to-end error of the fp32 islands (tiny SAM / iSAM of tests/test_sam_gpu.py and SAM-B at true width on a 3-D grid) against the
oracle for each arithmetic mode of vm_gemm_f32: 0 exact f32 MFMA, 2 split-bf16 3 products, 3 split-bf16 6 products"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from mmmm_amd import kernels as K
from oracle import vividmed as O
from tests._gpu_common import oracle_state, randomize_, rel
from tests.test_sam_gpu import _tiny_sams, _sam_cfg
dev = torch.device('cuda:0')


def tiny(mode):
    K.gemm_f32_mode(mode)
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
    return max(rel(a, b) for a, b in zip(masks, ref)), max(rel(a.grad, b.grad) for a, b in zip(pg, pc))


def full(mode, ref_cache={}):
    from mmmm_amd.models import build_sam
    K.gemm_f32_mode(mode)
    torch.manual_seed(0)
    sam = build_sam(patch_size=16, pos_embed_shape=(8, 32, 32))
    randomize_(sam, 60)
    sam.to(dev)
    g = torch.Generator().manual_seed(6)
    images = [torch.rand(3, 32, 256, 256, generator=g)]
    patch = [(4, 16, 16)]
    prompts = [torch.randn(2, 768, generator=g)]
    pg = [p.to(dev).requires_grad_() for p in prompts]
    masks = sam([x.to(dev) for x in images], patch, pg)
    sum(m.square().mean() for m in masks).backward()
    if 'm' not in ref_cache:
        pc = [p.clone().requires_grad_() for p in prompts]
        ref = O.sam_forward({f'sam.{k}': v for k, v in oracle_state(sam).items()}, O.SamCfg(), 'sam', images, patch, pc)
        sum(m.square().mean() for m in ref).backward()
        ref_cache['m'], ref_cache['g'] = ref[0].detach(), pc[0].grad
    return rel(masks[0], ref_cache['m']), rel(pg[0].grad, ref_cache['g'])


for mode in (0, 3, 2):
    print('mode', mode, 'tiny SAM masks / prompt grads', tiny(mode), 'SAM-B 3-D masks / prompt grads', full(mode), flush=True)
K.gemm_f32_mode(3)
