#!/usr/bin/env python3
"""Benchmark of the VividMed training step on MI355X (BASELINE.json metric: train images/sec/node).

    python bench.py --gpus N --steps K --warmup W            (N=1: plain python; N>1: torch.distributed.run)

A "step" = one MMMMForCausalLM.training_step (forward + backward through the HIP kernels, gradient checkpointing as
the reference's on_fit_start enables it) + bucketed RCCL gradient all-reduce + grad-norm clip (1.0) + AdamW
(lr 5e-5, wd 0.01 — conf/phase-vg/fit.yaml) on one synthetic batch that is resident in HBM before the timed region.
Default workload = BASELINE.json configs[1]: phase-vg LoRA bf16, 2D 448x448, batch 8 per GPU, CogVLM-7B + EVA-ViT-E
+ SAM-B + iSAM at full depth, random-init weights (no checkpoints / network in the image), synthetic data.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel = bf16 MFMA GEMM,
timed live with HIP events around every launch on its own stream) and `cpu_baseline` (the CPU oracle on the host
cores, bounded sample, rank 0 at N=1 only).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
from pathlib import Path

import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA peak (guides/MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (image shape, patch, pool, text tokens, grounding heads)
    'phase-vg-448': dict(image=(3, 1, 448, 448), patch=(1, 16, 16), pool=(1, 2, 2), text=256, sam=True,
                         desc='BASELINE configs[1]: phase-vg LoRA bf16, 2D 448x448, batch 8/GPU, CogVLM-7B + SAM-B + iSAM'),
    'phase-vlm-448': dict(image=(3, 1, 448, 448), patch=(1, 16, 16), pool=(1, 2, 2), text=256, sam=False,
                          desc='phase-vlm (sam=None) LoRA bf16, 2D 448x448, batch 8/GPU, CogVLM-7B'),
    'phase-grg-3d': dict(image=(3, 32, 256, 256), patch=(4, 16, 16), pool=(2, 2, 2), text=256, sam=True,
                         desc='BASELINE configs[3]: phase-grg 3D CT 32x256x256'),
    # BASELINE configs[4] shapes (conf/model-hr.yaml; bf16 here — the reference has no fp8 path): parity / capacity cases
    'model-hr-2d': dict(image=(3, 1, 896, 896), patch=(1, 16, 16), pool=(1, 2, 2), text=256, sam=True,
                        desc='BASELINE configs[4]: high-res 2D 896x896 (3137 ViT tokens, 784 LM image tokens per sample)'),
    'model-hr-3d': dict(image=(3, 64, 384, 384), patch=(8, 16, 16), pool=(2, 2, 2), text=256, sam=True,
                        desc='BASELINE configs[4]: high-res 3D 64x384x384 (4609 ViT tokens, 576 LM image tokens per sample)'),
    # BASELINE configs[2]: per-rank batch = half 2D 448x448, half 3D 32x256x256, variable text length, sam=None
    'phase-vlm-mixed': dict(image=(3, 1, 448, 448), patch=(1, 16, 16), pool=(1, 2, 2), text=256, sam=False, mixed=True,
                            desc='BASELINE configs[2]: phase-vlm, mixed 2D 448x448 / 3D 32x256x256 batch, text 128..512, CogVLM-7B'),
}


def train_flops_per_sample(w: dict, cfg, sam: bool, head_rows: str = 'all') -> float:
    """algorithmic training FLOPs per sample, recompute excluded (SURVEY.md §8d formulas). `head_rows`: 'all' = lm_head over all L rows
    (the formula of SURVEY §8d: the reference computes every row's logits, mmmm.py:333-341 reads the labelled ones), a number = over
    that many rows per sample (what this build executes: `lm_head` + CE on the labelled rows only, modeling_cogvlm.py)"""
    if w.get('mixed'):
        a = {k: v for k, v in w.items() if k != 'mixed'}
        return 0.5 * (train_flops_per_sample(a, cfg, sam, head_rows) + train_flops_per_sample(WORKLOADS['phase-grg-3d'], cfg, sam, head_rows))
    vc = cfg.vision_config
    d, f, h, i = vc['hidden_size'], vc['intermediate_size'], cfg.hidden_size, cfg.intermediate_size
    img, patch, pool = w['image'], w['patch'], w['pool']
    grid = [img[1 + k] // patch[k] for k in range(3)]
    n_patch = math.prod(grid)
    Nv = n_patch + 1
    Np = math.prod(g // p for g, p in zip(grid, pool))
    L = 1 + (Np + 2) + 1 + w['text']
    nl_v, nl_l = vc['num_hidden_layers'], cfg.num_hidden_layers
    F_vit_lin = nl_v * 2 * Nv * (4 * d * d + 2 * d * f) + 2 * n_patch * (768 * patch[0]) * d
    F_vit_attn = nl_v * 4 * Nv * Nv * d
    F_glu = 2 * Np * (d * h + 3 * h * i)
    F_lm_lin = nl_l * 2 * L * (4 * h * h + 3 * h * i)
    F_lm_attn = nl_l * 2 * L * L * h
    F_head = 2 * (L if head_rows == 'all' else float(head_rows)) * h * cfg.vocab_size
    F_lora = 2 * L * 64 * ((h + 3 * h) + (h + h) + 3 * (h + i)) * nl_l + 2 * Nv * 64 * ((d + 3 * d) + (d + d) + 2 * (d + f)) * nl_v
    # frozen linears: fwd + dgrad (x2); trainable linears / LoRA / lm_head / attention: x3; frozen SAM + iSAM: x2 each
    total = 2 * (F_vit_lin + F_glu + F_lm_lin) + 3 * (F_head + F_lora + F_vit_attn + F_lm_attn)
    if sam:
        Ns = n_patch
        F_sam = 12 * (2 * Ns * (4 * 768 ** 2 + 2 * 768 * 3072) + 4 * Ns * Ns * 768)
        total += 2 * 2 * F_sam
    return float(total)


def build(workload: dict, device, depth_scale: float = 1.0):
    from mmmm_amd.models.cogvlm.configuration_cogvlm import CogVLMConfig
    from mmmm_amd.models.lora import LoraConfig
    from mmmm_amd.models.mmmm import MMMMForCausalLM, MyPrecision, VisionArgs
    from mmmm_amd.data.synthetic import SpecialTokens
    from mmmm_amd.utils import apply_lora
    cfg = CogVLMConfig()
    if depth_scale != 1.0:
        cfg.num_hidden_layers = max(1, int(cfg.num_hidden_layers * depth_scale))
        cfg.vision_config['num_hidden_layers'] = max(1, int(cfg.vision_config['num_hidden_layers'] * depth_scale))
    tok = SpecialTokens(base_vocab=32000)
    torch.set_default_dtype(torch.bfloat16)
    with torch.device(device):
        sam = isam = mask_loss = isam_loss = None
        if workload['sam']:
            torch.set_default_dtype(torch.float32)
            from mmmm_amd.models.segvol import build_sam, build_instance_sam
            from mmmm_amd.models.loss import DiceFocalLoss
            from mmmm_amd.models.segvol.modeling.sam import InstanceSamLoss
            sam = build_sam(patch_size=16, pos_embed_shape=(8, 32, 32))
            isam = build_instance_sam(patch_size=16, num_instances=6, pos_embed_shape=(8, 32, 32))
            mask_loss = DiceFocalLoss(dice_weight=2, focal_weight=2, focal_gamma=2)
            isam_loss = InstanceSamLoss(use_neg_mask=False, box_l1_weight=5, box_giou_weight=2, disc_weight=2,
                                        disc_focal_gamma=2, disc_focal_alpha=0.85)
            torch.set_default_dtype(torch.bfloat16)
        model = MMMMForCausalLM.build(None, vision_override=VisionArgs(pos_embed_shape=(8, 32, 32), pt_pos_embed_shape=(35, 35), patch_size=16),
                                      tokenizer=tok, sam=sam, mask_loss=mask_loss, isam=isam, isam_loss=isam_loss, config=cfg,
                                      # Stage 1 / Stage 3 command lines of the reference README (:36, :40): --model.freeze_sam false --model.freeze_isam false
                                      freeze_sam=False, freeze_isam=False)
    torch.set_default_dtype(torch.float32)
    if workload['sam']:
        model.vg_proj.float()
    apply_lora(model, LoraConfig(r=64, lora_alpha=8, lora_dropout=0.05, use_rslora=True))
    with torch.no_grad():   # non-degenerate adapters / special embeddings (peft zero-inits B; a zero B hides the LoRA path)
        g = torch.Generator(device=device).manual_seed(1234)
        for n, p in model.named_parameters():
            if 'lora_B' in n:
                p.copy_(torch.randn(p.shape, device=device, generator=g, dtype=torch.float32) * 0.01)
            if n.endswith(('boi', 'eoi', 'cls_embedding.weight', 'cls_pos_embed.weight', 'position_embedding.weight', 'position_embeddings.weight')):
                p.copy_(torch.randn(p.shape, device=device, generator=g, dtype=torch.float32) * 0.02)
    MyPrecision().convert_module(model)
    model.train()
    if sam is not None:
        sam.eval(); isam.eval()
    model.on_fit_start()
    return model, tok


def make_batch(workload: dict, tok, B: int, device, seed: int):
    from mmmm_amd.data.synthetic import make_batch as mk
    inst = [(i % 2 == 1) for i in range(B)] if workload['sam'] else [False] * B
    if workload.get('mixed'):
        w3 = WORKLOADS['phase-grg-3d']
        g = torch.Generator().manual_seed(1000 + seed)
        texts = torch.randint(128, 513, (B,), generator=g).tolist()
        sel = [w3 if i % 2 else workload for i in range(B)]
        return mk([w['image'] for w in sel], [w['patch'] for w in sel], [w['pool'] for w in sel], texts, tok=tok, seed=seed,
                  grounding=False, n_pairs=4, instance=inst, device=device)
    return mk([workload['image']] * B, [workload['patch']] * B, [workload['pool']] * B, [workload['text']] * B, tok=tok,
              seed=seed, grounding=workload['sam'], n_pairs=4, instance=inst, device=device)


def _oracle_state_and_batch(workload: dict, cfg, n_dec: int, n_vit: int, tok):
    """state dict of the REAL module tree at the true widths with `n_dec` decoder and `n_vit` ViT-E layers (built on the meta device,
    materialised from one tiled 1 M-element normal sample: timing only), LoRA r64 applied, SAM-B + iSAM when the workload has them —
    and ONE image of the workload as the reference's Batch on the CPU."""
    import copy
    from mmmm_amd.data.synthetic import make_batch as mk
    from mmmm_amd.models.lora import LoraConfig
    from mmmm_amd.models.mmmm import MMMMForCausalLM, VisionArgs
    from mmmm_amd.utils import apply_lora
    c = copy.deepcopy(cfg)
    c.num_hidden_layers = n_dec
    c.vision_config = dict(c.vision_config, num_hidden_layers=n_vit)
    try:
        with torch.device('meta'):
            sam = isam = None
            if workload['sam']:
                from mmmm_amd.models.segvol import build_sam, build_instance_sam
                sam = build_sam(patch_size=16, pos_embed_shape=(8, 32, 32))
                isam = build_instance_sam(patch_size=16, num_instances=6, pos_embed_shape=(8, 32, 32))
            m = MMMMForCausalLM.build(None, vision_override=VisionArgs(pos_embed_shape=(8, 32, 32), pt_pos_embed_shape=(35, 35), patch_size=16),
                                      tokenizer=tok, sam=sam, mask_loss=None, isam=isam, isam_loss=None, config=c, freeze_sam=False, freeze_isam=False)
            apply_lora(m, LoraConfig(r=64, lora_alpha=8, lora_dropout=0.0, use_rslora=True))
    finally:
        torch.set_default_dtype(torch.float32)
    g = torch.Generator().manual_seed(0)
    base = torch.randn(1 << 20, generator=g) * 0.02
    trainable = {n for n, p in m.named_parameters() if p.requires_grad}
    sd = {}
    for k, v in m.state_dict().items():
        if not v.is_floating_point():
            sd[k] = torch.zeros(v.shape, dtype=v.dtype)
            continue
        n = v.numel()
        if k.endswith(('norm.weight', 'layernorm.weight', 'norm1.weight', 'norm2.weight')) or ('norm' in k.rsplit('.', 2)[-2] and k.endswith('.weight')):
            t = torch.ones(v.shape)
        else:
            t = base.repeat(-(-n // base.numel()))[:n].view(v.shape).clone()
        sd[k] = t.requires_grad_(k in trainable)
    batch = mk([workload['image']], [workload['patch']], [workload['pool']], [workload['text']], tok=tok, seed=5, grounding=workload['sam'],
               n_pairs=4, instance=[False], device=torch.device('cpu'))
    batch['image'] = [x.float() for x in batch['image']]
    return sd, batch, c


def cpu_baseline(workload: dict, cfg, tok) -> dict:
    """The CPU oracle (oracle/vividmed.py, pinned against the reference) timed on the host cores on a bounded sample of the SAME
    workload: its whole `training_step` (mmmm.py:296-352: ViT-E + GLU adapter + decoder + lm_head / weighted CE + SAM-B mask head and
    losses, forward AND backward, LoRA r64 on every linear) for ONE image at the TRUE layer widths with 1 + 1 and with 2 + 2 layers.
    Only the depth is extrapolated: T(32 + 63) = T(1 + 1) + (31 r + 62 (1 - r)) (T(2 + 2) - T(1 + 1)), r = the decoder layer's share of
    the algorithmic FLOPs of one decoder + one ViT-E layer. Reported baseline, not a target."""
    from oracle import vividmed as O
    ncpu = os.cpu_count() or 1
    vc = cfg.vision_config

    def step_cfg(c):
        ocfg = O.Cfg(vocab_size=c.vocab_size, hidden_size=c.hidden_size, intermediate_size=c.intermediate_size,
                     num_hidden_layers=c.num_hidden_layers, num_attention_heads=c.num_attention_heads, rms_norm_eps=c.rms_norm_eps,
                     vision=O.VisionCfg(hidden_size=vc['hidden_size'], num_heads=vc['num_heads'], num_hidden_layers=c.vision_config['num_hidden_layers'],
                                        intermediate_size=vc['intermediate_size'], layer_norm_eps=vc['layer_norm_eps'],
                                        patch_size=tuple(vc['patch_size']), pos_embed_shape=tuple(vc['pos_embed_shape']), in_channels=vc['in_channels']))
        if not workload['sam']:
            return O.StepCfg(lm=ocfg, sam=None, isam=None, mask_loss=None, isam_loss=None, bop_token_id=tok.bop_token_id, eop_token_id=tok.eop_token_id)
        return O.StepCfg(lm=ocfg, sam=O.SamCfg(), isam=O.SamCfg(num_instances=6, instance=True),
                         mask_loss=dict(dice_weight=2, focal_weight=2, focal_gamma=2), isam_loss=O.ISamLossCfg(),
                         bop_token_id=tok.bop_token_id, eop_token_id=tok.eop_token_id)

    def run(n_dec, n_vit):
        sd, batch, c = _oracle_state_and_batch(workload, cfg, n_dec, n_vit, tok)
        t0 = time.perf_counter()
        loss, _ = O.training_step(sd, step_cfg(c), batch, rope_dtype=torch.float32)
        loss.backward()
        return time.perf_counter() - t0

    # Core-count policy, fixed (round 5): min(32, host threads). All 256 threads of the GPU box are slower than 32 on layer-sized GEMMs (round 1:
    # 27 s per layer with 256 threads against 4.7 s on 8 cores), and a per-run probe of {8, 16, 32, 64, all} made `cores` — and with it the
    # value — wander between runs (0.030 / 0.053 / 0.022 images/s in rounds 3 / 4 / 5 on the same method). One number, always measured the same way.
    best = min(32, ncpu)
    torch.set_num_threads(best)
    t11 = run(1, 1)
    t22 = run(2, 2)
    # one decoder layer's share of (decoder layer + ViT-E layer), from the workload's own FLOP formulas
    one = dict(workload, mixed=False)
    import copy
    c1 = copy.deepcopy(cfg); c1.num_hidden_layers = 1; c1.vision_config = dict(vc, num_hidden_layers=0)
    c2 = copy.deepcopy(cfg); c2.num_hidden_layers = 0; c2.vision_config = dict(vc, num_hidden_layers=1)
    c0 = copy.deepcopy(cfg); c0.num_hidden_layers = 0; c0.vision_config = dict(vc, num_hidden_layers=0)
    f0 = train_flops_per_sample(one, c0, False)
    fd, fv = train_flops_per_sample(one, c1, False) - f0, train_flops_per_sample(one, c2, False) - f0
    r = fd / (fd + fv)
    delta = max(t22 - t11, 0.0)
    nl, nv = cfg.num_hidden_layers, vc['num_hidden_layers']
    per_image = t11 + ((nl - 1) * r + (nv - 1) * (1 - r)) * delta
    return {'value': 1.0 / per_image, 'unit': 'images/s', 'cores': best, 'kind': 'port',
            'sample': (f'oracle fp32 training_step (forward + backward, LoRA r64, {"SAM-B mask head + losses" if workload["sam"] else "no grounding heads"}) of ONE image '
                       f'of {workload["desc"].split(":")[0]} at the true widths: 1 + 1 layers {t11:.1f} s, 2 + 2 layers {t22:.1f} s, {best} of {ncpu} host threads '
                       f'(fixed policy: min(32, host threads)); depth extrapolated to {nl} + {nv} layers with the decoder layer taking {r:.2f} of the increment (FLOP share)')}


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py
    <same arguments>` as a child process (one rank per GPU, rendezvous on 127.0.0.1) and return its exit code. Rank 0 of the child
    job prints the JSON line on the inherited stdout."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def gemm_source_digest() -> str:
    """sha256 (first 16 hex digits) over the sources of the dominant GEMM kernel: ties a committed PMC measurement to the code it measured"""
    import hashlib
    h = hashlib.sha256()
    for f in ('gemm256.hip', 'gemm256w.hpp', 'gemm.hip', 'gemm_common.hpp', 'vm_tile.hpp'):
        h.update((ROOT / 'mmmm_amd' / 'csrc' / f).read_bytes())
    return h.hexdigest()[:16]


ALSO_BATCH = {'model-hr-2d': 4, 'model-hr-3d': 4}      # per-GPU batch of the high-resolution workloads (activations of 3 137 / 4 609 ViT tokens per image)


def run_also(workloads: list, args) -> list:
    """other workloads of BASELINE.json (phase-vlm: the one the north_star's 50 % target is quoted on; configs[3] phase-grg-3d and
    configs[4] model-hr-2d in bf16 and with the e4m3 frozen-weight GEMMs, spelled `model-hr-2d:fp8`) measured by child processes of this
    script, one after the other, BEFORE this process touches the GPU (a child needs the HBM to itself). A child is only started while
    `--also-budget` seconds of wall clock last: the default run has to stay within minutes. Returns their condensed lines."""
    import subprocess
    out = []
    t_all = time.perf_counter()
    for spec in workloads:
        wl, _, mode = spec.partition(':')
        if wl not in WORKLOADS or mode not in ('', 'fp8', 'exact-islands'):
            raise SystemExit(f'--also: unknown workload {spec}')
        if time.perf_counter() - t_all > args.also_budget:
            out.append({'workload': spec, 'skipped': f'--also-budget {args.also_budget:.0f} s used up by the workloads before it'})
            continue
        batch = min(args.batch, ALSO_BATCH.get(wl, args.batch))
        cmd = [sys.executable, str(Path(__file__).resolve()), '--workload', wl, '--steps', '12', '--warmup', '3', '--no-cpu-baseline',
               '--no-kernel-events', '--no-peak-probe', '--also', '', '--batch', str(batch), '--checkpointing', args.checkpointing,
               '--hbm-fraction', str(args.hbm_fraction)] + (['--fp8'] if mode == 'fp8' else [])
        if mode == 'exact-islands':
            # the fp32 islands entirely on six-product (fp32-exact) arithmetic: the step time the three-product default of the two SAM-B
            # image encoders (config.fp32_islands) is traded against
            cmd += ['--set', 'models.segvol.modeling.image_encoder.ENCODER_F32_SPLIT=3']
        t0 = time.perf_counter()
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        line = next((ln for ln in r.stdout.splitlines() if ln.startswith('{')), None)
        if r.returncode != 0 or line is None:
            out.append({'workload': spec, 'error': f'exit code {r.returncode}'})
            continue
        j = json.loads(line)
        out.append({'workload': spec, 'description': j['config']['description'], 'dtype': j['dtype'], 'per_gpu_batch': batch, 'value': j['value'],
                    'unit': j['unit'], 'ms_per_step': j['ms_per_step'], 'steps': j['steps'], 'warmup': j['warmup'],
                    'mfma_utilisation_step': j['mfma_utilisation_step'], 'model_tflops_per_image': j['model_tflops_per_image'],
                    'host_enqueue_ms': j.get('host_enqueue_ms'),
                    'gradient_checkpointing': j['config']['gradient_checkpointing'], 'wgrad_side_stream': j['config']['wgrad_side_stream'],
                    'wall_s': round(time.perf_counter() - t0, 1)})
    return out


_JSON_OUT = None


def reserve_stdout():
    """stdout carries ONE JSON line and nothing else. Libraries that print from C write to fd 1 as well (RCCL's version banner at the first
    collective: five lines on rank 0's stdout) — so from here on fd 1 is an alias of stderr and the JSON line goes to a private duplicate
    of the original stdout."""
    global _JSON_OUT
    if _JSON_OUT is None:
        sys.stdout.flush()
        _JSON_OUT = os.fdopen(os.dup(1), 'w')
        os.dup2(2, 1)
    return _JSON_OUT


def dry_run_cpu(args, rank: int, world: int):
    """the N-rank plumbing of this file (rendezvous, bucketed all-reduce overlapped with backward, barrier + max-over-ranks
    timing, one JSON line from rank 0) on CPU tensors over gloo. The VividMed model itself has no CPU path: a toy network
    stands in, and the line says so."""
    from mmmm_amd.ddp import BucketedGradAllReduce
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29512')
        dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.ReLU(), torch.nn.Linear(256, 256), torch.nn.ReLU(), torch.nn.Linear(256, 8))
    ddp = BucketedGradAllReduce(net.parameters(), world_size=world, bucket_bytes=64 << 10)
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3)
    batches = [torch.randn(args.batch, 64, generator=torch.Generator().manual_seed(1000 * rank + i)) for i in range(max(1, args.batches))]

    def step(i):
        ddp.zero_grad()
        loss = net(batches[i % len(batches)]).square().mean()
        loss.backward()
        ddp.finish()
        ddp.clip_grad_norm_(1.0)
        opt.step()
        return loss

    for i in range(args.warmup):
        step(i)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
        # every rank must hold the same parameters after identical updates of averaged gradients
        flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        ref = flat.clone()
        dist.broadcast(ref, 0)
        assert torch.equal(flat, ref), 'ranks diverged'
    if rank == 0:
        print(json.dumps({'metric': 'train images/sec/node', 'value': world * args.batch * args.steps / dt, 'unit': 'images/s',
                          'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
                          'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
                          'data': 'dry-run (CPU, toy network): launcher / collective plumbing check, NOT a measurement',
                          'dry_run': True, 'config': {'workload': 'dry-run-cpu', 'parallelism': f'dp{world}', 'backend': 'gloo'},
                          'loss': float(loss.detach())}), file=reserve_stdout(), flush=True)
    if world > 1:
        dist.destroy_process_group()


def apply_sets(sets) -> bool:
    """--set MODULE.NAME=VALUE: module-level constants of mmmm_amd for A/B measurements (python literals). Returns whether
    functional.WGRAD_SIDE_STREAM was among them (the calibration then leaves it alone)."""
    import ast
    import importlib
    forced = False
    for item in sets:
        path, _, val = item.partition('=')
        parts = path.split('.')
        m, k = None, len(parts) - 1
        while k > 0 and m is None:                 # the longest importable module prefix, then attributes (a class constant: models.mmmm.MMMMForCausalLM.concurrent_heads)
            name_k = 'mmmm_amd.' + '.'.join(parts[:k])
            try:
                m = importlib.import_module(name_k)
            except ModuleNotFoundError as e:       # only "this prefix is not a module": a failing import INSIDE a module is an error of its own
                if e.name is None or not name_k.startswith(e.name):
                    raise
                k -= 1
        if m is None:
            raise SystemExit(f'--set {item}: no module behind mmmm_amd.{path}')
        for a in parts[k:-1]:
            m = getattr(m, a)
        mod, name = '.'.join(parts[:-1]), parts[-1]
        if not hasattr(m, name):
            raise SystemExit(f'--set {item}: mmmm_amd.{mod} has no attribute {name}')
        setattr(m, name, ast.literal_eval(val))
        forced = forced or (mod == 'functional' and name == 'WGRAD_SIDE_STREAM')
    return forced


def _enc_split() -> int:
    from mmmm_amd.models.segvol.modeling import image_encoder
    return image_encoder.ENCODER_F32_SPLIT


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--hbm-fraction', type=float, default=0.76,
                    help='--checkpointing hbm: reserved-memory target of the calibration steps (the pool settles ~10 %% of the HBM above it)')
    ap.add_argument('--workload', default='phase-vg-448', choices=list(WORKLOADS))
    ap.add_argument('--batch', type=int, default=8, help='samples per GPU')
    ap.add_argument('--depth-scale', type=float, default=1.0, help='debug only: <1 shrinks depth and invalidates the number')
    ap.add_argument('--checkpointing', default='hbm', choices=['hbm', 'reference'],
                    help="'reference': recompute every transformer layer in backward (mmmm.py:232-233); 'hbm': keep the "
                         "activations of as many layers as the free HBM of this device holds (same results, less recompute)")
    ap.add_argument('--optimizer', default='flat', choices=['flat', 'torch'],
                    help="'flat': fused clip + AdamW kernel over the gradient buckets; 'torch': clip on the buckets + torch.optim.AdamW(fused)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--fp8', action='store_true',
                    help="BASELINE configs[4] 'fp8 MFMA path': frozen base-weight linears of the ViT-E / decoder in e4m3 (forward + input gradient); "
                         'everything else as in the bf16 run. Not valid for the bf16 headline workload.')
    ap.add_argument('--batches', type=int, default=4, help='distinct synthetic batches (seeded per rank) rotated through the timed region')
    ap.add_argument('--dry-run-cpu', action='store_true',
                    help='launcher / collective plumbing check without a GPU: gloo backend, a toy CPU network through the same '
                         'BucketedGradAllReduce; the printed line is marked dry_run and is NOT a measurement')
    ap.add_argument('--all-kernel-events', action='store_true',
                    help='also bracket attention / fp32 GEMM / LoRA launches (default: only the dominant bf16 GEMM)')
    ap.add_argument('--event-stride', type=int, default=16,
                    help='bracket a pseudo-random 1-in-n sample of the launches of the dominant kernel with HIP events (1 in 4 cost the step 1.5 ms of its 303: '
                         'an event record keeps consecutive kernels from overlapping their ramps; 1 in 16 costs 0.1 ms and still samples > 2 500 launches over the 50 default steps)')
    ap.add_argument('--no-kernel-events', action='store_true', help='skip the per-launch HIP event bracketing (roofline)')
    ap.add_argument('--no-calibrate', action='store_true',
                    help='counter-collection runs only (tools/pmc_step.sh): keep the planning step but skip the calibration steps that follow it')
    ap.add_argument('--no-peak-probe', action='store_true', help='skip the 2 s MFMA peak measurement (roofline.peak_measured)')
    ap.add_argument('--also-budget', type=float, default=330.0,
                    help='seconds of wall clock for the --also children together: a child is only started while the budget lasts (the rest are reported as skipped)')
    ap.add_argument('--cpu-baseline-budget', type=float, default=100.0,
                    help="seconds of host time for the CPU baselines of the --also workloads (oracle training_step of ONE image at the true widths, 1 + 1 and "
                         "2 + 2 layers, depth extrapolated — the headline's method); a workload is only timed while the budget lasts")
    ap.add_argument('--also', default='phase-vlm-448,phase-vlm-mixed,phase-grg-3d,model-hr-2d,model-hr-2d:fp8,model-hr-3d,phase-vg-448:exact-islands',
                    help="N = 1 only: further workloads measured by child processes BEFORE the headline run (12 timed steps each) and "
                         "reported under 'also' in the same JSON line — the north_star's target is quoted on phase-vlm; '' disables")
    ap.add_argument('--set', action='append', default=[], metavar='MODULE.NAME=VALUE',
                    help="A/B measurements (tools/ab_set.sh): set a module-level constant of mmmm_amd before the model is built, e.g. "
                         "functional.WGRAD_SIDE_STREAM=True, functional.NN_DGRAD=True, models.cogvlm.modeling_cogvlm.LM_HEAD_LABEL_ROWS=False. "
                         "The product reads no environment switches; a number measured with --set is an experiment, not the benchmark")
    ap.add_argument('--force-dist', action='store_true', help='one rank, but through the RCCL collectives (tests)')
    ap.add_argument('--fwd-pool', action='store_true', help='experiment: forward allocations from a memory pool of their own')
    ap.add_argument('--mem-summary', action='store_true', help='print the allocator summary after each calibration round')
    args = ap.parse_args()

    # N > 1 from a plain `python bench.py --gpus N`: start one fresh process per GPU through torch.distributed.run and pass its
    # exit code on. This happens BEFORE anything touches the GPU (no GPU call above this line): a process that has
    # initialised HIP must never exec / be replaced, children are the only safe way.
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    also = []
    if args.also and args.gpus == 1 and 'WORLD_SIZE' not in os.environ and not args.dry_run_cpu and args.depth_scale == 1.0 and not args.fp8:
        also = run_also([w for w in args.also.split(',') if w and w != args.workload], args)       # (`model-hr-2d:fp8` != `model-hr-2d`)
    reserve_stdout()          # (after the self-launch branch: the child ranks inherit the real stdout) before anything that may print from C
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    if args.dry_run_cpu:
        return dry_run_cpu(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the VividMed hot path has no CPU fallback')
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    use_dist = world > 1 or args.force_dist      # --force-dist: exercise the RCCL path on one rank
    if use_dist:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)

    from mmmm_amd import kernels as K, hip
    side_stream_forced = apply_sets(args.set)
    from mmmm_amd.ddp import BucketedGradAllReduce
    w = WORKLOADS[args.workload]
    model, tok = build(w, device, args.depth_scale)
    model.trainer.is_parallel = world > 1          # Lightning's flag (mmmm.py:263: dummy head forwards on ranks that used none)
    n_fp8 = 0
    if args.fp8:
        from mmmm_amd.models.lora import enable_fp8
        n_fp8 = enable_fp8(model)
    from mmmm_amd.ddp import grad_production_order
    trainable = grad_production_order(model)       # heads, lm_head, norm, decoder 31..0, embed_tokens, GLU, ViT-E 62..0, patch embedding
    assert len(trainable) == sum(p.requires_grad for p in model.parameters())
    ddp = BucketedGradAllReduce(trainable, world_size=world, force_collectives=use_dist, order='given',
                                registration=[p for p in model.parameters() if p.requires_grad])
    from mmmm_amd.optim import FlatAdamW
    if args.optimizer == 'flat':       # gradient clip (1.0) + AdamW in one kernel per bucket (mmmm_amd/optim.py)
        opt = FlatAdamW(ddp, lr=5e-5, weight_decay=0.01, max_grad_norm=1.0)
    else:
        opt = torch.optim.AdamW(trainable, lr=5e-5, weight_decay=0.01, fused=True)
    # distinct synthetic batches, seeded per rank, all resident in HBM before timing; the steps rotate through them (the mixed
    # workload draws new text lengths per batch, so sequence-length tables change from step to step as they do in training)
    batches = [make_batch(w, tok, args.batch, device, seed=1000 * rank + i) for i in range(max(1, args.batches))]
    it = [0]
    fwd_pool = torch.cuda.MemPool() if args.fwd_pool else None

    def step():
        batch = batches[it[0] % len(batches)]
        it[0] += 1
        ddp.zero_grad()
        if fwd_pool is not None:
            # forward allocations (mostly the activations kept for backward: long-lived) come from their own pool, the backward
            # pass's temporaries (autograd thread) from the default one: the two lifetimes no longer fragment each other's blocks
            with torch.cuda.use_mem_pool(fwd_pool):
                loss = model.training_step(batch)
        else:
            loss = model.training_step(batch)
        loss.backward()
        ddp.finish()
        if args.optimizer != 'flat':
            ddp.clip_grad_norm_(1.0)                  # gradient_clip_val 1 (conf/phase-vg/fit.yaml), on the flat buckets
        opt.step()
        return loss

    from mmmm_amd.models.lora import ActivationBudget
    plan = 'every layer recomputed'
    import mmmm_amd.functional as Fh
    side_stream_note = ['on' if Fh.WGRAD_SIDE_STREAM else 'off (functional.WGRAD_SIDE_STREAM)']
    if args.checkpointing == 'hbm':
        # planning step (untimed, not a warmup step): peak HBM with every layer checkpointed -> what is left over
        # becomes the activation budget of the following steps
        torch.cuda.reset_peak_memory_stats()
        step()
        torch.cuda.synchronize()
        total_hbm = torch.cuda.get_device_properties(device).total_memory
        budget = int(0.88 * total_hbm) - torch.cuda.max_memory_allocated() - (8 << 30)
        r0 = torch.cuda.max_memory_reserved()
        if use_dist:
            t = torch.tensor([budget], device=device, dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            budget = int(t.item())
        ActivationBudget.limit = max(budget, 0)
        if rank == 0:
            print(f'[plan] peak allocated {torch.cuda.max_memory_allocated() / 2**30:.1f} GB, reserved {r0 / 2**30:.1f} GB with every layer recomputed -> activation budget {budget / 2**30:.1f} GB', file=sys.stderr)
        # calibration (untimed): the per-layer estimate of ActivationBudget.claim covers the two big transformers, not the
        # grounding heads' fp32 volumes, the allocator's fragmentation or the frees deferred by the side streams. If a step
        # under the plan reserves more than --hbm-fraction of the HBM (the pool still grows ~10 % over the following steps before it settles) the caching allocator ends up flushing and re-allocating its pool
        # every step (3D workloads: 850 -> 1570 ms/step, thousands of hipMalloc / hipFree per step), so the budget is cut by
        # the overshoot and the step repeated.
        target = int(args.hbm_fraction * total_hbm)
        for _ in range(0 if args.no_calibrate else 5):
            torch.cuda.empty_cache()
            torch.cuda.reset_peak_memory_stats()
            try:
                for _ in range(3):
                    step()
                torch.cuda.synchronize()
            except torch.OutOfMemoryError:        # the plan did not even fit: shrink hard and try again
                if use_dist:
                    raise                          # (ranks must take the same number of collectives: no local retries)
                ddp.abort_step()                  # buckets, fp32 side accumulators, tensors parked by the aborted backward pass
                torch.cuda.empty_cache()
                ActivationBudget.limit = int(ActivationBudget.limit * 0.6)
                if rank == 0:
                    print(f'[calibrate] out of memory -> budget {ActivationBudget.limit / 2**30:.1f} GB', file=sys.stderr)
                continue
            r1 = torch.cuda.max_memory_reserved()
            if use_dist:
                t = torch.tensor([r1], device=device, dtype=torch.int64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                r1 = int(t.item())
            if rank == 0:
                print(f'[calibrate] budget {ActivationBudget.limit / 2**30:.1f} GB kept {ActivationBudget.last_plan}: reserved {r1 / 2**30:.1f} GB, allocated peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GB (target {target / 2**30:.0f} GB)', file=sys.stderr)
            if rank == 0 and args.mem_summary:
                print(torch.cuda.memory_summary(abbreviated=True), file=sys.stderr)
            if r1 <= target + (4 << 30) or ActivationBudget.limit == 0:
                break
            import mmmm_amd.functional as Fh
            if Fh.WGRAD_SIDE_STREAM and not side_stream_forced:
                # First resort: give up the weight-gradient side stream instead of kept layers. Every tensor its kernels read is
                # withheld from the allocator until that stream has passed it, and the stream owns a block pool of its own: on the
                # large workloads the pool then holds 50-90 GB more than is ever allocated (phase-grg-3d: reserved 244 GB for an
                # allocated peak of 188 GB with the side stream, 199 GB for 196 GB without). The stream is worth ~4 % of a step,
                # a recomputed third of the ViT ~15 %.
                Fh.WGRAD_SIDE_STREAM = False
                side_stream_note[0] = 'off (memory: reserved %.0f GB > target %.0f GB with it)' % (r1 / 2**30, target / 2**30)
                if rank == 0:
                    print('[calibrate] weight-gradient side stream off', file=sys.stderr)
                continue
            # reserved memory is ~linear in the kept bytes: interpolate between "nothing kept" (r0) and this plan (r1)
            scale = max(0.0, (target - r0) / max(r1 - r0, 1))
            ActivationBudget.limit = int(ActivationBudget.limit * min(scale, 0.95))
    for _ in range(args.warmup):
        loss = step()
    if args.checkpointing == 'hbm':
        plan = 'hbm budget %.0f GB: kept/total layers %s' % (ActivationBudget.limit / 2**30, ActivationBudget.last_plan)
    peak_measured = None
    if rank == 0 and world == 1 and not args.no_peak_probe:
        # the box's own sustained bf16 MFMA rate (2 s of bare MFMAs on random operands, every CU busy): nominal 2.5 PFLOP/s assumes
        # 2.4 GHz, the chip holds 1.9-1.95 GHz under matrix load (guide: DVFS give-back). Outside the timed region, then two untimed
        # steps so that the caches / clocks of the step are back before timing starts.
        peak_measured = K.ubench_mfma_bf16(2.0)
        for _ in range(2):
            step()
    use_events = not args.no_kernel_events
    if use_events:
        K.prof_reset()
        K.prof_stride(args.event_stride)
        # the dominant GEMM (1-in-stride sample) and the ~350 attention launches of a step; --all-kernel-events adds fp32 GEMM / LoRA
        K.prof_enable(True if args.all_kernel_events else (hip.PROF_GEMM_BF16, hip.PROF_ATTN))
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    model.freeze_python_gc()          # model, buckets, optimizer state and the batch are permanent: keep the cyclic GC off them
    ms0 = torch.cuda.memory_stats(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    ms1 = torch.cuda.memory_stats(device)
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if use_events:
        K.prof_enable(False)
    if use_dist:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    loss_v = float(loss.item())
    # host enqueue time of a step: inside the timed region the host runs ahead of the GPU until the launch queue throttles it, so its
    # per-step wall time there equals the GPU's. Five more steps (untimed), each started from an idle GPU: time until step() returns.
    host_enq = []
    for _ in range(5):
        torch.cuda.synchronize()
        h0 = time.perf_counter()
        step()
        host_enq.append((time.perf_counter() - h0) * 1e3)
    torch.cuda.synchronize()
    if world > 1:
        ddp.assert_replicas_equal(what='trainable parameters after the timed region')       # identical updates of averaged gradients

    if rank == 0:
        images = world * args.batch * args.steps
        value = images / dt
        fl_formula = train_flops_per_sample(w, model.config, w['sam'])
        # lm_head runs on the labelled rows only (the reference reads its logits there, mmmm.py:333-341): count what is executed
        from mmmm_amd.models.cogvlm import modeling_cogvlm as _mc
        n_lab = sum(float((b['vlm_inputs']['labels'] != -100).sum()) for b in batches) / (len(batches) * args.batch)
        fl_sample = train_flops_per_sample(w, model.config, w['sam'], head_rows=n_lab) if _mc.LM_HEAD_LABEL_ROWS else fl_formula
        out = {
            'metric': 'train images/sec/node', 'value': value, 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'fp8-e4m3 frozen-weight GEMMs (fwd + dgrad), bf16 elsewhere' if args.fp8 else 'bf16', 'data': 'synthetic',
            'config': {'workload': args.workload, 'description': w['desc'], 'per_gpu_batch': args.batch, 'global_batch': args.batch * world,
                       'text_tokens': w['text'], 'distinct_batches': len(batches), 'parallelism': f'dp{world}', 'weights': 'random-init', 'lora': 'r64 rsLoRA dropout 0.05', 'sam': 'SAM-B + iSAM fp32, unfrozen (README Stage 1: --model.freeze_sam false --model.freeze_isam false)' if w['sam'] else None,
                       'gradient_checkpointing': plan, 'wgrad_side_stream': side_stream_note[0], 'fp8_linears': n_fp8, 'fp32_islands': ('sam / isam_model / vg_proj fp32 on split-bf16 MFMA products: six per product (fp32-exact) everywhere except the 12 blocks of the two SAM-B image encoders, which use three (image_encoder.ENCODER_F32_SPLIT = %d)' % _enc_split()) if w['sam'] else None, 'optimizer': 'clip 1.0 + AdamW, ' + ('one fused kernel per gradient bucket' if args.optimizer == 'flat' else 'torch.optim fused'), 'depth_scale': args.depth_scale,
                       'resample': 'parity UNPINNED for one op on this path: luolib.models.spadop.resample (the position tables [C, 8, 32, 32] -> the image\'s patch grid, every step; visual.py:44-66, image_encoder.py:74-115) lives in an un-vendored submodule absent from /root/reference — mmmm_amd/models/resample.py and the oracle restate it as linear interpolation (align_corners False), and no test can see that guess being wrong'},
            'loss': loss_v,
            'host_enqueue_ms': min(host_enq), 'host_enqueue_median_ms': sorted(host_enq)[len(host_enq) // 2],
            'host_enqueue_note': 'time until step() has enqueued every launch, from an idle GPU (min / median of 5 untimed steps after the timed region); must stay below ms_per_step',
            # hipMalloc / hipFree calls of the caching allocator inside the timed region (measured harmless: a run with 1 and
            # runs with 43-65 calls in 12 steps take the same time; reserving a large segment up front changes nothing)
            'allocator': {k: int(ms1.get(k, 0) - ms0.get(k, 0)) for k in ('num_device_alloc', 'num_device_free', 'num_alloc_retries')}
                         | {'reserved_gb': round(ms1.get('reserved_bytes.all.current', 0) / 2**30, 1)},
            'model_tflops_per_image': fl_sample / 1e12,
            'model_tflops_per_image_survey_formula': fl_formula / 1e12,
            'labelled_rows_per_sample': n_lab,
            'mfma_utilisation_step': value / world * fl_sample / 1e12 / PEAK_BF16_TFLOPS,
            'mfma_utilisation_step_survey_formula': value / world * fl_formula / 1e12 / PEAK_BF16_TFLOPS,
            'mfma_utilisation_note': 'training FLOPs of the work this build executes (lm_head on the labelled rows only; `_survey_formula` counts lm_head over all L rows as SURVEY 8d does) / dense bf16 peak (2.5 PFLOP/s)' + ('; the e4m3 GEMMs of this run have a 5 PFLOP/s peak' if args.fp8 else ''),
        }
        if use_events:
            ms, fl, n = K.prof_collect(hip.PROF_GEMM_BF16)
            alg_bytes = K.prof_last_bytes() / max(n, 1)
            ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            traffic, traffic_src = None, None
            cands = sorted((Path(__file__).resolve().parent / 'profiles').glob('r*_gemm_traffic.json'))
            if cands:          # PMC passes cannot run inside this process: the newest committed rocprofv3 measurement ...
                tf = cands[-1]
                tj = json.loads(tf.read_text())
                digest = gemm_source_digest()
                if tj.get('source_digest') == digest:      # ... if it was taken on THESE kernel sources
                    traffic, traffic_src = tj['bytes_per_launch'], f'profiles/{tf.name}: ' + tj['source']
                else:
                    traffic_src = (f'profiles/{tf.name} is STALE (kernel sources changed since it was measured: digest {tj.get("source_digest")} '
                                   f'vs {digest}; it held {tj["bytes_per_launch"]:.3e} bytes per launch) — re-run tools/pmc_traffic.sh')
            out['roofline'] = {'bound': 'mfma', 'kernel': 'gemm256_k / gemm256w_k / gemm_nt_k<bf16> (vm_gemm_bf16)', 'achieved': ach, 'peak': PEAK_BF16_TFLOPS,
                               'unit': 'TFLOP/s', 'frac': ach / PEAK_BF16_TFLOPS,
                               'peak_measured': peak_measured, 'frac_of_measured': (ach / peak_measured) if peak_measured else None,
                               'peak_measured_note': 'sustained rate of bare v_mfma_f32_16x16x32_bf16 on this device, uniform random operands in registers, two waves per SIMD, all CUs, 2 s (second half timed): vm_ubench_mfma_bf16',
                               'traffic': traffic,
                               'traffic_unit': 'bytes per launch (L2 memory-side, FETCH_SIZE x2 + WRITE_SIZE)', 'traffic_source': traffic_src,
                               'algorithmic_bytes_per_launch': alg_bytes, 'algorithmic_flops_per_launch': fl / max(n, 1),
                               'launches': n, 'launch_sample': f'pseudo-random 1 in {args.event_stride} launches of the timed region', 'avg_launch_ms': ms / max(n, 1), 'kernel_time_share': ms * args.event_stride * 1e-3 / dt,
                               'note': 'algorithmic 2*M*N*(K+K2) FLOPs summed over the bracketed launches of the timed region / their summed HIP-event durations'}
            ms_a, fl_a, n_a = K.prof_collect(hip.PROF_ATTN)
            if n_a:
                stride_a = args.event_stride if not args.all_kernel_events else args.event_stride
                out['attention'] = {'kernel': 'a32::fwd_k (32-query four-cluster forward) / attn16_dkv_k + attn16_dq_ds_k (dQ from the stored dS^T) (vm_attn_*_bf16)', 'achieved_tflops': fl_a / (ms_a * 1e-3) / 1e12,
                                    'frac_of_bf16_peak': fl_a / (ms_a * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 'launches': n_a,
                                    'kernel_time_share': ms_a * stride_a * 1e-3 / dt}
            ms_f, fl_f, n_f = K.prof_collect(hip.PROF_GEMM_F32)
            if n_f:
                out['gemm_f32'] = {'achieved_tflops': fl_f / (ms_f * 1e-3) / 1e12, 'launches': n_f, 'kernel_time_share': ms_f * 1e-3 / dt}
        if also:
            out['also'] = also
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(w, model.config, tok)
            # the other workloads' CPU baselines by the same method (reduced depth x depth ratio), one per distinct workload,
            # while --cpu-baseline-budget lasts; a mixed batch is half 2-D, half 3-D images: the mean time per image of the two shapes
            t_cb, done = time.perf_counter(), {args.workload: out['cpu_baseline']}
            for a in also:
                wl = a['workload'].partition(':')[0]
                if 'value' not in a:
                    continue
                if wl not in done:
                    if time.perf_counter() - t_cb > args.cpu_baseline_budget:
                        done[wl] = {'skipped': f'--cpu-baseline-budget {args.cpu_baseline_budget:.0f} s used up by the workloads before it'}
                    elif WORKLOADS[wl].get('mixed'):
                        parts = [done['phase-vlm-448'] if 'value' in done.get('phase-vlm-448', {}) else cpu_baseline(WORKLOADS['phase-vlm-448'], model.config, tok),
                                 cpu_baseline(dict(WORKLOADS['phase-grg-3d'], sam=False), model.config, tok)]
                        done[wl] = {'value': 2.0 / sum(1.0 / q['value'] for q in parts), 'unit': 'images/s', 'cores': parts[0]['cores'], 'kind': 'port',
                                    'sample': 'half 2-D 448x448, half 3-D 32x256x256 images (text 256): mean time per image of [' + ' | '.join(q['sample'] for q in parts) + ']'}
                    else:
                        done[wl] = cpu_baseline(WORKLOADS[wl], model.config, tok)
                a['cpu_baseline'] = done[wl]
        print(json.dumps(out), file=reserve_stdout(), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
