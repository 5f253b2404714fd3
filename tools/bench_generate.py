"""Generation path at true width: CogVLM-7B + EVA-ViT-E + r64 LoRA, random-init bf16, greedy decoding.

    python tools/bench_generate.py [--batch 1 8] [--new 64] [--image 448]

Reports prefill time and decode tokens/s. Decode is HBM-bound: one step reads the language-expert weights of every layer
(the vision expert is never touched by a single-token call), the LoRA factors, lm_head, and the K / V rows of the cache:
`roofline.achieved` = those algorithmic bytes / step time, against 8 TB/s. Secondary measurement — not the headline metric
of bench.py (train images/s)."""
import argparse
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402


def decode_bytes(model, B: int, ctx: int) -> float:
    cfg = model.config
    per_layer = 0
    layer = model.model.layers[0]
    for n, p in layer.named_parameters():
        if 'vision_expert' in n or 'vision_mlp' in n:
            continue
        per_layer += p.numel() * p.element_size()
    head = model.lm_head.weight.numel() * model.lm_head.weight.element_size()
    kv = 2 * ctx * cfg.hidden_size * 2 * B
    return cfg.num_hidden_layers * (per_layer + kv) + head + model.model.norm.weight.numel() * 2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, nargs='+', default=[1, 8])
    ap.add_argument('--new', type=int, default=64)
    ap.add_argument('--text', type=int, default=64)
    ap.add_argument('--depth-scale', type=float, default=1.0)
    ap.add_argument('--graph', type=int, default=1)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    w = dict(bench.WORKLOADS['phase-vlm-448'])
    model, tok = bench.build(w, dev, args.depth_scale)
    model.eval()
    for B in args.batch:
        batch = bench.make_batch(w, tok, B, dev, seed=B)
        vi = batch['vlm_inputs']
        kw = dict(token_type_ids=vi['token_type_ids'], position_ids=vi['position_ids'], attention_mask=vi['attention_mask'],
                  image=batch['image'], patch_size=batch['patch_size'], pool_size=batch['pool_size'])
        model.generate(vi['input_ids'], **kw, max_new_tokens=4)            # warm-up (lazy kernel attributes, allocator)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.generate(vi['input_ids'], **kw, max_new_tokens=1)
        torch.cuda.synchronize()
        t_prefill = time.perf_counter() - t0
        t0 = time.perf_counter()
        out = model.generate(vi['input_ids'], **kw, max_new_tokens=args.new + 1, use_graph=bool(args.graph)) if args.graph else \
            model.generate(vi['input_ids'], **kw, max_new_tokens=args.new + 1)
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        step = (t_all - t_prefill) / args.new
        ctx = int(vi['attention_mask'].sum(1).max()) + args.new // 2
        nbytes = decode_bytes(model, B, ctx)
        print(json.dumps({
            'metric': 'decode tokens/sec', 'value': B / step, 'unit': 'tokens/s', 'batch': B, 'new_tokens': args.new,
            'ms_per_step': step * 1e3, 'prefill_ms': t_prefill * 1e3, 'prompt_tokens': int(vi['attention_mask'].sum(1).max()),
            'dtype': 'bf16', 'data': 'synthetic', 'weights': 'random-init', 'hip_graph': bool(args.graph),
            'roofline': {'bound': 'hbm', 'achieved': nbytes / step / 1e9, 'peak': 8000.0, 'unit': 'GB/s',
                         'frac': nbytes / step / 8e12, 'algorithmic_bytes_per_step': nbytes},
            'tokens': out.new_tokens[0, :8].tolist(),
        }), flush=True)


if __name__ == '__main__':
    main()
