"""Static vs continuous batching on a ragged workload (python -m karanta_ocr_amd.bench_serving).

N synthetic 1024x1024 pages, output lengths drawn uniformly from [t_min, t_max] (the reference's pages end at EOS
anywhere up to max_tokens 4000, karanta/pipeline.py:124), B decode slots on one GPU, random-init Qwen2-VL-2B.
Static: pages in arrival order, B at a time, each batch runs to its longest member (Engine.generate).
Continuous: scheduler.SlotScheduler refills a slot as soon as its page is done.  Prints one JSON line."""
import argparse
import json
import sys
import time

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="Qwen2-VL-2B")
    ap.add_argument("--pages", type=int, default=32)
    ap.add_argument("--slots", type=int, default=8)
    ap.add_argument("--t-min", type=int, default=64)
    ap.add_argument("--t-max", type=int, default=1024)
    ap.add_argument("--chunk", type=int, default=16)
    ap.add_argument("--page", type=int, default=1024)
    args = ap.parse_args()

    import torch  # noqa: F401  (same HIP runtime instance as the extension)
    from karanta_ocr_amd import image_processing as IP
    from karanta_ocr_amd.config import CONFIGS
    from karanta_ocr_amd.engine import Engine, PageRequest
    from karanta_ocr_amd.scheduler import SlotRequest, SlotScheduler
    from karanta_ocr_amd.weights import random_weights
    sys.path.insert(0, ".")
    from bench import build_prompt

    cfg = CONFIGS[args.model]
    rng = np.random.default_rng(7)
    limits = rng.integers(args.t_min, args.t_max + 1, size=args.pages).tolist()
    pages = []
    for i in range(args.pages):
        pv, g = IP.image_to_patches(IP.synthetic_page(i, args.page, args.page), max_pixels=1003520)
        pages.append(PageRequest(build_prompt(cfg, g[1] * g[2] // 4, rng), pv, [g]))
    P = max(len(p.input_ids) for p in pages)
    B = args.slots
    eng = Engine(cfg, max_batch=B, s_max=(P + args.t_max + args.chunk + 63) // 64 * 64,
                 max_patches=B * len(pages[0].pixel_values), max_prompt_tokens=B * P)
    eng.load_weights(random_weights(cfg, 0, as_bits=True))
    # random weights never emit EOS on purpose: make sure none of the eos ids can stop a page early
    eng.generate(pages[:1], 2)  # warm-up (kernel attributes, graphs are built lazily)

    t0 = time.perf_counter()
    static_tokens = 0
    for i in range(0, args.pages, B):
        res = eng.generate(pages[i:i + B], max(limits[i:i + B]), ignore_eos=True)
        static_tokens += sum(min(len(t), m) for t, m in zip(res.tokens, limits[i:i + B]))
    t_static = time.perf_counter() - t0

    sch = SlotScheduler(eng, max_tokens_cap=args.t_max, chunk=args.chunk, eos_token_ids=())
    t0 = time.perf_counter()
    out = sch.run([SlotRequest(p, m, tag=i) for i, (p, m) in enumerate(zip(pages, limits))])
    t_cont = time.perf_counter() - t0
    assert all(r.error is None and len(r.tokens) == m for r, m in zip(out, limits))
    # the same with overlapped admission (ViT + prefill of the next page on a second stream)
    sch2 = SlotScheduler(eng, max_tokens_cap=args.t_max, chunk=args.chunk, eos_token_ids=(), overlap=True)
    t0 = time.perf_counter()
    out2 = sch2.run([SlotRequest(p, m, tag=i) for i, (p, m) in enumerate(zip(pages, limits))])
    t_over = time.perf_counter() - t0
    assert all(r.error is None and np.array_equal(r.tokens, q.tokens) for r, q in zip(out2, out))
    print(json.dumps({
        "workload": f"{args.model}, {args.pages} synthetic {args.page}x{args.page} pages, {B} slots, output lengths "
                    f"U[{args.t_min},{args.t_max}] (mean {np.mean(limits):.0f}), random-init weights, greedy",
        "static_pages_per_s": round(args.pages / t_static, 3), "continuous_pages_per_s": round(args.pages / t_cont, 3),
        "continuous_overlapped_admission_pages_per_s": round(args.pages / t_over, 3),
        "speedup": round(t_static / t_cont, 3), "speedup_overlapped": round(t_static / t_over, 3), "tokens": int(sum(limits)), "static_tokens_checked": int(static_tokens),
        "continuous_slot_utilisation": round(sch.slot_steps_busy / max(1, sch.steps * B), 3),
        "decode_steps": {"continuous": sch.steps}, "chunk": args.chunk,
    }), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
