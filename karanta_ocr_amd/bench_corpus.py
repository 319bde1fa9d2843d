"""A page corpus through the reference's worker surface on one GPU (python -m karanta_ocr_amd.bench_corpus).

BASELINE.json config 4 is `bulk_processing.main` over a synthetic corpus with queue-fed workers, one vLLM server per
GPU, reported as throughput + tail latency.  Per GPU that is: W worker threads, each with its own `VLLMClient`
(/root/reference/bulk_processing/workers/inference_worker.py:324-339 — one client per Celery worker process), pulling
pages from a shared queue and calling `generate(messages, max_tokens=…, temperature=0)` with the page as a PNG
data-URL (`create_vision_message`, /root/reference/karanta/data/utils.py:283-297).  Everything behind the call runs
here: PNG decode, the GPU image front end (resize / normalise / patchify), ViT, prefill into a free slot, the
continuous-batching decode graph.  Output lengths are drawn from U[t_min, t_max] through `max_tokens` (random-init
weights never stop by themselves).  Prints one JSON line: pages/s, tokens/s, p50 / p95 / p99 request latency.
The 8-GPU figure of config 4 is this process once per GPU (pages are independent; dp.LeastLoadedRouter or the
reference's Redis queues feed them)."""
import argparse
import json
import queue
import threading
import time

import numpy as np

PROMPT = ("Below is the image of one page of a document. Just return the plain text representation of this document as if "
          "you were reading it naturally.")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="Qwen2-VL-2B")
    ap.add_argument("--pages", type=int, default=256)
    ap.add_argument("--workers", type=int, default=32)
    ap.add_argument("--slots", type=int, default=32)
    ap.add_argument("--t-min", type=int, default=64)
    ap.add_argument("--t-max", type=int, default=512)
    ap.add_argument("--chunk", type=int, default=8, help="decode steps between two looks at the finished flags (r3 sweep: 6-8 best)")
    ap.add_argument("--page", type=int, default=1024)
    ap.add_argument("--distinct", type=int, default=16, help="distinct synthetic scans (encoded once, cycled)")
    ap.add_argument("--admit", type=int, default=8, help="pages one admission may prefill together")
    ap.add_argument("--admit-min", type=int, default=6, help="free slots (and waiting requests) an admission waits for while others decode")
    ap.add_argument("--admit-max-wait", type=int, default=8, help="... but never longer than this many scheduler steps")
    ap.add_argument("--overlap-cus", type=int, default=None,
                    help="overlap admissions with decoding: ViT + prefill on a stream restricted to this many CUs "
                         "(kr_stream_create_cu_mask; 0 = an ordinary second stream; unset = admissions interrupt the decode graph)")
    ap.add_argument("--host-images", action="store_true", help="PIL resize on the host instead of the GPU front end")
    ap.add_argument("--guided", action="store_true", help="every request carries the pipeline's guided_regex")
    args = ap.parse_args()

    import torch  # noqa: F401
    from karanta_ocr_amd import image_processing as IP
    from karanta_ocr_amd import serving as S
    from karanta_ocr_amd.clients import VLLMClient
    from karanta_ocr_amd.config import CONFIGS
    from karanta_ocr_amd.engine import Engine
    from karanta_ocr_amd.weights import random_weights

    cfg = CONFIGS[args.model]
    rng = np.random.default_rng(11)
    limits = rng.integers(args.t_min, args.t_max + 1, size=args.pages).tolist()
    urls = [IP.encode_png_data_url(IP.synthetic_page(100 + i, args.page, args.page)) for i in range(args.distinct)]
    tok = S.ByteTokenizer(cfg)
    front = S.ChatFrontend(cfg, tok, max_pixels=1003520, device_images=not args.host_images)
    probe = front.parse({"messages": [{"role": "user", "content": [{"type": "text", "text": PROMPT},
                                                                    {"type": "image_url", "image_url": {"url": urls[0]}}]}],
                         "max_tokens": 1})
    P = len(probe.input_ids)
    n_patch = int(np.prod(probe.grids[0]))
    B = args.slots
    eng = Engine(cfg, max_batch=B, s_max=(P + args.t_max + args.chunk + 64 + 63) // 64 * 64,
                 max_patches=args.admit * n_patch, max_prompt_tokens=args.admit * P, admission_cus=args.overlap_cus or None)
    eng.load_weights(random_weights(cfg, 0, as_bits=True))
    srv = S.LocalServer(eng, front, log=lambda *_: None, continuous=True, max_tokens_cap=args.t_max, chunk=args.chunk,
                        honor_temperature=False, admit_min=args.admit_min, admit_max_wait=args.admit_max_wait,
                        overlap_admissions=args.overlap_cus is not None)
    port = 8791
    S.register_local_server(port, srv)
    guided = (r"---\nprimary_language: (?:[a-z]{2}|null)\nis_rotation_valid: (?:True|False|true|false)\n"
              r"rotation_correction: (?:0|90|180|270)\nis_table: (?:True|False|true|false)\n"
              r"is_diagram: (?:True|False|true|false)\n(?:---|---\n[\s\S]+)")   # karanta/pipeline.py:304-307

    def request(i):
        kw = {"guided_regex": guided} if args.guided else {}
        return dict(messages=[{"role": "user", "content": [{"type": "text", "text": PROMPT},
                                                            {"type": "image_url", "image_url": {"url": urls[i % len(urls)]}}]}],
                    max_tokens=int(limits[i]), temperature=0.0, **kw)

    VLLMClient(port=port).generate(**request(0))          # warm-up: kernel attributes, graph capture, guide tables
    todo: "queue.Queue" = queue.Queue()
    for i in range(args.pages):
        todo.put(i)
    lat, toks, errors = [0.0] * args.pages, [0] * args.pages, []

    def worker():
        c = VLLMClient(port=port, max_retries=0)
        while True:
            try:
                i = todo.get_nowait()
            except queue.Empty:
                return
            t0 = time.perf_counter()
            try:
                r = c.generate(**request(i))
                toks[i] = int(r["usage"]["completion_tokens"])
            except Exception as e:  # counted, reported, never hidden
                errors.append(f"page {i}: {e}")
            lat[i] = time.perf_counter() - t0

    t0 = time.perf_counter()
    ts = [threading.Thread(target=worker) for _ in range(args.workers)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    wall = time.perf_counter() - t0
    S.unregister_local_server(port)
    srv.close()
    la = np.sort(np.asarray(lat))
    pct = lambda q: round(float(la[min(len(la) - 1, int(q * len(la)))]), 3)
    print(json.dumps({
        "workload": f"{args.model}, {args.pages} requests over {args.distinct} synthetic {args.page}x{args.page} PNG scans through "
                    f"VLLMClient.generate -> LocalServer(continuous), {args.workers} worker threads, {B} decode slots, "
                    f"max_tokens U[{args.t_min},{args.t_max}] (mean {np.mean(limits):.0f}), prompt {P} tokens, "
                    f"admit_min {args.admit_min} (max wait {args.admit_max_wait} x {args.chunk} steps), "
                    f"{'admissions interrupt the decode graph' if args.overlap_cus is None else 'admissions overlapped on ' + (str(args.overlap_cus) + ' CUs' if args.overlap_cus else 'an unmasked second stream')}, "
                    f"{'host PIL' if args.host_images else 'GPU'} image front end, {'guided_regex' if args.guided else 'greedy'}, "
                    f"random-init weights",
        "pages_per_s": round(args.pages / wall, 3), "tokens_per_s": round(sum(toks) / wall, 1), "wall_s": round(wall, 2),
        "latency_s": {"p50": pct(0.50), "p95": pct(0.95), "p99": pct(0.99), "max": round(float(la[-1]), 3)},
        "completion_tokens": int(sum(toks)), "expected_tokens": int(sum(limits)) if not args.guided else None,
        "errors": errors[:5], "n_errors": len(errors),
    }), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
