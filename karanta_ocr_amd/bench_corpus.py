"""A page corpus through the reference's worker surface (python -m karanta_ocr_amd.bench_corpus [--gpus N]).

BASELINE.json config 4 is `bulk_processing.main` over a synthetic corpus with queue-fed workers, one vLLM server per
GPU, reported as throughput + tail latency.  Per GPU that is: W worker threads, each with its own `VLLMClient`
(/root/reference/bulk_processing/workers/inference_worker.py:324-339 — one client per Celery worker process), pulling
pages from a queue and calling `generate(messages, max_tokens=…, temperature=0)` with the page as a PNG
data-URL (`create_vision_message`, /root/reference/karanta/data/utils.py:283-297).  Everything behind the call runs
here: PNG decode, the GPU image front end (resize / normalise / patchify), ViT, prefill into a free slot, the
continuous-batching decode graph.  Output lengths are drawn from U[t_min, t_max] through `max_tokens` (random-init
weights never stop by themselves).  Prints one JSON line: pages/s, tokens/s, p50 / p95 / p99 request latency.

Without --gpus: ONE engine in this process behind `LocalServer` (no HTTP): the per-GPU figure of rounds 1-3.

--gpus N: THE NODE, as one command — config 4 as the reference runs it:
  * N servers started by `launch.start_servers` (`python -m karanta_ocr_amd.cli serve <model dir> --port P`, HIP_VISIBLE_DEVICES=i:
    /root/reference/scripts/start_multiple_vllm_servers.sh:271-310), rank 0 reading the checkpoint ONCE and the weight arena
    going to the other GPUs over RCCL / xGMI; --model-dir names the checkpoint, otherwise a random-init one of --model is
    written in the hub layout first (tools/synthetic_checkpoint.py);
  * the submit loop of /root/reference/bulk_processing/main.py:30-60: every request goes to the port queue
    `dp.LeastLoadedRouter` picks (`GPURouter.get_best_queue`, /root/reference/bulk_processing/utils/gpu_router.py:10-20: fewest
    outstanding tasks), --workers threads per port consume their port's queue over HTTP, one `VLLMClient` each
    (/root/reference/bulk_processing/scripts/start_multiple_celery_workers.sh:254-297: workers bound to a port);
  * one JSON line: node pages/s, per-GPU pages/s, merged p50 / p95 / p99, `rccl_ranks` (what the servers logged about the
    broadcast; null for one server)."""
import argparse
import json
import os
import queue
import sys
import threading
import time

import numpy as np

PROMPT = ("Below is the image of one page of a document. Just return the plain text representation of this document as if "
          "you were reading it naturally.")


GUIDED = (r"---\nprimary_language: (?:[a-z]{2}|null)\nis_rotation_valid: (?:True|False|true|false)\n"
          r"rotation_correction: (?:0|90|180|270)\nis_table: (?:True|False|true|false)\n"
          r"is_diagram: (?:True|False|true|false)\n(?:---|---\n[\s\S]+)")   # karanta/pipeline.py:304-307


def run_corpus(n_pages: int, request, ports, workers_per_port: int, client_kw=None):
    """The reference's dispatch, in one process: a submit loop that sends request i to the port queue with the fewest
    outstanding tasks (dp.LeastLoadedRouter = GPURouter.get_best_queue), `workers_per_port` threads per port that consume
    their port's queue with a VLLMClient of their own.  Returns (wall seconds, latencies, completion tokens, errors,
    requests served per port)."""
    from .clients import VLLMClient
    from .dp import LeastLoadedRouter

    ports = list(ports)
    lock = threading.Lock()
    router = LeastLoadedRouter(ports)
    queues = {f"gpu_queue_{p}": queue.Queue() for p in ports}
    lat, toks, errors = [0.0] * n_pages, [0] * n_pages, []
    served = {p: 0 for p in ports}
    t_submit = [0.0] * n_pages

    def worker(port):
        c = VLLMClient(port=port, max_retries=0, **(client_kw or {}))
        qname = f"gpu_queue_{port}"
        while True:
            i = queues[qname].get()
            if i is None:
                return
            try:
                r = c.generate(**request(i))
                toks[i] = int(r["usage"]["completion_tokens"])
            except Exception as e:  # counted, reported, never hidden
                errors.append(f"page {i} (port {port}): {e}")
            lat[i] = time.perf_counter() - t_submit[i]
            with lock:
                router.done(qname)
                served[port] += 1

    threads = [threading.Thread(target=worker, args=(p,), daemon=True) for p in ports for _ in range(workers_per_port)]
    [t.start() for t in threads]
    t0 = time.perf_counter()
    window = 2 * workers_per_port          # tasks a port queue may hold before the submit loop waits (bounded, as a broker is)
    for i in range(n_pages):
        while True:
            with lock:
                qname = router.get_best_queue()
                if router._outstanding[qname] < window:
                    router.submit(qname)
                    break
            time.sleep(0.0005)
        t_submit[i] = time.perf_counter()
        queues[qname].put(i)
    for q in queues.values():
        for _ in range(workers_per_port):
            q.put(None)
    [t.join() for t in threads]
    return time.perf_counter() - t0, lat, toks, errors, served


def node_main(args) -> int:
    """--gpus N (see the module docstring)."""
    import re
    import tempfile

    from . import image_processing as IP
    from . import launch
    from .clients import VLLMClient
    from .config import CONFIGS

    log = lambda m: print(m, file=sys.stderr, flush=True)
    cfg = CONFIGS[args.model]
    tmp = None
    model_dir = args.model_dir
    if model_dir is None:
        from .tools.synthetic_checkpoint import write_checkpoint
        tmp = tempfile.mkdtemp(prefix="karanta_ckpt_")
        model_dir = os.path.join(tmp, "karantaocr-" + args.model.lower())
        t0 = time.perf_counter()
        write_checkpoint(model_dir, cfg, 0, "v4", max_pixels=1003520)
        log(f"[bench_corpus] wrote a random-init {args.model} checkpoint in the hub layout to {model_dir} in {time.perf_counter() - t0:.0f}s")
    gpus = list(range(args.gpus))
    ports = [launch.free_port() for _ in gpus]
    rng = np.random.default_rng(11)
    limits = rng.integers(args.t_min, args.t_max + 1, size=args.pages).tolist()
    urls = [IP.encode_png_data_url(IP.synthetic_page(100 + i, args.page, args.page)) for i in range(args.distinct)]
    s_len = (2048 + args.t_max + args.chunk + 127) // 64 * 64
    extra = ["--served-model-name", "karantaocr", "--max-num-seqs", str(args.slots), "--max-model-len", str(s_len), "--greedy",
             "--admit-min", str(args.admit_min), "--admit-max-wait", str(args.admit_max_wait), "--max-tokens-cap", str(args.t_max),
             "--host", "127.0.0.1"] + (["--host-images"] if args.host_images else [])
    log_dir = args.log_dir or os.path.join(tmp or tempfile.mkdtemp(prefix="karanta_logs_"), "vllm_logs")
    group = launch.start_servers(model_dir, gpus, ports, extra, log_dir, timeout_s=args.timeout, log=log, preflight=not args.no_preflight,
                                 server_cmd=args.server_cmd.split() if args.server_cmd else None)
    try:
        def request(i):
            kw = {"guided_regex": GUIDED} if args.guided else {}
            return dict(messages=[{"role": "user", "content": [{"type": "text", "text": PROMPT},
                                                                {"type": "image_url", "image_url": {"url": urls[i % len(urls)]}}]}],
                        max_tokens=int(limits[i]), temperature=0.0, **kw)

        for p in ports:                       # warm-up of every server: kernel attributes, graph capture, guide tables
            VLLMClient(port=p, host="127.0.0.1").generate(**request(0))
        wall, lat, toks, errors, served = run_corpus(args.pages, request, ports, args.workers, {"host": "127.0.0.1"})
    finally:
        codes = group.stop()
    rccl = None
    for lg in group.logs:                     # cli.make_server logs the broadcast it took part in
        try:
            m = re.search(r"weight broadcast: .* over (\d+) RCCL ranks", open(lg, errors="replace").read())
        except OSError:
            m = None
        if m:
            rccl = int(m.group(1)) if rccl is None else min(rccl, int(m.group(1)))
    la = np.sort(np.asarray(lat))
    pct = lambda q: round(float(la[min(len(la) - 1, int(q * len(la)))]), 3)
    print(json.dumps({
        "metric": "pages_per_sec (node) + request latency, BASELINE config 4's shape",
        "workload": f"{args.model}, {args.pages} requests over {args.distinct} synthetic {args.page}x{args.page} PNG scans, submit loop -> "
                    f"least-loaded port queue -> {args.workers} VLLMClient worker threads per port -> HTTP -> {len(ports)} x "
                    f"`karanta_ocr_amd.cli serve` (one per GPU, {args.slots} decode slots each), max_tokens U[{args.t_min},{args.t_max}] "
                    f"(mean {np.mean(limits):.0f}), admit_min {args.admit_min}, {'guided_regex' if args.guided else 'greedy'}, "
                    f"{'checkpoint ' + args.model_dir if args.model_dir else 'random-init weights in the hub layout'}",
        "n_gpus": len(ports), "pages_per_s": round(args.pages / wall, 3), "tokens_per_s": round(sum(toks) / wall, 1), "wall_s": round(wall, 2),
        "per_gpu_pages_per_s": [round(served[p] / wall, 3) for p in ports], "per_gpu_requests": [served[p] for p in ports],
        "latency_s": {"p50": pct(0.50), "p95": pct(0.95), "p99": pct(0.99), "max": round(float(la[-1]), 3)},
        "completion_tokens": int(sum(toks)), "expected_tokens": int(sum(limits)) if not args.guided else None,
        "rccl_ranks": rccl, "weights": ("rank 0 read the checkpoint once; RCCL broadcast to the other servers" if len(ports) > 1
                                        else "one server: it read the checkpoint itself"),
        "server_exit_codes": codes, "errors": errors[:5], "n_errors": len(errors),
    }), flush=True)
    if tmp and not args.keep:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    return 1 if errors else 0



def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="Qwen2-VL-2B")
    ap.add_argument("--pages", type=int, default=256)
    ap.add_argument("--workers", type=int, default=32)
    ap.add_argument("--slots", type=int, default=32)
    ap.add_argument("--t-min", type=int, default=64)
    ap.add_argument("--t-max", type=int, default=512)
    ap.add_argument("--chunk", type=int, default=2,
                    help="decode steps between two looks at the finished flags (r4: 2 with launch-ahead; the r3 loop wanted 6-8)")
    ap.add_argument("--page", type=int, default=1024)
    ap.add_argument("--distinct", type=int, default=16, help="distinct synthetic scans (encoded once, cycled)")
    ap.add_argument("--admit", type=int, default=8, help="pages one admission may prefill together")
    ap.add_argument("--admit-min", type=int, default=6, help="free slots (and waiting requests) an admission waits for while others decode")
    ap.add_argument("--admit-max-wait", type=int, default=48, help="... but never longer than this many scheduler steps (of --chunk decode steps)")
    ap.add_argument("--overlap-cus", type=int, default=None,
                    help="overlap admissions with decoding: ViT + prefill on a stream restricted to this many CUs "
                         "(kr_stream_create_cu_mask; 0 = an ordinary second stream; unset = admissions interrupt the decode graph)")
    ap.add_argument("--no-launch-ahead", action="store_true",
                    help="the r3 loop: wait for a chunk's flags before queueing the next one (use with --chunk 8 --admit-max-wait 8)")
    ap.add_argument("--host-images", action="store_true", help="PIL resize on the host instead of the GPU front end")
    ap.add_argument("--upload-at-admission", action="store_true",
                    help="r3 behaviour: the decoded page crosses PCIe at admission, on the scheduler thread (default: at parse time, in "
                         "the request's thread, on a stream of its own)")
    ap.add_argument("--guided", action="store_true", help="every request carries the pipeline's guided_regex")
    ap.add_argument("--gpus", type=int, default=None,
                    help="the node: N servers through launch.py (real cli, one checkpoint read + RCCL broadcast), least-loaded routing, "
                         "--workers client threads PER PORT, over HTTP (see the module docstring); unset: one engine in this process")
    ap.add_argument("--model-dir", default=None, help="--gpus: a checkpoint directory (default: a random-init one of --model is written)")
    ap.add_argument("--server-cmd", default=None, help="--gpus: the server command instead of `python -m karanta_ocr_amd.cli` (tests)")
    ap.add_argument("--log-dir", default=None)
    ap.add_argument("--timeout", type=float, default=600.0, help="--gpus: seconds to wait for the servers' /health")
    ap.add_argument("--no-preflight", action="store_true")
    ap.add_argument("--keep", action="store_true", help="--gpus: keep the written checkpoint / logs")
    args = ap.parse_args(argv)
    if args.gpus is not None:
        return node_main(args)

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
    front = S.ChatFrontend(cfg, tok, max_pixels=1003520, device_images=not args.host_images,
                           upload_device=None if (args.host_images or args.upload_at_admission) else "cuda:0")
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
                        overlap_admissions=args.overlap_cus is not None, launch_ahead=not args.no_launch_ahead)
    port = 8791
    S.register_local_server(port, srv)
    guided = GUIDED

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
    sched = srv.scheduler_stats()                         # (includes the warm-up request's admission and steps)
    S.unregister_local_server(port)
    srv.close()
    la = np.sort(np.asarray(lat))
    pct = lambda q: round(float(la[min(len(la) - 1, int(q * len(la)))]), 3)
    print(json.dumps({
        "workload": f"{args.model}, {args.pages} requests over {args.distinct} synthetic {args.page}x{args.page} PNG scans through "
                    f"VLLMClient.generate -> LocalServer(continuous), {args.workers} worker threads, {B} decode slots, "
                    f"max_tokens U[{args.t_min},{args.t_max}] (mean {np.mean(limits):.0f}), prompt {P} tokens, "
                    f"admit_min {args.admit_min} (max wait {args.admit_max_wait} x {args.chunk} steps), "
                    f"{'chunks queued one ahead, ' if not args.no_launch_ahead and args.overlap_cus is None else ''}"
                    f"{'admissions interrupt the decode graph' if args.overlap_cus is None else 'admissions overlapped on ' + (str(args.overlap_cus) + ' CUs' if args.overlap_cus else 'an unmasked second stream')}, "
                    f"{'host PIL' if args.host_images else 'GPU'} image front end, {'guided_regex' if args.guided else 'greedy'}, "
                    f"random-init weights",
        "pages_per_s": round(args.pages / wall, 3), "tokens_per_s": round(sum(toks) / wall, 1), "wall_s": round(wall, 2),
        "latency_s": {"p50": pct(0.50), "p95": pct(0.95), "p99": pct(0.99), "max": round(float(la[-1]), 3)},
        "completion_tokens": int(sum(toks)), "expected_tokens": int(sum(limits)) if not args.guided else None,
        "scheduler": sched, "errors": errors[:5], "n_errors": len(errors),
    }), flush=True)
    eng.close()
    return 1 if errors else 0


if __name__ == "__main__":
    sys.exit(main())
