"""``vllm serve``-shaped command line for the MI355X engine.

The reference starts its model server itself: ``karanta.pipeline`` builds
``vllm serve <model> --port N --disable-log-requests --uvicorn-log-level warning --served-model-name karantaocr
--tensor-parallel-size T --data-parallel-size D --limit-mm-per-prompt '{"video": 0}' [--gpu-memory-utilization F]
[--max-model-len M] [passthrough...]`` (/root/reference/karanta/pipeline.py:707-734), and its shell scripts run
``python -m vllm.entrypoints.openai.api_server --model <model> --port N --dtype bfloat16 [--max-model-len M]
[--trust-remote-code]`` once per GPU (scripts/start_multiple_vllm_servers.sh:283-294).  This module accepts both
spellings, so an executable named ``vllm`` that runs ``python -m karanta_ocr_amd.cli "$@"`` (or replacing
``vllm.entrypoints.openai.api_server`` by ``karanta_ocr_amd.cli`` in the scripts) puts this engine behind the
unmodified pipeline: same port, same served model name, the log lines the pipeline waits for on the server's
output (:769-800), SIGTERM / SIGINT to stop (:745-751).

vLLM flags with no meaning here are accepted and ignored (tensor / data parallel sizes other than 1 are refused:
one process drives one GPU, as the reference's own multi-GPU script does; select the GPU with HIP_VISIBLE_DEVICES).
"""
from __future__ import annotations

import argparse
import os
import signal
import sys
import threading
from typing import List, Optional


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="vllm", description=__doc__.split("\n\n")[0])
    ap.add_argument("command", nargs="?", default=None, help="`serve` (vllm serve <model> ...) or omitted (api_server style)")
    ap.add_argument("model_pos", nargs="?", default=None, metavar="model", help="model directory (config.json, *.safetensors, tokenizer.json)")
    ap.add_argument("--model", default=None, help="api_server spelling of the model directory")
    ap.add_argument("--port", type=int, default=8000)
    ap.add_argument("--host", default="0.0.0.0")
    ap.add_argument("--served-model-name", default=None)
    ap.add_argument("--fp8-activations", action="store_true",
                    help="fp8 checkpoints: W8A8 prefill (dynamic per-token e4m3 activations on the fp8 matrix instruction, as vLLM "
                         "serves them; prefill GEMMs 1.9x faster, logits move by a few per cent of their range) — default: bf16 activations")
    ap.add_argument("--max-model-len", type=int, default=16384)
    ap.add_argument("--max-num-seqs", type=int, default=8, help="decode slots (<= 32; above 16: hidden_size <= 2048 or == 3584)")
    ap.add_argument("--max-num-batched-tokens", type=int, default=None,
                    help="prompt tokens one admission (ViT + prefill of the requests entering free slots together) may hold: sizes "
                         "the engine's activation buffers (vLLM's flag for its prefill budget).  Default: max(--max-model-len, 16384) — "
                         "eight 1024x1024 pages at the class-default max_pixels; image patches per admission: 4x this")
    ap.add_argument("--tensor-parallel-size", type=int, default=1)
    ap.add_argument("--data-parallel-size", type=int, default=1)
    ap.add_argument("--gpu-memory-utilization", type=float, default=None)
    ap.add_argument("--dtype", default="bfloat16")
    ap.add_argument("--limit-mm-per-prompt", default=None)
    ap.add_argument("--uvicorn-log-level", default=None)
    ap.add_argument("--api-key", default=None)
    ap.add_argument("--task", default=None)
    ap.add_argument("--disable-log-requests", action="store_true")
    ap.add_argument("--trust-remote-code", action="store_true")
    ap.add_argument("--enforce-eager", action="store_true")
    # this engine's own knobs
    ap.add_argument("--max-tokens-cap", type=int, default=6000, help="largest max_tokens a request may ask for")
    ap.add_argument("--static-batching", action="store_true", help="static batches instead of the slot scheduler")
    ap.add_argument("--host-images", action="store_true", help="PIL resize on the host instead of the GPU image front end")
    ap.add_argument("--max-pixels", type=int, default=None,
                    help="override the checkpoint's preprocessor_config.json (default without one: 1003520)")
    ap.add_argument("--min-pixels", type=int, default=None)
    ap.add_argument("--greedy", action="store_true", help="ignore request temperatures")
    ap.add_argument("--admit-min", type=int, default=1,
                    help="while sequences decode, wait for this many free slots + waiting requests before an admission "
                         "(6: +8 %% pages/s on a saturated server, profiles/r03_corpus_sweep.txt; 1: lowest latency)")
    ap.add_argument("--admit-max-wait", type=int, default=48, help="... but at most this many scheduler steps of 2 decode steps")
    ap.add_argument("--quantization", default=None, choices=("fp8",),
                    help="decoder Linears as fp8 codes + row scales (vLLM's flag; implied by a checkpoint with a quantization_config)")
    ap.add_argument("--max-logprobs", type=int, default=None,
                    help="record log-probabilities in the decode graph: the largest top_logprobs a request may ask for (0..20)")
    return ap


def parse_args(argv: Optional[List[str]] = None):
    ap = build_parser()
    args, unknown = ap.parse_known_args(argv)
    if args.command is not None and args.command != "serve":
        if args.model_pos is None and args.model is None:   # `cli.py <model>` without the verb
            args.model_pos, args.command = args.command, "serve"
        else:
            ap.error(f"unknown command {args.command!r} (only `serve`)")
    model = args.model_pos or args.model
    if not model:
        ap.error("no model directory given")
    if args.tensor_parallel_size != 1 or args.data_parallel_size != 1:
        ap.error("one process serves one GPU: run one server per GPU (HIP_VISIBLE_DEVICES=i), as "
                 "scripts/start_multiple_vllm_servers.sh does; tensor / data parallel sizes must be 1")
    if args.max_logprobs is not None and not 0 <= args.max_logprobs <= 20:
        ap.error("--max-logprobs must be in 0..20")
    if not 1 <= args.max_num_seqs <= 32:
        ap.error("--max-num-seqs must be in 1..32 (above 16: hidden_size <= 2048 or == 3584)")
    if args.max_num_batched_tokens is not None and args.max_num_batched_tokens < args.max_model_len:
        ap.error("--max-num-batched-tokens must be >= --max-model-len (the longest prompt one request may carry)")
    args.model_dir = model
    args.served_model_name = args.served_model_name or os.path.basename(os.path.normpath(model))
    args.ignored = unknown
    return args


def _checkpoint_is_fp8(model_dir: str) -> bool:
    """config.json carries a quantization_config for fp8 weights (compressed-tensors / fp8 methods)."""
    import json
    try:
        with open(os.path.join(model_dir, "config.json")) as f:
            q = json.load(f).get("quantization_config") or {}
    except OSError:
        return False
    blob = json.dumps(q).lower()
    return bool(q) and ("fp8" in blob or "float8" in blob or '"num_bits": 8' in blob and '"type": "float"' in blob)


def preprocessor_pixels(model_dir: str):
    """(min_pixels, max_pixels) of the checkpoint's ``preprocessor_config.json`` — what vLLM's Qwen2-VL processor uses
    for every request (the shipped file is primary; the transformers class default, 1 003 520, only
    applies without it).  Both spellings: top-level ``min_pixels`` / ``max_pixels`` (transformers 4) and
    ``size = {"shortest_edge": min, "longest_edge": max}`` (transformers >= 4.49 fast processors).  Missing file or keys
    -> (None, None)."""
    import json
    try:
        with open(os.path.join(model_dir, "preprocessor_config.json")) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None, None
    size = d.get("size") if isinstance(d.get("size"), dict) else {}
    lo = d.get("min_pixels", size.get("shortest_edge"))
    hi = d.get("max_pixels", size.get("longest_edge"))
    ok = lambda v: int(v) if isinstance(v, (int, float)) and v > 0 else None
    return ok(lo), ok(hi)


def admission_budget(args, cfg, max_pixels: int):
    """(max_prompt_tokens, max_patches) of the engine: what ONE admission (the requests that enter free slots together: one
    ViT + prefill pass) may hold — not a multiple of the slot count.  The scheduler fills an admission up to these budgets
    (scheduler.SlotScheduler._admit) and a request larger than them is refused with 400, so the token budget is at least
    --max-model-len; a prompt holds at most as many image tokens as tokens, i.e. 4x as many patches (2x2 merge), plus one
    partial 64-patch block per page.  (Round 3 sized both by max_num_seqs: 2.1 M patches / 262 k tokens of activation
    buffers — tens of GB at the 7B width — for 32 slots with a hub preprocessor_config.json.)"""
    tokens = int(args.max_num_batched_tokens or max(args.max_model_len, 16384))
    merge2 = cfg.vision.spatial_merge_size ** 2
    per_page = min(max_pixels // (cfg.vision.patch_size ** 2), merge2 * args.max_model_len) + 64
    patches = max(merge2 * tokens + 64 * args.max_num_seqs, per_page)
    return tokens, patches


def make_server(args, log=print):
    """Engine + front end + LocalServer from parsed arguments (weights and tokenizer from args.model_dir)."""
    from . import image_processing as IP
    from .engine import Engine
    from .serving import ChatFrontend, HFTokenizer, LocalServer, load_chat_template
    from .weights import load_checkpoint

    from .dp import load_or_receive_weights, serving_group_env
    from .weights import load_config

    rank, world = serving_group_env()      # (0, 1) unless launch.py started this server as one of a node's group
    cfg = load_config(args.model_dir)
    weight_dtype = "fp8" if (args.quantization == "fp8" or _checkpoint_is_fp8(args.model_dir)) else "bf16"
    # image size bounds: command line > the checkpoint's preprocessor_config.json > the transformers class default
    ck_min, ck_max = preprocessor_pixels(args.model_dir)
    min_pixels = args.min_pixels or ck_min or IP.MIN_PIXELS
    max_pixels = args.max_pixels or ck_max or IP.MAX_PIXELS_CLASS_DEFAULT
    if min_pixels > max_pixels:
        raise ValueError(f"min_pixels {min_pixels} > max_pixels {max_pixels}")
    log(f"image preprocessing: min_pixels={min_pixels} max_pixels={max_pixels} "
        f"({'--max-pixels' if args.max_pixels else 'preprocessor_config.json' if ck_max else 'class default'})")
    max_prompt_tokens, max_patches = admission_budget(args, cfg, max_pixels)
    log(f"admission budget: {max_prompt_tokens} prompt tokens / {max_patches} image patches per ViT + prefill pass, "
        f"{args.max_num_seqs} decode slots of {args.max_model_len} tokens")
    # cache rows per slot: the model length + the steps a slot may run past its limit before the scheduler looks (2 chunks of up to 8
    # with launch-ahead) + the parking row
    eng = Engine(cfg, device="cuda:0", max_batch=args.max_num_seqs, s_max=(args.max_model_len + 17 + 63) // 64 * 64,
                 max_patches=max_patches, max_prompt_tokens=max_prompt_tokens,
                 weight_dtype=weight_dtype, fp8_activations=bool(getattr(args, "fp8_activations", False)) or None)
    # one server: read the checkpoint.  A launch.py group: rank 0 reads it ONCE, the arena goes to the other GPUs over
    # RCCL / xGMI (north_star: "RCCL broadcast of weights over xGMI"); a failure stops every server of the group
    info = load_or_receive_weights(eng.w, rank, world, lambda: eng.load_weights(load_checkpoint(args.model_dir)[1]),
                                   stream=eng.s, log=log)
    if world > 1:
        log(f"weight broadcast: {info['bytes'] / 1e9:.2f} GB in {info['bcast_s']:.3f}s over {info['rccl_ranks']} RCCL ranks")
    template = load_chat_template(args.model_dir)
    log("chat template: " + ("the checkpoint's own (chat_template.json / tokenizer_config.json)" if template
                             else "none shipped with the checkpoint: hand-coded Qwen2-VL template"))
    front = ChatFrontend(cfg, HFTokenizer(os.path.join(args.model_dir, "tokenizer.json"), cfg), min_pixels=min_pixels,
                         max_pixels=max_pixels,
                         max_model_len=args.max_model_len, device_images=not args.host_images, chat_template=template,
                         upload_device=None if args.host_images else "cuda:0")
    return LocalServer(eng, front, served_model_name=args.served_model_name, log=log, continuous=not args.static_batching,
                       max_tokens_cap=min(args.max_tokens_cap, args.max_model_len), honor_temperature=not args.greedy,
                       max_logprobs=args.max_logprobs, admit_min=args.admit_min, admit_max_wait=args.admit_max_wait)


def main(argv: Optional[List[str]] = None, make=make_server, on_ready=None) -> int:
    """on_ready(httpd, server, stop_event): called once the port is bound (tests and embedding callers; a real deployment
    stops the server with SIGTERM / SIGINT, as the pipeline does: /root/reference/karanta/pipeline.py:745-751)."""
    from .serving import serve_http

    args = parse_args(argv)
    log = lambda *a: print(*a, file=sys.stderr, flush=True)   # the pipeline reads the server's stderr and stdout alike
    if args.ignored:
        log(f"ignoring vLLM arguments with no meaning for this engine: {' '.join(args.ignored)}")
    srv = make(args, log=log)
    httpd = serve_http(srv, port=args.port, host=args.host)
    stop = threading.Event()
    for sig in (signal.SIGTERM, signal.SIGINT):
        try:
            signal.signal(sig, lambda *_: stop.set())
        except ValueError:      # not the main thread (tests)
            pass
    if on_ready is not None:
        on_ready(httpd, srv, stop)
    stop.wait()
    httpd.shutdown()
    srv.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
