"""Engine vs oracle on a CHECKPOINT DIRECTORY: the parity check a maintainer with real weights runs.

    python -m oracle.checkpoint_parity MODEL_DIR [--page scan.png] [--pages 2] [--steps 16] [--policy fp32|bf16]
                                       [--max-pixels N] [--prompt "..."] [--json out.json]

TEST INFRASTRUCTURE (it lives under oracle/ because it runs the oracle: no module of karanta_ocr_amd/ imports it, and
tests/test_layout.py enforces that).  The call sequence is the reference's own direct-inference script,
/root/reference/karanta/training/test_trained_model.py:76-99: chat template -> processor -> generate(do_sample=False) ->
batch_decode; here twice over the same inputs —

  * the MI355X engine loaded exactly as the server loads it (weights.load_checkpoint, serving.HFTokenizer, the checkpoint's own
    chat template and preprocessor_config.json), and
  * the build's CPU oracle (oracle/qwen2vl_oracle.py, by default the fp32 policy = what Hugging Face computes on the CPU) at FULL
    depth on the same weights,

— and prints, per page: the prompt length, the logit error at every step as a fraction of the logit range (the engine run
is TEACHER-FORCED with the oracle's tokens so that every step compares), argmax equality at every step and at the decisive ones
(oracle top-2 margin above twice the tolerance), and the two DECODED STRINGS of the free-running greedy generations — "greedy
text identical to the CPU reference" (BASELINE.json north_star) at the text level.  Exit status 0 when every page is within
the tolerance of oracle/tolerances.py and equal at every decisive step.

Pages: --page FILE (any image PIL reads; repeatable) and / or --pages N synthetic scans (image_processing.synthetic_page, the
bench's generator).  A 7B checkpoint at full depth needs ~45 GB of host memory for the oracle's fp32 weights and a few minutes of
CPU time per page; --vit-blocks / --layers truncate BOTH sides for a quick look."""
from __future__ import annotations

import argparse
import dataclasses
import json
import os
import sys
import time
from typing import Dict, List, Optional

import numpy as np

from oracle import qwen2vl_oracle as O
from oracle.tolerances import LOGIT_TOL_REL, LOGIT_TOL_REL_FP32, TOKEN_MARGIN_FACTOR

REFERENCE_PROMPT = ("Below is the image of one page of a PDF document. Just return the plain text representation of this document "
                    "as if you were reading it naturally.")   # the shape of configs/prompts/open_ai_data_generation.yaml:12-20


def compare_page(eng, cfg, weights: Dict[str, np.ndarray], parsed, steps: int, policy: str, tol_rel: Optional[float] = None) -> dict:
    """One page: the oracle's greedy run (`steps` tokens, logits at every step) against a teacher-forced engine run and a free
    engine run.  `parsed`: serving.ParsedRequest with host pixel_values (ChatFrontend(device_images=False))."""
    from karanta_ocr_amd.engine import PageRequest

    tol_rel = tol_rel if tol_rel is not None else (LOGIT_TOL_REL_FP32 if policy == "fp32" else LOGIT_TOL_REL)
    t0 = time.perf_counter()
    o_tok, o_log = O.generate_greedy(cfg, weights, parsed.input_ids[None], parsed.pixel_values, parsed.grids, steps, policy=policy,
                                     ignore_eos=True, return_logits=True)
    t_oracle = time.perf_counter() - t0
    o_tok, o_log = np.asarray(o_tok[0]), np.asarray(o_log[0])
    page = PageRequest(parsed.input_ids, parsed.pixel_values, list(parsed.grids))
    forced = eng.generate([page], steps, ignore_eos=True, return_logits=True, force_tokens=o_tok[None, :steps - 1])
    free = eng.generate([page], steps)                       # EOS-aware, as a request is served
    g_log, g_tok = np.asarray(forced.logits[0]), np.asarray(forced.tokens[0])
    rng_ = float(np.abs(o_log[0]).max())
    tol = tol_rel * rng_
    errs = [float(np.abs(g_log[i] - o_log[i]).max()) for i in range(steps)]
    part = np.partition(o_log, -2, axis=-1)
    margins = part[:, -1] - part[:, -2]
    decisive = [i for i in range(steps) if margins[i] > TOKEN_MARGIN_FACTOR * tol]
    equal = [int(g_tok[i]) == int(o_tok[i]) for i in range(steps)]
    eos = set(cfg.eos_token_ids)
    o_free = list(o_tok[: next((i + 1 for i, t in enumerate(o_tok) if int(t) in eos), steps)])
    return {
        "prompt_tokens": int(len(parsed.input_ids)), "image_tokens": int(sum(g[0] * g[1] * g[2] for g in parsed.grids) // 4),
        "policy": policy, "steps": steps, "logit_range": round(rng_, 4), "tol_rel": tol_rel,
        "rel_err_per_step": [round(e / rng_, 5) for e in errs], "max_rel_err": round(max(errs) / rng_, 5),
        "argmax_equal": f"{sum(equal)}/{steps}", "decisive_steps": len(decisive),
        "decisive_argmax_equal": f"{sum(equal[i] for i in decisive)}/{len(decisive)}", "min_margin_rel": round(float(margins.min()) / rng_, 5),
        "oracle_tokens": [int(t) for t in o_free], "engine_tokens": [int(t) for t in free.tokens[0]],
        "engine_finish_reason": free.finish_reasons[0], "oracle_seconds": round(t_oracle, 1),
        "pass": bool(max(errs) < tol and all(equal[i] for i in decisive)),
    }


def run(model_dir: str, page_files: List[str], n_synthetic: int, steps: int, policy: str, max_pixels: Optional[int], prompt: str,
        vit_blocks: Optional[int] = None, layers: Optional[int] = None, device: str = "cuda:0", log=print) -> dict:
    from PIL import Image

    from karanta_ocr_amd import cli
    from karanta_ocr_amd import image_processing as IP
    from karanta_ocr_amd import serving as S
    from karanta_ocr_amd.engine import Engine
    from karanta_ocr_amd.weights import as_f32, load_checkpoint

    t0 = time.perf_counter()
    cfg, tensors = load_checkpoint(model_dir)
    if vit_blocks or layers:
        cfg = dataclasses.replace(cfg, vision=dataclasses.replace(cfg.vision, depth=vit_blocks or cfg.vision.depth),
                                  text=dataclasses.replace(cfg.text, num_layers=layers or cfg.text.num_layers))
    fp8 = cli._checkpoint_is_fp8(model_dir)
    ck_min, ck_max = cli.preprocessor_pixels(model_dir)
    min_px, max_px = ck_min or IP.MIN_PIXELS, max_pixels or ck_max or IP.MAX_PIXELS_CLASS_DEFAULT
    tok = S.HFTokenizer(os.path.join(model_dir, "tokenizer.json"), cfg)
    front = S.ChatFrontend(cfg, tok, min_pixels=min_px, max_pixels=max_px, chat_template=S.load_chat_template(model_dir))
    log(f"{model_dir}: {cfg.vision.depth} ViT blocks, {cfg.text.num_layers} layers, hidden {cfg.text.hidden_size}, "
        f"{'fp8' if fp8 else 'bf16'} weights, max_pixels {max_px}; loaded in {time.perf_counter() - t0:.1f}s")
    images = [np.asarray(Image.open(f).convert("RGB")) for f in page_files] + [IP.synthetic_page(i, 1024, 1024) for i in range(n_synthetic)]
    names = list(page_files) + [f"synthetic_page({i})" for i in range(n_synthetic)]
    reqs = [front.parse({"model": "parity", "max_tokens": steps, "temperature": 0.0,
                         "messages": [{"role": "user", "content": [{"type": "text", "text": prompt},
                                                                   {"type": "image_url", "image_url": {"url": IP.encode_png_data_url(im)}}]}]})
            for im in images]
    eng = Engine(cfg, device=device, max_batch=1, s_max=(max(len(r.input_ids) for r in reqs) + steps + 127) // 64 * 64,
                 max_patches=max(sum(g[1] * g[2] for g in r.grids) for r in reqs) + 64, max_prompt_tokens=max(len(r.input_ids) for r in reqs),
                 weight_dtype="fp8" if fp8 else "bf16")
    eng.load_weights(tensors)
    if fp8:      # the oracle runs on the values the engine's codes + row scales stand for
        from karanta_ocr_amd.weights import fp8_dequantized_weights
        tensors = fp8_dequantized_weights(tensors, cfg)
    # the oracle's parameters as resident fp32 arrays (bf16 bit patterns would be re-expanded on every access)
    weights = {k: as_f32(v) for k, v in tensors.items()}
    del tensors
    out = {"model_dir": model_dir, "config": cfg.name, "weights": "fp8" if fp8 else "bf16", "pages": []}
    try:
        for name, r in zip(names, reqs):
            res = compare_page(eng, cfg, weights, r, steps, policy)
            res["page"] = name
            res["oracle_text"] = tok.decode([t for t in res["oracle_tokens"] if t not in cfg.eos_token_ids])
            res["engine_text"] = tok.decode([t for t in res["engine_tokens"] if t not in cfg.eos_token_ids])
            res["text_equal"] = res["oracle_text"] == res["engine_text"]
            out["pages"].append(res)
            log(f"{name}: P={res['prompt_tokens']} max logit error {100 * res['max_rel_err']:.2f} % of the range (tolerance "
                f"{100 * res['tol_rel']:.0f} %), argmax {res['argmax_equal']}, decisive {res['decisive_argmax_equal']}, "
                f"text equal: {res['text_equal']}\n  oracle: {res['oracle_text']!r}\n  engine: {res['engine_text']!r}")
    finally:
        eng.close()
    out["pass"] = all(p["pass"] for p in out["pages"])
    return out


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("model_dir")
    ap.add_argument("--page", action="append", default=[], help="an image file (repeatable)")
    ap.add_argument("--pages", type=int, default=None, help="synthetic 1024x1024 scans (default 1 when no --page is given)")
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--policy", default="fp32", choices=("fp32", "bf16"))
    ap.add_argument("--max-pixels", type=int, default=None, help="override the checkpoint's preprocessor_config.json")
    ap.add_argument("--prompt", default=REFERENCE_PROMPT)
    ap.add_argument("--vit-blocks", type=int, default=None, help="truncate the vision tower (both sides)")
    ap.add_argument("--layers", type=int, default=None, help="truncate the decoder (both sides)")
    ap.add_argument("--json", default=None, help="write the full report here")
    a = ap.parse_args(argv)
    n_syn = a.pages if a.pages is not None else (0 if a.page else 1)
    rep = run(a.model_dir, a.page, n_syn, a.steps, a.policy, a.max_pixels, a.prompt, a.vit_blocks, a.layers,
              log=lambda *x: print(*x, file=sys.stderr, flush=True))
    if a.json:
        with open(a.json, "w") as f:
            json.dump(rep, f, indent=1)
    print(json.dumps({"pass": rep["pass"], "pages": [{k: p[k] for k in ("page", "max_rel_err", "argmax_equal", "decisive_argmax_equal",
                                                                         "text_equal")} for p in rep["pages"]]}))
    return 0 if rep["pass"] else 1


if __name__ == "__main__":
    raise SystemExit(main())
