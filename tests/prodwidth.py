"""Helpers of the production-width parity tests (not a test module).

The bench model (Qwen2-VL-2B / 7B widths, full vocabulary) at TRUNCATED DEPTH is the largest thing the oracle finishes
in seconds, and it is what makes the engine pick its production kernel instantiations: dec_wide_kernel<.,24> /
dec_narrow_kernel<..,24,2,..> (K = 1536) or the K = 3584 ones, the deferred split-K chain through several layers,
256x256 GEMM tiles at M = 4900+, attn_varlen_kernel<80> over one 4900-token segment (77 KV tiles).
"""
from __future__ import annotations

import dataclasses
from typing import Optional, Sequence, Tuple

import numpy as np


def truncated_config(name: str, vit_depth: int, layers: int, untie: bool = True):
    """CONFIGS[name] with full widths / heads / vocabulary and fewer ViT blocks and decoder layers.  `untie`: a random
    TIED head makes a random-init model echo its last input token with a margin of hundreds of logits (logit_i =
    h . E_i with h ~ E_token) — a model that emits the same token under almost any bug; the untied variant runs the
    same kernels (the engine packs its own lm_head copy either way) on logits that depend on the whole computation."""
    from karanta_ocr_amd.config import CONFIGS
    base = CONFIGS[name]
    text = dataclasses.replace(base.text, num_layers=layers, tie_word_embeddings=(base.text.tie_word_embeddings and not untie))
    return dataclasses.replace(base, name=f"{name}-w-v{vit_depth}-l{layers}", vision=dataclasses.replace(base.vision, depth=vit_depth),
                               text=text)


def page_inputs(cfg, index: int, h: int, w: int, max_pixels: int, n_pre: int, n_post: int, seed: int):
    """(input_ids, pixel_values, grid) of one synthetic scan in the chat-message order (text, image, text)."""
    from karanta_ocr_amd import image_processing as IP
    pv, grid = IP.image_to_patches(IP.synthetic_page(index, h, w), max_pixels=max_pixels)
    T = grid[1] * grid[2] // 4
    rng = np.random.default_rng(seed)
    hi = min(150000, cfg.text.vocab_size - 1)
    ids = np.concatenate([rng.integers(0, hi, n_pre), [cfg.vision_start_token_id], [cfg.image_token_id] * T,
                          [cfg.vision_end_token_id], rng.integers(0, hi, n_post)]).astype(np.int64)
    return ids, pv, grid


def margins(ref_logits: np.ndarray) -> np.ndarray:
    """Top-1 minus top-2 of every step's oracle logits [steps, V]."""
    part = np.partition(ref_logits, -2, axis=-1)
    return part[:, -1] - part[:, -2]


def compare_generation(got_tokens: Sequence[int], got_logits: Optional[np.ndarray], ref_tokens: Sequence[int],
                       ref_logits: np.ndarray, tol: float, what: str = "") -> Tuple[int, int]:
    """Greedy parity under a stated logit tolerance `tol`.

    Walks the steps while the two token sequences agree (after the first disagreement the inputs differ and nothing
    can be compared).  At every such step the full logit vector must agree within 1.5 * tol (when the engine's logits
    are given).  A step is DECISIVE when the oracle's top-2 margin exceeds 2 * tol: there the tokens must be equal
    (an argmax flip would be an error larger than the tolerance).  At a near-tie the tokens may differ — that ends the
    walk — or agree, in which case the walk goes on.  Returns (decisive steps compared, steps walked)."""
    m = margins(ref_logits)
    decisive = walked = 0
    for i in range(min(len(got_tokens), len(ref_tokens))):
        if got_logits is not None:
            err = float(np.abs(got_logits[i] - ref_logits[i]).max())
            assert err < 1.5 * tol, f"{what} step {i}: logits off by {err:.4f} (1.5 tol = {1.5 * tol:.4f})"
        if m[i] > 2 * tol:
            assert int(got_tokens[i]) == int(ref_tokens[i]), \
                f"{what} step {i}: engine {int(got_tokens[i])} vs oracle {int(ref_tokens[i])} at margin {m[i]:.3f} > 2 tol {2 * tol:.3f}"
            decisive += 1
        elif int(got_tokens[i]) != int(ref_tokens[i]):
            break                         # a legitimate flip at a near-tie: the sequences part here
        walked += 1
    return decisive, walked


def compare_teacher_forced(got_tokens: Sequence[int], got_logits: np.ndarray, ref_tokens: Sequence[int], ref_logits: np.ndarray,
                           tol: float, what: str = "") -> int:
    """Parity of a TEACHER-FORCED engine run (Engine.generate(force_tokens = the oracle's tokens)): the engine saw the
    oracle's prefix at every step, so every step compares — the full logit vector within 1.5 * tol, and the engine's
    argmax equal to the oracle's wherever the oracle's top-2 margin exceeds 2 * tol.  Returns the decisive-step count."""
    m = margins(ref_logits)
    decisive = 0
    for i in range(len(ref_tokens)):
        err = float(np.abs(got_logits[i] - ref_logits[i]).max())
        assert err < 1.5 * tol, f"{what} step {i}: logits off by {err:.4f} (1.5 tol = {1.5 * tol:.4f})"
        if m[i] > 2 * tol:
            assert int(got_tokens[i]) == int(ref_tokens[i]), \
                f"{what} step {i}: engine {int(got_tokens[i])} vs oracle {int(ref_tokens[i])} at margin {m[i]:.3f} > 2 tol {2 * tol:.3f}"
            decisive += 1
    return decisive
