"""Host-side weight store for the Qwen2-VL hot path.

Names follow the Hugging Face state-dict (transformers-5 layout,
``model.visual.*`` / ``model.language_model.*`` / ``lm_head.weight``) so a real
checkpoint — the thing the reference points ``vllm serve`` at,
/root/reference/karanta/pipeline.py:707-742 — loads without a mapping table; the
transformers-4 hub layout (``visual.*`` / ``model.layers.*``) is renamed on load.

Everything here is numpy on the host.  Packing for the device lives in engine.py.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Iterable, Optional

import numpy as np

from .config import ModelConfig, from_hf_config_dict

# ----------------------------------------------------------------------------- bf16


def bf16_round(x: np.ndarray) -> np.ndarray:
    """Round fp32 → nearest-even bf16, returned as fp32 (NaN preserved)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    out = ((u + r) & np.uint32(0xFFFF0000)).view(np.float32)
    return np.where(np.isnan(x), x, out)


def to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """fp32 → bf16 bit pattern (uint16), round-to-nearest-even."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    return ((u + r) >> 16).astype(np.uint16)


def from_bf16_bits(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << 16).view(np.float32)


# ----------------------------------------------------------------------------- names


def weight_shapes(cfg: ModelConfig) -> Dict[str, tuple]:
    """Every parameter of the model, HF name → shape."""
    v, t = cfg.vision, cfg.text
    s: Dict[str, tuple] = {}
    s["model.visual.patch_embed.proj.weight"] = (
        v.embed_dim, v.in_channels, v.temporal_patch_size, v.patch_size, v.patch_size)
    v25 = v.variant == "qwen2_5"   # RMSNorm (no bias), biased SwiGLU MLP (TF25: modeling_qwen2_5_vl.py:85-97, :294-300)
    for i in range(v.depth):
        p = f"model.visual.blocks.{i}."
        s[p + "norm1.weight"] = (v.embed_dim,)
        if not v25:
            s[p + "norm1.bias"] = (v.embed_dim,)
        s[p + "norm2.weight"] = (v.embed_dim,)
        if not v25:
            s[p + "norm2.bias"] = (v.embed_dim,)
        s[p + "attn.qkv.weight"] = (3 * v.embed_dim, v.embed_dim)
        s[p + "attn.qkv.bias"] = (3 * v.embed_dim,)
        s[p + "attn.proj.weight"] = (v.embed_dim, v.embed_dim)
        s[p + "attn.proj.bias"] = (v.embed_dim,)
        if v25:
            for n in ("gate_proj", "up_proj"):
                s[p + f"mlp.{n}.weight"] = (v.mlp_dim, v.embed_dim)
                s[p + f"mlp.{n}.bias"] = (v.mlp_dim,)
            s[p + "mlp.down_proj.weight"] = (v.embed_dim, v.mlp_dim)
            s[p + "mlp.down_proj.bias"] = (v.embed_dim,)
        else:
            s[p + "mlp.fc1.weight"] = (v.mlp_dim, v.embed_dim)
            s[p + "mlp.fc1.bias"] = (v.mlp_dim,)
            s[p + "mlp.fc2.weight"] = (v.embed_dim, v.mlp_dim)
            s[p + "mlp.fc2.bias"] = (v.embed_dim,)
    s["model.visual.merger.ln_q.weight"] = (v.embed_dim,)
    if not v25:
        s["model.visual.merger.ln_q.bias"] = (v.embed_dim,)
    s["model.visual.merger.mlp.0.weight"] = (v.merge_dim, v.merge_dim)
    s["model.visual.merger.mlp.0.bias"] = (v.merge_dim,)
    s["model.visual.merger.mlp.2.weight"] = (v.hidden_size, v.merge_dim)
    s["model.visual.merger.mlp.2.bias"] = (v.hidden_size,)
    s["model.language_model.embed_tokens.weight"] = (t.vocab_size, t.hidden_size)
    for i in range(t.num_layers):
        p = f"model.language_model.layers.{i}."
        s[p + "self_attn.q_proj.weight"] = (t.q_dim, t.hidden_size)
        s[p + "self_attn.q_proj.bias"] = (t.q_dim,)
        s[p + "self_attn.k_proj.weight"] = (t.kv_dim, t.hidden_size)
        s[p + "self_attn.k_proj.bias"] = (t.kv_dim,)
        s[p + "self_attn.v_proj.weight"] = (t.kv_dim, t.hidden_size)
        s[p + "self_attn.v_proj.bias"] = (t.kv_dim,)
        s[p + "self_attn.o_proj.weight"] = (t.hidden_size, t.q_dim)
        s[p + "mlp.gate_proj.weight"] = (t.intermediate_size, t.hidden_size)
        s[p + "mlp.up_proj.weight"] = (t.intermediate_size, t.hidden_size)
        s[p + "mlp.down_proj.weight"] = (t.hidden_size, t.intermediate_size)
        s[p + "input_layernorm.weight"] = (t.hidden_size,)
        s[p + "post_attention_layernorm.weight"] = (t.hidden_size,)
    s["model.language_model.norm.weight"] = (t.hidden_size,)
    if not t.tie_word_embeddings:
        s["lm_head.weight"] = (t.vocab_size, t.hidden_size)
    return s


def _random_tensor(shapes, seed: int, idx: int, as_bits: bool) -> np.ndarray:
    name, shape = shapes[idx]
    rng = np.random.default_rng([seed, idx])
    n = int(np.prod(shape))
    if name.endswith("bias"):
        w = rng.standard_normal(n, dtype=np.float32) * np.float32(0.02)
    elif len(shape) == 1:
        w = np.float32(1.0) + rng.standard_normal(n, dtype=np.float32) * np.float32(0.1)
    elif "embed_tokens" in name:
        w = rng.standard_normal(n, dtype=np.float32) * np.float32(0.5)
    else:
        fan_in = int(np.prod(shape[1:]))
        w = rng.standard_normal(n, dtype=np.float32) * np.float32(1.0 / np.sqrt(fan_in))
    w = w.reshape(shape)
    return to_bf16_bits(w) if as_bits else bf16_round(w)


def random_weights(cfg: ModelConfig, seed: int = 0, as_bits: bool = False, threads: Optional[int] = None,
                   only: Optional[Iterable[str]] = None) -> Dict[str, np.ndarray]:
    """Seeded random-init weights, every value exactly representable in bf16.

    One ``np.random.default_rng([seed, index])`` stream per tensor (PCG64: stable across
    machines and numpy versions), so the golden script, the oracle, the tests and the GPU
    engine all see bit-identical parameters without shipping them.

    Matrices ~ N(0, 1/sqrt(fan_in)); norm weights 1 + N(0, 0.1); biases N(0, 0.02).
    With ``as_bits`` the tensors are returned as bf16 bit patterns (uint16) — half
    the host memory for the full-size bench models.  The tensors are independent streams, so they
    are drawn on a few host threads (numpy's generators release the GIL): same bits, a fraction of
    the wall time for the 2B / 7B bench models.  ``only``: just these tensors (the same bits they have in the full set).
    """
    shapes = list(weight_shapes(cfg).items())
    if only is not None:
        want = set(only)
        return {shapes[i][0]: _random_tensor(shapes, seed, i, as_bits) for i in range(len(shapes)) if shapes[i][0] in want}

    def one(idx: int) -> np.ndarray:
        return _random_tensor(shapes, seed, idx, as_bits)

    total = sum(int(np.prod(sh)) for _, sh in shapes)
    if threads is None:
        threads = min(8, os.cpu_count() or 1) if total > (1 << 26) else 1
    if threads <= 1:
        return {shapes[i][0]: one(i) for i in range(len(shapes))}
    from concurrent.futures import ThreadPoolExecutor
    # largest first, so the big embedding / lm_head tensors do not end up alone at the tail
    order = sorted(range(len(shapes)), key=lambda i: -int(np.prod(shapes[i][1])))
    with ThreadPoolExecutor(max_workers=threads) as ex:
        done = dict(zip(order, ex.map(one, order)))
    return {shapes[i][0]: done[i] for i in range(len(shapes))}


# ----------------------------------------------------------------------------- checkpoints

_V4_RENAMES = (
    ("visual.", "model.visual."),
    ("model.embed_tokens.", "model.language_model.embed_tokens."),
    ("model.layers.", "model.language_model.layers."),
    ("model.norm.", "model.language_model.norm."),
)


def canonical_name(name: str) -> str:
    """Map a transformers-4 hub name to the transformers-5 name used internally."""
    if name.startswith("model.visual.") or name.startswith("model.language_model.") or name == "lm_head.weight":
        return name
    for old, new in _V4_RENAMES:
        if name.startswith(old):
            return new + name[len(old):]
    return name


def load_config(model_dir: str) -> ModelConfig:
    """``config.json`` of a local checkpoint directory -> :class:`ModelConfig` (a few KB: every rank of a multi-GPU
    launch reads it, only rank 0 reads the tensors — launch.py)."""
    with open(os.path.join(model_dir, "config.json")) as f:
        return from_hf_config_dict(json.load(f), name=os.path.basename(model_dir.rstrip("/")))


def load_checkpoint(model_dir: str) -> tuple:
    """Load ``config.json`` + ``*.safetensors`` from a local directory.

    Returns ``(ModelConfig, {name: np.ndarray})``.  bf16 tensors come back as uint16
    bit patterns (safetensors has no numpy bf16); callers use :func:`from_bf16_bits`.
    """
    from safetensors import safe_open  # local import: only needed with real weights

    cfg = load_config(model_dir)
    import torch

    tensors: Dict[str, np.ndarray] = {}
    files = sorted(f for f in os.listdir(model_dir) if f.endswith(".safetensors"))
    if not files:
        raise FileNotFoundError(f"no *.safetensors under {model_dir}")
    raw: Dict[str, np.ndarray] = {}
    for fn in files:
        with safe_open(os.path.join(model_dir, fn), framework="pt") as sf:
            for k in sf.keys():
                t = sf.get_tensor(k)
                if t.dtype == torch.bfloat16:
                    arr = t.view(torch.int16).numpy().view(np.uint16)
                else:
                    arr = t.float().numpy()          # float8_e4m3fn codes come back as their (unscaled) values
                raw[k] = arr
    # fp8 checkpoints (compressed-tensors, e.g. the reference's allenai/olmOCR-7B-0725-FP8, karanta/constants.py:23):
    # `<linear>.weight` holds e4m3 values and `<linear>.weight_scale` one scale per output channel ([N, 1]) or per
    # tensor.  They are dequantised here; Engine(weight_dtype="fp8") re-derives codes + row scales (identical codes when
    # the checkpoint used the full e4m3 range per row).  Activation scales of static schemes are not used.
    for k in [k for k in raw if k.endswith(".weight_scale") or k.endswith(".weight_scale_inv")]:
        base = k[: k.rindex(".")] + ".weight"
        sc = as_f32(raw.pop(k)).reshape(-1)
        if base not in raw:
            continue
        w = as_f32(raw[base])
        if sc.size not in (1, w.shape[0]):
            raise ValueError(f"{k}: {sc.size} scales for a weight of shape {w.shape} (per-tensor or per-output-channel only)")
        raw[base] = w * (sc.reshape(-1, 1) if sc.size > 1 else sc[0])
    for k in [k for k in raw if k.endswith(".input_scale")]:
        raw.pop(k)
    for k, arr in raw.items():
        tensors[canonical_name(k)] = arr
    return cfg, tensors


def as_f32(w: np.ndarray) -> np.ndarray:
    return from_bf16_bits(w) if w.dtype == np.uint16 else np.asarray(w, dtype=np.float32)


def pack_w16x64(w: np.ndarray) -> np.ndarray:
    """Row-major ``[N, K]`` -> the decode layout ``[N/16][K/32][4][16][8]`` (flattened back to an
    ``[N, K]``-shaped array for bookkeeping).

    One (16-row, 32-column) block is exactly one ``v_mfma_f32_16x16x32_bf16`` A/B fragment set in
    lane order: lane ``l = 16*g + r`` holds row ``r``, columns ``8g .. 8g+7``.  So a wave fetches a
    block with ONE 16-byte-per-lane load covering 1 KiB of contiguous memory (8 full 128-byte
    lines), and a 16-row group is one linear stream along K.  kr_linear_decode streams it;
    kr_gemm_bf16 reads the same copy with ``w_packed=1``."""
    n, k = w.shape
    if n % 16 or k % 64:
        raise ValueError(f"pack_w16x64: shape {w.shape} is not a multiple of (16, 64)")
    return np.ascontiguousarray(w.reshape(n // 16, 16, k // 32, 4, 8).transpose(0, 2, 3, 1, 4)).reshape(n, k)


def pack_rows32(x: np.ndarray) -> np.ndarray:
    """Row-major activations ``[M <= 32, K]`` -> the packed layout of 17..32-row decode batches
    ``[K/64][2][2][4][16][8]`` (include/karanta_hip.h, kr_pack_rows32; rows >= M are zero): element (row b, column k)
    sits at chunk k // 64, column tile b // 16, k-step (k // 32) % 2, lane group (k // 8) % 4, lane row b % 16, k % 8 —
    the operand of one ``v_mfma_f32_16x16x32_bf16`` is 1 KiB of contiguous memory in lane order.  Returned flat."""
    m, k = x.shape
    if m > 32 or k % 64:
        raise ValueError(f"pack_rows32: shape {x.shape} (at most 32 rows, K % 64 == 0)")
    full = np.zeros((32, k), x.dtype)
    full[:m] = x
    return np.ascontiguousarray(full.reshape(2, 16, k // 64, 2, 4, 8).transpose(2, 0, 3, 4, 1, 5)).reshape(-1)


def unpack_rows32(xp: np.ndarray, k: int) -> np.ndarray:
    """Inverse of :func:`pack_rows32`: flat packed buffer -> ``[32, K]``."""
    return np.ascontiguousarray(np.asarray(xp).reshape(k // 64, 2, 2, 4, 16, 8).transpose(1, 4, 0, 2, 3, 5)).reshape(32, k)


# ----------------------------------------------------------------------------- fp8 (OCP e4m3fn) weight-only quantisation
# BASELINE.json config 5: decoder Linears in fp8 with one fp32 scale per output channel, activations / lm_head / ViT in
# bf16.  gfx950 converts OCP e4m3fn (4 exponent bits, bias 7, 3 mantissa bits, max 448, no infinities) in hardware
# (v_cvt_scalef32_pk_bf16_fp8); every e4m3 value is exact in bf16.
def _e4m3_table() -> np.ndarray:
    codes = np.arange(256, dtype=np.int64)
    sign = np.where(codes & 0x80, -1.0, 1.0)
    e, m = (codes >> 3) & 0xF, codes & 0x7
    val = np.where(e == 0, m / 8.0 * 2.0 ** -6, (1.0 + m / 8.0) * 2.0 ** (e - 7.0))
    val = sign * val
    val[(codes & 0x7F) == 0x7F] = np.nan          # the two NaN encodings
    return val.astype(np.float32)


E4M3 = _e4m3_table()
E4M3_MAX = 448.0


def fp8_e4m3_to_f32(q: np.ndarray) -> np.ndarray:
    return E4M3[np.asarray(q, np.uint8)]


def f32_to_fp8_e4m3_fast(x: np.ndarray) -> np.ndarray:
    """`f32_to_fp8_e4m3` by integer arithmetic on the float32 bit patterns (a dozen vectorised passes instead of a
    binary search per element: the 6.5 G weights of a 7B decoder quantise in minutes, not hours)."""
    x = np.ascontiguousarray(x, np.float32)
    bits = x.view(np.uint32)
    sign = ((bits >> np.uint32(24)) & np.uint32(0x80)).astype(np.uint8)
    u = bits & np.uint32(0x7FFFFFFF)
    # normal e4m3 range (|x| >= 2^-6): rebias the exponent 127 -> 7, round the 23-bit mantissa to 3 bits, ties to even;
    # a mantissa carry walks into the exponent by itself
    r = u - np.uint32((127 - 7) << 23)
    r = (r + np.uint32(0x7FFFF) + ((r >> np.uint32(20)) & np.uint32(1))) >> np.uint32(20)
    # subnormal range (|x| < 2^-6): the code is round_half_even(|x| * 2^9), 0 .. 8 (8 = the smallest normal)
    sub = np.rint(np.minimum(np.abs(x), np.float32(1.0)) * np.float32(512.0)).astype(np.uint32)
    is_sub = u < np.uint32((127 - 6) << 23)
    code = np.where(is_sub, sub, r)
    code = np.minimum(code, np.uint32(0x7E)).astype(np.uint8)        # saturate (also tames the rebias wrap of tiny values)
    return code | sign


def f32_to_fp8_e4m3(x: np.ndarray) -> np.ndarray:
    """Round to nearest e4m3fn code, ties to the even mantissa, saturating at +-448 (finite input)."""
    x = np.asarray(x, np.float32)
    mag = np.minimum(np.abs(x).astype(np.float64), E4M3_MAX)
    pos = E4M3[:0x7F].astype(np.float64)           # codes 0x00..0x7E: increasing, 0 .. 448
    hi = np.clip(np.searchsorted(pos, mag, side="left"), 0, 0x7E)
    lo = np.clip(hi - 1, 0, 0x7E)
    d_lo, d_hi = mag - pos[lo], pos[hi] - mag
    pick_hi = (d_hi < d_lo) | ((d_hi == d_lo) & ((hi & 1) == 0))
    code = np.where(pick_hi, hi, lo).astype(np.uint8)
    return (code | np.where(np.signbit(x), 0x80, 0).astype(np.uint8)).astype(np.uint8)


def quantize_fp8_rows(w: np.ndarray) -> tuple:
    """[N, K] fp32 -> (codes uint8 [N, K], scale fp32 [N]) with w ~ scale[:, None] * e4m3(codes): per output channel,
    scale = max|row| / 448 (1 for an all-zero row)."""
    w = np.asarray(w, np.float32)
    amax = np.abs(w).max(axis=1)
    scale = np.where(amax > 0, amax / np.float32(E4M3_MAX), np.float32(1.0)).astype(np.float32)
    return f32_to_fp8_e4m3_fast(w / scale[:, None]), scale


def pack_w16x64_fp8(q: np.ndarray) -> np.ndarray:
    """Row-major uint8 ``[N, K]`` -> ``[N/16][K/64][4][16][16]``: one (16-row, 64-column) block is 1 KiB in lane order,
    lane ``l = 16*g + r`` holding row ``r``, columns ``16g .. 16g+15`` — one 16-byte load per lane feeds two
    ``v_mfma_f32_16x16x32_bf16`` k-steps (columns 16g..16g+7, then 16g+8..16g+15; x is read in the same order)."""
    n, k = q.shape
    if n % 16 or k % 64:
        raise ValueError(f"pack_w16x64_fp8: shape {q.shape} is not a multiple of (16, 64)")
    return np.ascontiguousarray(q.reshape(n // 16, 16, k // 64, 4, 16).transpose(0, 2, 3, 1, 4)).reshape(n, k)


def unpack_w16x64_fp8(p: np.ndarray) -> np.ndarray:
    n, k = p.shape
    return np.ascontiguousarray(p.reshape(n // 16, k // 64, 4, 16, 16).transpose(0, 3, 1, 2, 4)).reshape(n, k)


FP8_LINEARS = ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj",
               "mlp.down_proj")


def fp8_dequantized_weights(weights: Dict[str, np.ndarray], cfg: ModelConfig) -> Dict[str, np.ndarray]:
    """The state dict with every decoder Linear replaced by scale * e4m3(codes) (fp32): the model the fp8 engine
    computes with, for the oracle and for loading.  Biases, norms, embeddings, lm_head and the ViT are untouched."""
    out = dict(weights)
    for i in range(cfg.text.num_layers):
        for n in FP8_LINEARS:
            key = f"model.language_model.layers.{i}.{n}.weight"
            q, s = quantize_fp8_rows(as_f32(weights[key]))
            out[key] = (fp8_e4m3_to_f32(q) * s[:, None]).astype(np.float32)
    return out


def unpack_w16x64(p: np.ndarray) -> np.ndarray:
    n, k = p.shape
    return np.ascontiguousarray(p.reshape(n // 16, k // 32, 4, 16, 8).transpose(0, 3, 1, 2, 4)).reshape(n, k)
