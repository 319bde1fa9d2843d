"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product package.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; ``karanta_ocr_amd`` must not (tests/test_layout.py enforces it).

What it restates
----------------
The reference (karanta-ocr) holds *no* model arithmetic: the Qwen2-VL image-encoder +
text-decoder forward runs inside an external ``vllm serve`` process
(/root/reference/karanta/pipeline.py:707-742 spawn, :317-319 POST;
/root/reference/bulk_processing/workers/vllm_client.py:209 POST) or inside Hugging Face
``model.generate`` (/root/reference/karanta/training/test_trained_model.py:76-99).
``vllm`` is unpinned and absent; ``transformers`` is pinned 4.53.3 in the reference's
uv.lock (4.51.3 in requirements.txt:65) and present in the build container as 5.15.0.
This file restates the *published* Qwen2-VL algorithm in numpy, function by function;
each docstring names the reference call site it serves and the transformers symbol whose
semantics it follows ("TF:" = transformers/models/qwen2_vl/).

Pinning
-------
The reference has no golden vectors for this path.  The oracle is
pinned against outputs of the Hugging Face implementation run in the build container:
``tests/golden/make_golden.py`` generated ``tests/golden/*.npz`` from transformers
5.15.0 (seeded tiny configs, fp32, eager attention) and ``tests/test_oracle_golden.py``
checks every function here against them.

dtype policy
------------
``policy="fp32"`` is plain fp32 (what the HF CPU path computes).
``policy="bf16"`` rounds activations to bf16 at the points where the HIP engine stores
bf16 to HBM (kernel boundaries), keeping fp32 accumulation inside each op — the
"same dtype policy" oracle that SURVEY.md §7 asks greedy-token equality to be stated
against.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

F32 = np.float32

# =============================================================================
# small helpers
# =============================================================================


def bf16_round(x: np.ndarray) -> np.ndarray:
    """fp32 → nearest-even bf16 → fp32."""
    x = np.ascontiguousarray(x, dtype=F32)
    u = x.view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    out = ((u + r) & np.uint32(0xFFFF0000)).view(F32)
    return np.where(np.isnan(x), x, out)


class _Policy:
    """"fp32" | "bf16" | "w8a8".  "w8a8" = "bf16" plus dynamic per-token fp8 (e4m3) quantisation of the activations that
    enter the decoder's Linears during PREFILL — what the engine's W8A8 prefill computes (kr_quantize_rows_fp8 +
    kr_gemm_fp8a) and what vLLM does for the reference's default checkpoint OLMO_7B_0725_FP8
    (/root/reference/karanta/constants.py:23); the weights are whatever the caller passes (the dequantised fp8 state
    dict); decode steps keep bf16 activations, as the engine's weight-only decode kernels do."""

    def __init__(self, name: str):
        if name not in ("fp32", "bf16", "w8a8"):
            raise ValueError(name)
        self.act_fp8 = name == "w8a8"
        self.name = "bf16" if name == "w8a8" else name

    def __call__(self, x: np.ndarray) -> np.ndarray:
        return bf16_round(x) if self.name == "bf16" else np.asarray(x, dtype=F32)


def fp8_e4m3_round(x: np.ndarray) -> np.ndarray:
    """Nearest OCP e4m3fn value (ties to the even mantissa), saturating at +-448: 3 mantissa bits above 2^-6, a fixed
    step of 2^-9 below it (the subnormals).  Restated arithmetically, independently of the engine's integer version."""
    x = np.asarray(x, dtype=np.float64)
    mag = np.minimum(np.abs(x), 448.0)
    _, e = np.frexp(mag)                       # mag = m * 2^e with m in [0.5, 1): the leading bit has weight 2^(e-1)
    step = np.exp2(np.maximum(e - 1, -6) - 3.0)  # spacing of the e4m3 grid around mag
    q = np.minimum(np.rint(mag / step) * step, 448.0)   # np.rint rounds half to even: the even multiple = the even mantissa
    return np.copysign(q, x).astype(F32)


def fake_quant_rows_fp8(x: np.ndarray) -> np.ndarray:
    """Dynamic per-token fp8 quantisation and back: scale = max|row| / 448 in f32 (1 for an all-zero row),
    value = e4m3(x / scale) * scale — the A operand of the engine's W8A8 GEMM as real numbers."""
    x = np.asarray(x, dtype=F32)
    amax = np.abs(x).max(axis=-1, keepdims=True)
    scale = np.where(amax > 0, amax / F32(448.0), F32(1.0)).astype(F32)
    return (fp8_e4m3_round((x / scale).astype(F32)) * scale).astype(F32)


def _w(weights: Dict[str, np.ndarray], name: str) -> np.ndarray:
    w = weights[name]
    if w.dtype == np.uint16:  # bf16 bit pattern
        return (w.astype(np.uint32) << 16).view(F32)
    return np.asarray(w, dtype=F32)


def linear(x: np.ndarray, w: np.ndarray, b: Optional[np.ndarray] = None) -> np.ndarray:
    """``torch.nn.Linear``: x @ w.T + b, fp32 accumulate."""
    y = np.asarray(x, dtype=F32) @ np.asarray(w, dtype=F32).T  # (no copy of w: BLAS takes the transpose flag)
    if b is not None:
        y = y + np.asarray(b, dtype=F32)
    return y.astype(F32, copy=False)


def softmax_lastdim(x: np.ndarray) -> np.ndarray:
    """fp32 softmax.  Exponents are clamped at -87 (exp(-87) = 1.6e-38, the smallest normal fp32):
    terms below that would be subnormal — they change no result bit that survives the division,
    and subnormal arithmetic is ~40x slower on the host cores."""
    m = x.max(axis=-1, keepdims=True)
    e = x - m
    np.maximum(e, F32(-87.0), out=e)
    np.exp(e, out=e)
    e /= e.sum(axis=-1, keepdims=True)
    return e.astype(F32, copy=False)


def rotate_half(x: np.ndarray) -> np.ndarray:
    """TF:modeling_qwen2_vl.py:173-177."""
    h = x.shape[-1] // 2
    return np.concatenate([-x[..., h:], x[..., :h]], axis=-1)


def quick_gelu(x: np.ndarray) -> np.ndarray:
    """ViT MLP activation, ``x * sigmoid(1.702 x)`` (TF:modeling_qwen2_vl.py:293-301)."""
    x = x.astype(F32)
    return (x / (1.0 + np.exp(-1.702 * x))).astype(F32)


def gelu_erf(x: np.ndarray) -> np.ndarray:
    """Exact GELU of the PatchMerger MLP (``nn.GELU()``, TF:modeling_qwen2_vl.py:277-290)."""
    x = x.astype(np.float64)
    try:  # scipy's vectorised erf when present (same values; ~100x faster than math.erf per element)
        from scipy.special import erf
    except Exception:  # pragma: no cover
        erf = np.vectorize(math.erf, otypes=[np.float64])
    return (0.5 * x * (1.0 + erf(x / math.sqrt(2.0)))).astype(F32)


def silu(x: np.ndarray) -> np.ndarray:
    x = x.astype(F32)
    return (x / (1.0 + np.exp(-x))).astype(F32)


def layer_norm(x: np.ndarray, w: np.ndarray, b: np.ndarray, eps: float = 1e-6) -> np.ndarray:
    """``nn.LayerNorm`` with bias, eps 1e-6 (ViT blocks and merger)."""
    x = x.astype(F32)
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    return ((x - mu) / np.sqrt(var + eps) * w + b).astype(F32)


def rms_norm(x: np.ndarray, w: np.ndarray, eps: float = 1e-6, pol: Optional[_Policy] = None) -> np.ndarray:
    """``Qwen2VLRMSNorm`` (TF:modeling_qwen2_vl.py:96-110):
    ``w * (x * rsqrt(mean(x^2) + eps)).to(input_dtype)`` — the cast to the input dtype
    happens *before* the multiply by ``w``."""
    x = x.astype(F32)
    var = (x * x).mean(axis=-1, keepdims=True)
    y = x * (1.0 / np.sqrt(var + F32(eps)))
    if pol is not None:
        y = pol(y)
    return (w * y).astype(F32)


# =============================================================================
# image front end (host side of the reference: the PNG/JPEG data-URL inside
# create_vision_message, /root/reference/karanta/data/utils.py:269-297, becomes
# pixel_values + image_grid_thw inside the server)
# =============================================================================

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def smart_resize(height: int, width: int, factor: int = 28,
                 min_pixels: int = 56 * 56, max_pixels: int = 14 * 14 * 4 * 1280) -> Tuple[int, int]:
    """TF:image_processing_pil_qwen2_vl.py:57-83.  Python ``round`` (banker's) and
    float64 sqrt, exactly as there."""
    if max(height, width) / min(height, width) > 200:
        raise ValueError("absolute aspect ratio must be smaller than 200")
    h_bar = round(height / factor) * factor
    w_bar = round(width / factor) * factor
    if h_bar * w_bar > max_pixels:
        beta = math.sqrt((height * width) / max_pixels)
        h_bar = max(factor, math.floor(height / beta / factor) * factor)
        w_bar = max(factor, math.floor(width / beta / factor) * factor)
    elif h_bar * w_bar < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        h_bar = math.ceil(height * beta / factor) * factor
        w_bar = math.ceil(width * beta / factor) * factor
    return h_bar, w_bar


def patchify(image_chw: np.ndarray, patch: int = 14, merge: int = 2, temporal: int = 2) -> Tuple[np.ndarray, int, int]:
    """TF:image_processing_pil_qwen2_vl.py:152-187 — ``[C,H,W]`` fp32 → ``[gh*gw, C*T*p*p]``
    in (gh/m, gw/m, m, m, C, T, p, p) order with the single frame duplicated T times."""
    c, h, w = image_chw.shape
    gh, gw = h // patch, w // patch
    x = image_chw.astype(F32).reshape(c, gh // merge, merge, patch, gw // merge, merge, patch)
    x = x.transpose(1, 4, 2, 5, 0, 3, 6)  # gh/m, gw/m, m, m, C, p, p
    x = np.broadcast_to(x[:, :, :, :, :, None, :, :], (*x.shape[:5], temporal, patch, patch))
    return np.ascontiguousarray(x).reshape(gh * gw, c * temporal * patch * patch), gh, gw


def preprocess_image(img_rgb_u8: np.ndarray, min_pixels: int = 56 * 56, max_pixels: int = 28 * 28 * 1280,
                     patch: int = 14, merge: int = 2, temporal: int = 2) -> Tuple[np.ndarray, Tuple[int, int, int]]:
    """resize(bicubic, PIL) → ×1/255 → (x-mean)/std → patchify
    (TF:image_processing_pil_qwen2_vl.py:126-150, :226-229, :152-187).
    ``img_rgb_u8`` is HWC uint8.  Returns (pixel_values [N,1176] fp32, (1,gh,gw))."""
    from PIL import Image

    h, w = img_rgb_u8.shape[:2]
    rh, rw = smart_resize(h, w, patch * merge, min_pixels, max_pixels)
    pil = Image.fromarray(img_rgb_u8, mode="RGB")
    if (rh, rw) != (h, w):
        pil = pil.resize((rw, rh), resample=Image.BICUBIC)
    x = np.asarray(pil, dtype=np.uint8).astype(F32).transpose(2, 0, 1)  # CHW
    x = x * F32(1.0 / 255.0)
    mean = np.asarray(CLIP_MEAN, dtype=F32)[:, None, None]
    std = np.asarray(CLIP_STD, dtype=F32)[:, None, None]
    x = (x - mean) / std
    pv, gh, gw = patchify(x, patch, merge, temporal)
    return pv, (1, gh, gw)


# =============================================================================
# positions
# =============================================================================


def vision_position_ids(grid_thw: Sequence[Sequence[int]], merge: int = 2) -> np.ndarray:
    """(h, w) id of every patch in 2×2-block-major order (TF:vision_utils.py:81-127)."""
    out = []
    for t, h, w in grid_thw:
        hp, wp = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        shp = (h // merge, merge, w // merge, merge)
        hp = hp.reshape(shp).transpose(0, 2, 1, 3).reshape(-1)
        wp = wp.reshape(shp).transpose(0, 2, 1, 3).reshape(-1)
        out.append(np.tile(np.stack([hp, wp], axis=-1), (t, 1)))
    return np.concatenate(out, axis=0).astype(np.int64)


def vision_rotary_cos_sin(pos_hw: np.ndarray, head_dim: int, theta: float = 10000.0) -> Tuple[np.ndarray, np.ndarray]:
    """``VisionRotaryEmbedding(head_dim // 2)`` then ``cat(freqs, freqs)`` →
    cos/sin ``[N, head_dim]`` (TF:modeling_qwen2_vl.py:239-248, :711-713)."""
    dim = head_dim // 2
    inv = (1.0 / (float(theta) ** (np.arange(0, dim, 2, dtype=np.float64) / dim))).astype(F32)
    fr = (pos_hw[:, :, None].astype(F32) * inv[None, None, :]).reshape(pos_hw.shape[0], -1)  # [N, dim]
    emb = np.concatenate([fr, fr], axis=-1)
    return np.cos(emb).astype(F32), np.sin(emb).astype(F32)


def get_rope_index(input_ids: np.ndarray, image_grid_thw: Sequence[Sequence[int]], image_token_id: int,
                   merge: int = 2) -> Tuple[np.ndarray, np.ndarray]:
    """3-D M-RoPE position ids ``[3,B,P]`` and ``rope_delta [B]`` for unpadded prompts
    (TF:modeling_qwen2_vl.py:914-1016 with :862-912).  Text runs count up 1-D on all three
    axes; an image run gets t=start, h=start+row, w=start+col over (gh/m, gw/m) and the
    next text position is start + max(gh,gw)/m."""
    input_ids = np.asarray(input_ids)
    b, p = input_ids.shape
    pos = np.zeros((3, b, p), dtype=np.int64)
    deltas = np.zeros((b,), dtype=np.int64)
    grids = iter(image_grid_thw)
    for bi in range(b):
        ids = input_ids[bi]
        is_img = ids == image_token_id
        cur = 0
        i = 0
        cols: List[np.ndarray] = []
        while i < p:
            j = i
            while j < p and is_img[j] == is_img[i]:
                j += 1
            if not is_img[i]:
                n = j - i
                cols.append(np.tile(np.arange(n)[None, :] + cur, (3, 1)))
                cur += n
            else:
                t, gh, gw = next(grids)
                lh, lw = gh // merge, gw // merge
                if t * lh * lw != j - i:
                    raise ValueError("image token run does not match grid")
                tt, hh, ww = np.meshgrid(np.arange(t), np.arange(lh) + cur, np.arange(lw) + cur, indexing="ij")
                cols.append(np.stack([tt.reshape(-1) + cur, hh.reshape(-1), ww.reshape(-1)], axis=0))
                cur += max(gh, gw) // merge
            i = j
        allp = np.concatenate(cols, axis=1)
        pos[:, bi] = allp
        deltas[bi] = allp.max() + 1 - p
    return pos, deltas


def mrope_cos_sin(position_ids: np.ndarray, head_dim: int, theta: float,
                  mrope_section: Sequence[int]) -> Tuple[np.ndarray, np.ndarray]:
    """cos/sin ``[B,P,head_dim]`` with the M-RoPE section interleave already applied:
    ``Qwen2VLRotaryEmbedding`` (TF:modeling_qwen2_vl.py:117-169) followed by the
    channel selection of ``apply_multimodal_rotary_pos_emb`` (:180-222): channel chunks of
    sizes ``mrope_section*2`` take their angle from axis ``i % 3`` (t,h,w,t,h,w)."""
    # inv_freq is evaluated in float64 and rounded once to fp32 (torch's fp32 powf differs
    # from numpy's by <=1 ulp per element; at position p that is an angle error of p*6e-8,
    # the same size as the fp32 rounding of the angle itself).
    inv = (1.0 / (float(theta) ** (np.arange(0, head_dim, 2, dtype=np.float64) / head_dim))).astype(F32)
    fr = position_ids[..., None].astype(F32) * inv  # [3,B,P,hd/2]
    emb = np.concatenate([fr, fr], axis=-1)  # [3,B,P,hd]
    cos3, sin3 = np.cos(emb).astype(F32), np.sin(emb).astype(F32)
    sec = list(mrope_section) * 2
    cos_parts, sin_parts, off = [], [], 0
    for i, s in enumerate(sec):
        cos_parts.append(cos3[i % 3, ..., off:off + s])
        sin_parts.append(sin3[i % 3, ..., off:off + s])
        off += s
    return np.concatenate(cos_parts, -1), np.concatenate(sin_parts, -1)


# =============================================================================
# vision tower
# =============================================================================


def vision_window_index(grid_thw: Sequence[Sequence[int]], spatial_merge_size: int, window_size: int, patch_size: int):
    """``get_vision_window_index`` (TF:vision_utils.py:130-188) of Qwen2.5-VL: the order in which the merged
    (2x2-patch) units are visited window by window, and the cumulative window lengths in PATCHES.  Windows are
    ``window_size // merge // patch`` merged units on a side; the grid is padded on the right / bottom to a whole
    number of windows (by a full window when it already divides: those empty windows drop out)."""
    index_all, cu = [], [0]
    base = 0
    ws = window_size // spatial_merge_size // patch_size
    unit = spatial_merge_size ** 2
    for t, h, w in grid_thw:
        lh, lw = h // spatial_merge_size, w // spatial_merge_size
        idx = np.arange(t * lh * lw).reshape(t, lh, lw)
        ph, pw = ws - lh % ws, ws - lw % ws
        nh, nw = (lh + ph) // ws, (lw + pw) // ws
        pad = np.full((t, lh + ph, lw + pw), -100, np.int64)
        pad[:, :lh, :lw] = idx
        pad = pad.reshape(t, nh, ws, nw, ws).transpose(0, 1, 3, 2, 4).reshape(t, nh * nw, ws, ws)
        seqlens = (pad != -100).sum((2, 3)).reshape(-1)
        flat = pad.reshape(-1)
        index_all.append(flat[flat != -100] + base)
        cu.extend((np.cumsum(seqlens) * unit + cu[-1]).tolist())
        base += t * lh * lw
    cu = np.asarray(cu, np.int64)
    cu = cu[np.concatenate([[True], cu[1:] != cu[:-1]])]      # unique_consecutive
    return np.concatenate(index_all), cu


def _attention_segments(q, k, v, seg, H, hd, scale, pol):
    n = q.shape[0]
    o = np.zeros((n, H, hd), dtype=F32)
    for s in range(len(seg) - 1):
        a, b = int(seg[s]), int(seg[s + 1])
        for hh in range(H):  # one 2-D BLAS call per head (numpy's batched matmul is not BLAS-backed)
            qh, kh, vh = (np.ascontiguousarray(t[a:b, hh]) for t in (q, k, v))
            sc = (qh @ kh.T).astype(F32, copy=False)
            sc *= scale
            pr = pol(softmax_lastdim(sc))
            o[a:b, hh] = pr @ vh
    return o


def vit_forward_qwen2_5(pixel_values: np.ndarray, grid_thw: Sequence[Sequence[int]], weights: Dict[str, np.ndarray],
                        vcfg, policy: str = "fp32", return_intermediates: bool = False):
    """``Qwen2_5_VisionTransformerPretrainedModel.forward`` (TF25 = transformers/models/qwen2_5_vl/
    modeling_qwen2_5_vl.py:408-472): PatchEmbed -> tokens and rotary tables reordered window by window (units of
    2x2 patches) -> ``depth`` x Qwen2_5_VLVisionBlock (:294-322: RMSNorm, attention inside the windows except in
    the ``fullatt_block_indexes`` blocks, biased SwiGLU MLP :85-97) -> PatchMerger with RMSNorm (:137-150) ->
    merged tokens put back in image order."""
    pol = _Policy(policy)
    pre = "model.visual."
    D, H = vcfg.embed_dim, vcfg.num_heads
    hd = D // H
    unit = vcfg.spatial_merge_size ** 2
    inter = {}
    x = pol(linear(pol(pixel_values), _w(weights, pre + "patch_embed.proj.weight").reshape(D, -1)))
    inter["patch_embed"] = x
    n = x.shape[0]
    widx, cu_win = vision_window_index(grid_thw, vcfg.spatial_merge_size, vcfg.window_size, vcfg.patch_size)
    inter["window_index"], inter["cu_window_seqlens"] = widx, cu_win
    x = x.reshape(n // unit, unit, D)[widx].reshape(n, D)
    pos = vision_position_ids(grid_thw, vcfg.spatial_merge_size)
    cos, sin = vision_rotary_cos_sin(pos, hd)
    cos = cos.reshape(n // unit, unit, -1)[widx].reshape(n, -1)
    sin = sin.reshape(n // unit, unit, -1)[widx].reshape(n, -1)
    seg_full = np.cumsum([0] + [t * h * w for t, h, w in grid_thw])
    scale = F32(hd ** -0.5)
    eps = 1e-6
    for li in range(vcfg.depth):
        p = f"{pre}blocks.{li}."
        h1 = pol(rms_norm(x, _w(weights, p + "norm1.weight"), eps, pol))
        qkv = pol(linear(h1, _w(weights, p + "attn.qkv.weight"), _w(weights, p + "attn.qkv.bias"))).reshape(n, 3, H, hd)
        q, k, v = qkv[:, 0], qkv[:, 1], qkv[:, 2]
        q = pol(q * cos[:, None, :] + rotate_half(q) * sin[:, None, :])
        k = pol(k * cos[:, None, :] + rotate_half(k) * sin[:, None, :])
        seg = seg_full if li in tuple(vcfg.fullatt_block_indexes) else cu_win
        o = pol(_attention_segments(q, k, v, seg, H, hd, scale, pol).reshape(n, D))
        x = pol(x + linear(o, _w(weights, p + "attn.proj.weight"), _w(weights, p + "attn.proj.bias")))
        h2 = pol(rms_norm(x, _w(weights, p + "norm2.weight"), eps, pol))
        g = pol(linear(h2, _w(weights, p + "mlp.gate_proj.weight"), _w(weights, p + "mlp.gate_proj.bias")))
        u = pol(linear(h2, _w(weights, p + "mlp.up_proj.weight"), _w(weights, p + "mlp.up_proj.bias")))
        a = pol(pol(silu(g)) * u)
        x = pol(x + linear(a, _w(weights, p + "mlp.down_proj.weight"), _w(weights, p + "mlp.down_proj.bias")))
        inter[f"block{li}"] = x
    m = pol(rms_norm(x, _w(weights, pre + "merger.ln_q.weight"), eps, pol)).reshape(-1, D * unit)
    m = pol(gelu_erf(linear(m, _w(weights, pre + "merger.mlp.0.weight"), _w(weights, pre + "merger.mlp.0.bias"))))
    m = pol(linear(m, _w(weights, pre + "merger.mlp.2.weight"), _w(weights, pre + "merger.mlp.2.bias")))
    m = m[np.argsort(widx)]
    inter["merged"] = m
    return (m, inter) if return_intermediates else m


def vit_forward(pixel_values: np.ndarray, grid_thw: Sequence[Sequence[int]], weights: Dict[str, np.ndarray],
                vcfg, policy: str = "fp32", return_intermediates: bool = False):
    """``Qwen2VisionTransformerPretrainedModel.forward`` (TF:modeling_qwen2_vl.py:700-731):
    PatchEmbed (:251-274, Conv3d ≡ GEMM) → ``depth`` × Qwen2VLVisionBlock (:425-449) with
    full non-causal attention per image (:342-422) → PatchMerger (:277-290).
    ``vcfg`` is any object with the VisionConfig attributes.  Returns merged ``[T, d]``."""
    if getattr(vcfg, "variant", "qwen2") == "qwen2_5":
        return vit_forward_qwen2_5(pixel_values, grid_thw, weights, vcfg, policy, return_intermediates)
    pol = _Policy(policy)
    pre = "model.visual."
    D, H = vcfg.embed_dim, vcfg.num_heads
    hd = D // H
    inter = {}
    x = pol(pixel_values)
    wpe = _w(weights, pre + "patch_embed.proj.weight").reshape(D, -1)
    x = pol(linear(x, wpe))
    inter["patch_embed"] = x
    pos = vision_position_ids(grid_thw, vcfg.spatial_merge_size)
    cos, sin = vision_rotary_cos_sin(pos, hd)
    inter["cos"], inter["sin"] = cos, sin
    seg = np.cumsum([0] + [t * h * w for t, h, w in grid_thw])
    scale = F32(hd ** -0.5)
    for li in range(vcfg.depth):
        p = f"{pre}blocks.{li}."
        h1 = pol(layer_norm(x, _w(weights, p + "norm1.weight"), _w(weights, p + "norm1.bias")))
        qkv = pol(linear(h1, _w(weights, p + "attn.qkv.weight"), _w(weights, p + "attn.qkv.bias")))
        n = qkv.shape[0]
        qkv = qkv.reshape(n, 3, H, hd)
        q, k, v = qkv[:, 0], qkv[:, 1], qkv[:, 2]  # [N,H,hd]
        # apply_rotary_pos_emb_vision: fp32 math, result cast back (:225-236)
        q = pol(q * cos[:, None, :] + rotate_half(q) * sin[:, None, :])
        k = pol(k * cos[:, None, :] + rotate_half(k) * sin[:, None, :])
        o = np.zeros((n, H, hd), dtype=F32)
        for s in range(len(seg) - 1):
            a, b = seg[s], seg[s + 1]
            for hh in range(H):  # one 2-D BLAS call per head (numpy's batched matmul is not BLAS-backed)
                qh, kh, vh = (np.ascontiguousarray(t[a:b, hh]) for t in (q, k, v))  # [n,hd]
                sc = (qh @ kh.T).astype(F32, copy=False)
                sc *= scale
                pr = pol(softmax_lastdim(sc))  # softmax fp32, cast to activation dtype (:334)
                o[a:b, hh] = pr @ vh
        o = pol(o.reshape(n, D))
        x = pol(x + linear(o, _w(weights, p + "attn.proj.weight"), _w(weights, p + "attn.proj.bias")))
        h2 = pol(layer_norm(x, _w(weights, p + "norm2.weight"), _w(weights, p + "norm2.bias")))
        f1 = pol(quick_gelu(linear(h2, _w(weights, p + "mlp.fc1.weight"), _w(weights, p + "mlp.fc1.bias"))))
        x = pol(x + linear(f1, _w(weights, p + "mlp.fc2.weight"), _w(weights, p + "mlp.fc2.bias")))
        inter[f"block{li}"] = x
    m = pol(layer_norm(x, _w(weights, pre + "merger.ln_q.weight"), _w(weights, pre + "merger.ln_q.bias")))
    m = m.reshape(-1, vcfg.embed_dim * vcfg.spatial_merge_size ** 2)
    m = pol(gelu_erf(linear(m, _w(weights, pre + "merger.mlp.0.weight"), _w(weights, pre + "merger.mlp.0.bias"))))
    m = pol(linear(m, _w(weights, pre + "merger.mlp.2.weight"), _w(weights, pre + "merger.mlp.2.bias")))
    inter["merged"] = m
    return (m, inter) if return_intermediates else m


# =============================================================================
# text decoder
# =============================================================================


@dataclass
class KVCache:
    """Per-layer lists of ``[B, KVH, S, hd]`` arrays (grown by concatenation)."""
    k: List[Optional[np.ndarray]]
    v: List[Optional[np.ndarray]]

    @classmethod
    def empty(cls, n_layers: int) -> "KVCache":
        return cls([None] * n_layers, [None] * n_layers)

    def length(self) -> int:
        return 0 if self.k[0] is None else self.k[0].shape[2]


def apply_mrope(x_bhsd: np.ndarray, cos: np.ndarray, sin: np.ndarray) -> np.ndarray:
    """``q*cos + rotate_half(q)*sin`` with cos/sin ``[B,S,hd]`` broadcast over heads
    (TF:modeling_qwen2_vl.py:219-221)."""
    return (x_bhsd * cos[:, None] + rotate_half(x_bhsd) * sin[:, None]).astype(F32)


def decoder_layer(x: np.ndarray, li: int, weights, tcfg, cos, sin, cache: KVCache, pol: _Policy,
                  attn_valid: Optional[np.ndarray] = None) -> np.ndarray:
    """``Qwen2VLDecoderLayer`` (TF:modeling_qwen2_vl.py:559-624) with ``Qwen2VLAttention``
    (:469-556; q/k/v bias, no o bias, GQA by repeat_kv :305-314, scale hd^-0.5, fp32 softmax)
    and ``Qwen2MLP`` (:453-466).  ``x`` is ``[B,S,d]``; the cache is appended in place.
    ``attn_valid`` ``[B, S_total]`` masks padded cache slots (ragged batches)."""
    p = f"model.language_model.layers.{li}."
    B, S, d = x.shape
    H, KVH, hd = tcfg.num_heads, tcfg.num_kv_heads, tcfg.head_dim
    aq = fake_quant_rows_fp8 if (pol.act_fp8 and S > 1) else (lambda t: t)   # W8A8 prefill: per-token fp8 activations
    h = rms_norm(x, _w(weights, p + "input_layernorm.weight"), tcfg.rms_norm_eps, pol)
    h = aq(pol(h))
    q = pol(linear(h, _w(weights, p + "self_attn.q_proj.weight"), _w(weights, p + "self_attn.q_proj.bias")))
    k = pol(linear(h, _w(weights, p + "self_attn.k_proj.weight"), _w(weights, p + "self_attn.k_proj.bias")))
    v = pol(linear(h, _w(weights, p + "self_attn.v_proj.weight"), _w(weights, p + "self_attn.v_proj.bias")))
    q = q.reshape(B, S, H, hd).transpose(0, 2, 1, 3)
    k = k.reshape(B, S, KVH, hd).transpose(0, 2, 1, 3)
    v = v.reshape(B, S, KVH, hd).transpose(0, 2, 1, 3)
    q = pol(apply_mrope(q, cos, sin))
    k = pol(apply_mrope(k, cos, sin))
    past = cache.length() if cache.k[li] is not None else 0
    if cache.k[li] is None:
        cache.k[li], cache.v[li] = k, v
    else:
        cache.k[li] = np.concatenate([cache.k[li], k], axis=2)
        cache.v[li] = np.concatenate([cache.v[li], v], axis=2)
    kk, vv = cache.k[li], cache.v[li]
    St = kk.shape[2]
    g = H // KVH
    kk_r = np.repeat(kk, g, axis=1)
    vv_r = np.repeat(vv, g, axis=1)
    sc = np.matmul(q, kk_r.transpose(0, 1, 3, 2)).astype(F32) * F32(hd ** -0.5)
    qi = np.arange(S)[:, None] + past
    ki = np.arange(St)[None, :]
    mask = ki <= qi  # causal
    if attn_valid is not None:
        mask = mask[None, None] & attn_valid[:, None, None, :]
    sc = np.where(mask, sc, np.finfo(F32).min)  # HF masks with finfo.min, not -inf
    pr = pol(softmax_lastdim(sc))
    o = np.matmul(pr, vv_r).astype(F32)
    o = pol(o.transpose(0, 2, 1, 3).reshape(B, S, H * hd))
    x = pol(x + linear(aq(o), _w(weights, p + "self_attn.o_proj.weight")))
    h2 = aq(pol(rms_norm(x, _w(weights, p + "post_attention_layernorm.weight"), tcfg.rms_norm_eps, pol)))
    gate = linear(h2, _w(weights, p + "mlp.gate_proj.weight"))
    up = linear(h2, _w(weights, p + "mlp.up_proj.weight"))
    if pol.name == "bf16":
        # HF bf16 rounds gate and up separately before silu*mul; the HIP engine fuses
        # silu(g)*u on fp32 accumulators and rounds once.  The oracle follows the engine.
        act = pol(silu(gate) * up)
    else:
        act = silu(gate) * up
    x = pol(x + linear(aq(act), _w(weights, p + "mlp.down_proj.weight")))
    return x


def lm_head_weight(weights, tcfg) -> np.ndarray:
    if tcfg.tie_word_embeddings or "lm_head.weight" not in weights:
        return _w(weights, "model.language_model.embed_tokens.weight")
    return _w(weights, "lm_head.weight")


def decoder_forward(inputs_embeds: np.ndarray, position_ids: np.ndarray, weights, tcfg, cache: KVCache,
                    policy: str = "fp32", last_only: bool = True) -> np.ndarray:
    """``Qwen2VLTextModel.forward`` (TF:modeling_qwen2_vl.py:762-845) + ``lm_head`` (:1320-1323).
    ``inputs_embeds [B,S,d]``, ``position_ids [3,B,S]``.  Returns fp32 logits
    (``[B,V]`` for the last position, or ``[B,S,V]``)."""
    pol = _Policy(policy)
    cos, sin = mrope_cos_sin(position_ids, tcfg.head_dim, tcfg.rope_theta, tcfg.mrope_section)
    if pol.name == "bf16":
        cos, sin = pol(cos), pol(sin)  # HF casts cos/sin to x.dtype (:169)
    x = pol(inputs_embeds)
    for li in range(tcfg.num_layers):
        x = decoder_layer(x, li, weights, tcfg, cos, sin, cache, pol)
    x = pol(rms_norm(x, _w(weights, "model.language_model.norm.weight"), tcfg.rms_norm_eps, pol))
    if last_only:
        x = x[:, -1]
    return linear(x, lm_head_weight(weights, tcfg))


def embed_and_scatter(input_ids: np.ndarray, image_embeds: Optional[np.ndarray], weights, cfg) -> np.ndarray:
    """``embed_tokens[input_ids]`` then ``masked_scatter`` of the image embeds at
    ``input_ids == image_token_id`` in row-major order (TF:modeling_qwen2_vl.py:1159-1168)."""
    emb = _w(weights, "model.language_model.embed_tokens.weight")[np.asarray(input_ids)]
    emb = np.array(emb, dtype=F32)
    if image_embeds is not None:
        mask = np.asarray(input_ids) == cfg.image_token_id
        if int(mask.sum()) != image_embeds.shape[0]:
            raise ValueError("Image features and image tokens do not match")
        emb[mask] = image_embeds
    return emb


def resize_bicubic_u8(img: np.ndarray, out_h: int, out_w: int, tables) -> np.ndarray:
    """PIL's ``Image.resize(..., BICUBIC)`` on an HWC uint8 image, restated (Pillow src/libImaging/Resample.c:
    ImagingResampleHorizontal_8bpc then ImagingResampleVertical_8bpc; an axis whose size does not change is not
    resampled).  ``tables(in_size, out_size)`` -> (bounds, coeffs): the build's image_processing.resample_tables."""
    def one_axis(a, out_size):            # resamples axis 1 of [rows, in_size, C]
        bounds, coeffs = tables(a.shape[1], out_size)
        out = np.empty((a.shape[0], out_size, a.shape[2]), np.uint8)
        a64 = a.astype(np.int64)
        for xx in range(out_size):
            x0, n = int(bounds[xx, 0]), int(bounds[xx, 1])
            ss = (a64[:, x0:x0 + n, :] * coeffs[xx, :n].astype(np.int64)[None, :, None]).sum(1) + (1 << 21)
            out[:, xx, :] = np.clip(ss >> 22, 0, 255).astype(np.uint8)
        return out
    img = np.asarray(img, np.uint8)
    h, w, _ = img.shape
    if out_w != w:
        img = one_axis(img, out_w)
    if out_h != h:
        img = one_axis(img.transpose(1, 0, 2), out_h).transpose(1, 0, 2)
    return np.ascontiguousarray(img)


def _mix32(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, np.uint64) & 0xFFFFFFFF
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & 0xFFFFFFFF
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & 0xFFFFFFFF
    x ^= x >> np.uint64(16)
    return x


def gumbel_noise(seed: int, n: int, vocab: int) -> np.ndarray:
    """G_i of the build's sampler (kr_gumbel_argmax, kr_decode.hip): -ln(-ln(u_i)), u_i = ((h_i >> 9) + 0.5) 2^-23,
    h_i = mix(mix(seed ^ n * 0x9E3779B1) + i), n = index of the token being generated.  fp32 like the kernel.
    (23 bits + 0.5 is exact in fp32, so u stays inside [2^-24, 1 - 2^-24] and the noise is finite: see gumbel_u.)"""
    base = _mix32(np.uint64((int(seed) ^ ((int(n) * 0x9E3779B1) & 0xFFFFFFFF)) & 0xFFFFFFFF))
    h = _mix32((base + np.arange(vocab, dtype=np.uint64)) & 0xFFFFFFFF)
    return -np.log(-np.log(gumbel_u(h), dtype=np.float32), dtype=np.float32)


def gumbel_u(h: np.ndarray) -> np.ndarray:
    """32-bit hash -> uniform in the OPEN interval (0, 1), exactly representable in fp32: ((h >> 9) + 0.5) * 2^-23."""
    return ((np.asarray(h, np.uint64) >> np.uint64(9)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -23)


def sample_scores(logits: np.ndarray, temperature: float, seed: int, n: int) -> np.ndarray:
    """What the sampler takes the argmax of: logits / T + G (T > 0), the logits themselves (T == 0)."""
    logits = np.asarray(logits, np.float32)
    if temperature <= 0:
        return logits
    return logits * np.float32(1.0 / np.float32(temperature)) + gumbel_noise(seed, n, logits.shape[-1])


# ----------------------------------------------------------------------------- guided decoding + log-probabilities
# What vLLM's guided-decoding logits processor and its `logprobs` output compute for the requests the reference
# sends (guided_regex: /root/reference/karanta/pipeline.py:304-307; response_format:
# /root/reference/bulk_processing/workers/vllm_client.py:196; logprobs / top_logprobs:
# /root/reference/karanta/data/create_batch_data_prompts.py:117-118), restated over a byte DFA given as data
# (trans [S, 256], accept [S], state 0 = dead): a token is allowed iff it has bytes and they keep the automaton
# alive; EOS is allowed iff the state accepts.  Plain loops: this is the checker, not the product.
def guide_walk(trans: np.ndarray, state: int, data: bytes) -> int:
    for byte in data:
        if state == 0:
            break
        state = int(trans[state, byte])
    return state


def guide_token_mask(trans: np.ndarray, accept: np.ndarray, state: int, token_bytes: Sequence[bytes],
                     eos_ids: Sequence[int]) -> np.ndarray:
    allowed = np.zeros(len(token_bytes), dtype=bool)
    for i, tb in enumerate(token_bytes):
        allowed[i] = state != 0 and len(tb) > 0 and guide_walk(trans, state, tb) != 0
    for e in eos_ids:
        if 0 <= e < len(token_bytes):
            allowed[e] = state != 0 and bool(accept[state])
    return allowed


def log_softmax(logits: np.ndarray) -> np.ndarray:
    x = np.asarray(logits, dtype=np.float64)
    m = x.max(axis=-1, keepdims=True)
    return x - m - np.log(np.exp(x - m).sum(axis=-1, keepdims=True))


def top_logprobs(logits: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """(ids [k], log-probs [k]) of the k most probable tokens of one row, ties to the lowest id."""
    lp = log_softmax(logits)
    order = np.argsort(-np.asarray(logits, dtype=np.float64), kind="stable")[:k]
    return order.astype(np.int64), lp[order]


def generate_greedy(cfg, weights, input_ids: np.ndarray, pixel_values: Optional[np.ndarray],
                    image_grid_thw: Optional[Sequence[Sequence[int]]], max_new_tokens: int,
                    policy: str = "fp32", ignore_eos: bool = False,
                    return_logits: bool = False, temperature: float = 0.0, seed: int = 0,
                    guide=None, token_bytes: Optional[Sequence[bytes]] = None, return_raw_logits: bool = False,
                    image_embeds: Optional[np.ndarray] = None):
    """The call sequence of /root/reference/karanta/training/test_trained_model.py:76-99
    (``model.generate(**inputs, max_new_tokens=N)`` with ``do_sample=False``), i.e. what a
    ``temperature=0`` request to the reference's vLLM server computes
    (/root/reference/karanta/pipeline.py:166-171): ViT → scatter → prefill → greedy decode,
    stopping at EOS.  Single-sequence or equal-length batch (no padding).
    temperature > 0 (the reference's first attempt sends 0.1, pipeline.py:281,301): Gumbel-max sampling with the
    build's counter-based noise (`sample_scores`); the returned "logits" are then the noisy scores.
    guide = (trans, accept, start) with token_bytes: scores of tokens the automaton forbids become -inf before the
    argmax (after temperature / noise), the state follows the chosen token's bytes.  return_raw_logits: also the
    model's own logits per step (what log-probabilities are taken from).  image_embeds: the merged ViT output when the
    caller has already run :func:`vit_forward` on these pixels (tests that also check the ViT, or that run two decoders
    on one image) — pixel_values is then not needed."""
    input_ids = np.asarray(input_ids)
    B, P = input_ids.shape
    img = image_embeds
    if img is None and pixel_values is not None:
        img = vit_forward(pixel_values, image_grid_thw, weights, cfg.vision, policy)
    emb = embed_and_scatter(input_ids, img, weights, cfg)
    if image_grid_thw is not None:
        pos, delta = get_rope_index(input_ids, image_grid_thw, cfg.image_token_id, cfg.vision.spatial_merge_size)
    else:
        pos = np.tile(np.arange(P)[None, None, :], (3, B, 1))
        delta = np.zeros((B,), dtype=np.int64)
    cache = KVCache.empty(cfg.text.num_layers)
    gstate = [guide[2]] * B if guide is not None else None

    def noisy(lg, n):
        sc = np.stack([sample_scores(lg[b], temperature, seed, n) for b in range(B)])
        if guide is not None:
            for b in range(B):
                ok = guide_token_mask(guide[0], guide[1], gstate[b], token_bytes, cfg.eos_token_ids)
                sc[b] = np.where(ok, sc[b], -np.inf)
        return sc

    raw = decoder_forward(emb, pos, weights, cfg.text, cache, policy)
    raw_logits = [raw]
    logits = noisy(raw, 0)
    all_logits = [logits]
    out = np.zeros((B, 0), dtype=np.int64)
    done = np.zeros((B,), dtype=bool)
    for step in range(max_new_tokens):
        nxt = logits.argmax(axis=-1)
        nxt = np.where(done, cfg.pad_token_id, nxt)
        if guide is not None:
            for b in range(B):
                if not done[b] and int(nxt[b]) not in cfg.eos_token_ids:
                    gstate[b] = guide_walk(guide[0], gstate[b], token_bytes[int(nxt[b])])
        out = np.concatenate([out, nxt[:, None]], axis=1)
        if not ignore_eos:
            done |= np.isin(nxt, cfg.eos_token_ids)
            if done.all():
                break
        if step == max_new_tokens - 1:
            break
        e = embed_and_scatter(nxt[:, None], None, weights, cfg)
        ppos = np.tile((P + step + delta)[None, :, None], (3, 1, 1))
        raw = decoder_forward(e, ppos, weights, cfg.text, cache, policy)
        raw_logits.append(raw)
        logits = noisy(raw, step + 1)
        all_logits.append(logits)
    if return_raw_logits:
        return out, np.stack(all_logits, axis=1), np.stack(raw_logits, axis=1)
    if return_logits:
        return out, np.stack(all_logits, axis=1)
    return out
