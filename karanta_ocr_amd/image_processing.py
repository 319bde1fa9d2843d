"""Host-side image front end: data-URL / PIL image -> ``pixel_values`` + ``image_grid_thw``.

This is the server side of the reference's request format: ``create_vision_message``
(/root/reference/karanta/data/utils.py:269-297) puts a base64 PNG (or JPEG) data-URL in the
chat message; the vLLM server the reference talks to (/root/reference/karanta/pipeline.py:317)
turns it into Qwen2-VL patches.  The arithmetic follows the Hugging Face PIL processor
(transformers/models/qwen2_vl/image_processing_pil_qwen2_vl.py:57-83 smart_resize,
:126-150 resize, :226-229 rescale/normalize, :152-187 patchify): bicubic PIL resize on the host
(SURVEY.md §7 "bicubic resize parity"), float32 normalisation, patch order
(gh/2, gw/2, 2, 2, C, T, 14, 14) with the frame duplicated T=2 times.
"""
from __future__ import annotations

import base64
import io
import math
from typing import List, Sequence, Tuple

import numpy as np

CLIP_MEAN = np.asarray((0.48145466, 0.4578275, 0.40821073), dtype=np.float32)
CLIP_STD = np.asarray((0.26862954, 0.26130258, 0.27577711), dtype=np.float32)

MIN_PIXELS = 56 * 56
MAX_PIXELS_CLASS_DEFAULT = 28 * 28 * 1280  # 1 003 520: transformers class default ("grid A")
MAX_PIXELS_HUB = 12845056  # Qwen2-VL-Instruct preprocessor_config.json as recalled ("grid B")


def smart_resize(height: int, width: int, factor: int = 28, min_pixels: int = MIN_PIXELS,
                 max_pixels: int = MAX_PIXELS_CLASS_DEFAULT) -> Tuple[int, int]:
    if max(height, width) / min(height, width) > 200:
        raise ValueError(
            f"absolute aspect ratio must be smaller than 200, got {max(height, width) / min(height, width)}")
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


def decode_data_url(url: str):
    """``data:image/png;base64,...`` (or raw base64) -> PIL RGB image."""
    from PIL import Image

    if url.startswith("data:"):
        url = url.split(",", 1)[1]
    img = Image.open(io.BytesIO(base64.b64decode(url)))
    return img.convert("RGB")


def image_to_patches(img, min_pixels: int = MIN_PIXELS, max_pixels: int = MAX_PIXELS_CLASS_DEFAULT,
                     patch: int = 14, merge: int = 2, temporal: int = 2) -> Tuple[np.ndarray, Tuple[int, int, int]]:
    """PIL image or HWC uint8 array -> (pixel_values fp32 [gh*gw, 3*T*p*p], (1, gh, gw))."""
    from PIL import Image

    if isinstance(img, np.ndarray):
        if img.ndim == 2:
            img = np.stack([img] * 3, axis=-1)
        img = Image.fromarray(img.astype(np.uint8), mode="RGB")
    else:
        img = img.convert("RGB")
    w, h = img.size
    rh, rw = smart_resize(h, w, patch * merge, min_pixels, max_pixels)
    if (rh, rw) != (h, w):
        img = img.resize((rw, rh), resample=Image.BICUBIC)
    x = np.asarray(img, dtype=np.uint8).astype(np.float32)  # HWC
    x = x * np.float32(1.0 / 255.0)
    x = (x - CLIP_MEAN) / CLIP_STD
    gh, gw = rh // patch, rw // patch
    # HWC -> (gh/m, m, p, gw/m, m, p, C) -> (gh/m, gw/m, m, m, C, p, p)
    x = x.reshape(gh // merge, merge, patch, gw // merge, merge, patch, 3)
    x = x.transpose(0, 3, 1, 4, 6, 2, 5)
    x = np.broadcast_to(x[:, :, :, :, :, None], (gh // merge, gw // merge, merge, merge, 3, temporal, patch, patch))
    pv = np.ascontiguousarray(x).reshape(gh * gw, 3 * temporal * patch * patch)
    return pv, (1, gh, gw)


# ----------------------------------------------------------------------------- device front end: coefficient tables
# PIL's resize (what the HF PIL processor calls, image_processing_pil_qwen2_vl.py:126-150) is a separable
# convolution in fixed point: per output coordinate a window [xmin, xmin + n) of the input and n integer weights
# (Pillow src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc; 22 fractional bits for 8-bit
# images), horizontal pass into a uint8 image, then vertical pass.  The tables are tiny and are built here in the
# same double arithmetic; kr_image_resize_bicubic_u8 applies them on the GPU, bit-identical to PIL.
PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_tables(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray]:
    """(bounds int32 [out, 2] = (first input index, taps), coeffs int32 [out, ksize]) of PIL's bicubic resample
    of one axis from in_size to out_size (box = the whole axis)."""
    support = 2.0
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = support * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    coeffs = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        if ww != 0.0:
            k = [w / ww for w in k]
        bounds[xx] = (xmin, xmax)
        for x, w in enumerate(k):   # C: (int)(+-0.5 + w * (1 << PRECISION_BITS)), truncation toward zero
            v = w * (1 << PRECISION_BITS)
            coeffs[xx, x] = int(-0.5 + v) if w < 0 else int(0.5 + v)
    return bounds, coeffs


def batch_patches(images: Sequence, **kw) -> Tuple[np.ndarray, List[Tuple[int, int, int]]]:
    pvs, grids = [], []
    for im in images:
        pv, g = image_to_patches(im, **kw)
        pvs.append(pv)
        grids.append(g)
    return (np.concatenate(pvs, axis=0) if pvs else np.zeros((0, 1176), np.float32)), grids


def synthetic_page(index: int, height: int = 1024, width: int = 1024) -> np.ndarray:
    """Seeded synthetic scan: light noisy background, two text columns of
    dark runs.  uint8 RGB ``[H, W, 3]``."""
    rng = np.random.default_rng(20251031 + index)
    img = np.clip(rng.normal(238, 6, size=(height, width)), 0, 255).astype(np.uint8)
    margin = max(8, width // 16)
    col_w = (width - 3 * margin) // 2
    for col in range(2):
        x0 = margin + col * (col_w + margin)
        y = margin
        while y < height - margin:
            rh = int(rng.integers(12, 19))
            x = x0
            while x < x0 + col_w - 18:
                run = int(rng.integers(18, 141))
                run = min(run, x0 + col_w - x)
                img[y:y + rh, x:x + run] = int(rng.integers(15, 71))
                x += run + int(rng.integers(6, 15))
            y += rh + int(rng.integers(6, 13))
    return np.stack([img] * 3, axis=-1)


def encode_png_data_url(img_u8: np.ndarray) -> str:
    from PIL import Image

    buf = io.BytesIO()
    Image.fromarray(img_u8).save(buf, format="PNG")
    return "data:image/png;base64," + base64.b64encode(buf.getvalue()).decode("ascii")
