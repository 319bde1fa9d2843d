"""Host-side integer work of the hot path: rotary position ids / tables and the work lists the
attention kernels consume.  Pure numpy (cheap, O(tokens)); arithmetic stays on the GPU.

Semantics follow Hugging Face Qwen2-VL ("TF:" = transformers/models/qwen2_vl/modeling_qwen2_vl.py,
line numbers as cited in SURVEY.md §8 a-ii), which is what the reference's server computes for a
``create_vision_message`` request (/root/reference/karanta/data/utils.py:269-297).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, List, Sequence, Tuple

import numpy as np

from .weights import bf16_round

F32 = np.float32


# ----------------------------------------------------------------------------- ViT rotary
def vision_position_ids(grid_thw: Sequence[Sequence[int]], merge: int = 2) -> np.ndarray:
    """(h, w) index of every patch, 2x2-merge-block-major (TF:vision_utils.py:81-127)."""
    out = []
    for t, h, w in grid_thw:
        hh = np.arange(h)[:, None].repeat(w, 1)
        ww = np.arange(w)[None, :].repeat(h, 0)
        shp = (h // merge, merge, w // merge, merge)
        hh = hh.reshape(shp).transpose(0, 2, 1, 3).reshape(-1)
        ww = ww.reshape(shp).transpose(0, 2, 1, 3).reshape(-1)
        out.append(np.tile(np.stack([hh, ww], -1), (t, 1)))
    return np.concatenate(out, 0).astype(np.int64) if out else np.zeros((0, 2), np.int64)


def vision_rotary_tables(grid_thw: Sequence[Sequence[int]], head_dim: int, merge: int = 2,
                         theta: float = 10000.0) -> Tuple[np.ndarray, np.ndarray]:
    """fp32 cos/sin ``[N, head_dim]``: VisionRotaryEmbedding(head_dim//2) then cat(freqs, freqs)
    (TF:239-248, :711-713).  Kept in fp32: apply_rotary_pos_emb_vision computes in fp32 (TF:225-236)."""
    pos = vision_position_ids(grid_thw, merge)
    dim = head_dim // 2
    inv = (1.0 / (float(theta) ** (np.arange(0, dim, 2, dtype=np.float64) / dim))).astype(F32)
    fr = (pos[:, :, None].astype(F32) * inv[None, None, :]).reshape(pos.shape[0], -1)
    emb = np.concatenate([fr, fr], -1)
    return np.cos(emb).astype(F32), np.sin(emb).astype(F32)


# ----------------------------------------------------------------------------- M-RoPE
def rope_index_one(input_ids: np.ndarray, grids: Sequence[Sequence[int]], image_token_id: int,
                   merge: int = 2) -> Tuple[np.ndarray, int]:
    """3-D position ids ``[3, P]`` and rope delta of ONE unpadded prompt (TF:914-1016, :862-912).
    ``grids`` are the (t, gh, gw) of this prompt's images, in order."""
    ids = np.asarray(input_ids).reshape(-1)
    p = ids.shape[0]
    is_img = ids == image_token_id
    # run boundaries
    change = np.flatnonzero(np.diff(is_img.astype(np.int8))) + 1
    starts = np.concatenate([[0], change])
    ends = np.concatenate([change, [p]])
    cols: List[np.ndarray] = []
    cur = 0
    gi = 0
    for a, b in zip(starts, ends):
        n = int(b - a)
        if not is_img[a]:
            cols.append(np.tile(np.arange(n, dtype=np.int64)[None, :] + cur, (3, 1)))
            cur += n
        else:
            if gi >= len(grids):
                raise ValueError("more image-token runs than image grids")
            t, gh, gw = (int(v) for v in grids[gi])
            gi += 1
            lh, lw = gh // merge, gw // merge
            if t * lh * lw != n:
                raise ValueError(f"Image features and image tokens do not match, tokens: {n}, features: {t * lh * lw}")
            tt = np.repeat(np.arange(t, dtype=np.int64), lh * lw) + cur
            hh = np.tile(np.repeat(np.arange(lh, dtype=np.int64), lw), t) + cur
            ww = np.tile(np.arange(lw, dtype=np.int64), t * lh) + cur
            cols.append(np.stack([tt, hh, ww], 0))
            cur += max(gh, gw) // merge
    if gi != len(grids):
        raise ValueError("fewer image-token runs than image grids")
    pos = np.concatenate(cols, 1) if cols else np.zeros((3, 0), np.int64)
    delta = int(pos.max()) + 1 - p if p else 0
    return pos, delta


def mrope_tables(pos3: np.ndarray, head_dim: int, theta: float, mrope_section: Sequence[int],
                 round_bf16: bool = True) -> Tuple[np.ndarray, np.ndarray]:
    """cos/sin ``[P, head_dim]`` fp32 with the t/h/w section interleave applied
    (Qwen2VLRotaryEmbedding TF:117-169 + channel selection of apply_multimodal_rotary_pos_emb
    TF:180-222).  ``round_bf16``: HF returns cos/sin in the activation dtype (TF:169), so a bf16
    model multiplies by bf16-rounded tables."""
    inv = (1.0 / (float(theta) ** (np.arange(0, head_dim, 2, dtype=np.float64) / head_dim))).astype(F32)
    fr = pos3[..., None].astype(F32) * inv  # [3, P, hd/2]
    sel = np.concatenate([np.full(s, i % 3) for i, s in enumerate(mrope_section)])  # axis per freq index
    ang = np.take_along_axis(fr, sel[None, None, :].repeat(fr.shape[1], 1), axis=0)[0]  # [P, hd/2]
    emb = np.concatenate([ang, ang], -1)
    cos, sin = np.cos(emb).astype(F32), np.sin(emb).astype(F32)
    if round_bf16:
        cos, sin = bf16_round(cos), bf16_round(sin)
    return cos, sin


def rope_inv_freq(head_dim: int, theta: float) -> np.ndarray:
    return (1.0 / (float(theta) ** (np.arange(0, head_dim, 2, dtype=np.float64) / head_dim))).astype(F32)


# ----------------------------------------------------------------------------- attention work lists
@dataclass
class AttnPlan:
    """Work lists for kr_qkv_prep / kr_attn_varlen over a set of segments (images or sequences).

    Segment s holds ``lens[s]`` tokens, contiguous in the flattened activation starting at row
    ``tok0[s]``; its keys live at K rows ``k_row0[s] + j`` and V^T blocks ``vt_blk0[s] + j // 64``.
    """
    blk_tok0: np.ndarray   # int32 [n_blk]
    blk_ntok: np.ndarray   # int32 [n_blk]
    blk_k_row0: np.ndarray  # int64 [n_blk]
    blk_vt_blk: np.ndarray  # int64 [n_blk]
    qblk: np.ndarray       # int32 [n_qblk, 4]
    qblk_len: np.ndarray   # int32 [n_qblk, 2]
    n_tokens: int
    n_vt_blocks: int
    q_block: int = 128     # queries per work-list entry: 128 (4-wave workgroups) or 256 (8-wave, kr_attn_varlen_q)


def pick_q_block(lens: Sequence[int], causal: bool = False) -> int:
    """Queries per attention workgroup: 128 (4 waves) or 256 (8 waves sharing one staged K / V^T tile).  r2 microbench on
    the final kernel (outputs bit-identical either way): ViT 8 x 4900 tokens 1.154 ms (128) vs 1.127 (256) per block; ONE
    19 276-token page (config 5) 2.166 vs 2.071 ms; causal prefill 8 x 1394 tokens 0.082 vs 0.096 ms per layer — its
    query blocks see 1 .. 22 key tiles, and half as many, twice as long workgroups schedule worse.  The 8-wave shape
    pays where a full-attention segment has dozens of K / V tiles to stage; short segments (Qwen2.5-VL's 64-token
    windows) would leave most of its waves without queries."""
    lens = [int(n) for n in lens if int(n) > 0]
    return 256 if (not causal) and lens and sum(lens) / len(lens) >= 2048 else 128


def make_attn_plan(lens: Sequence[int], k_row0: Sequence[int], vt_blk0: Sequence[int], causal: bool,
                   q_block: Optional[int] = None) -> AttnPlan:
    if q_block is None:
        q_block = pick_q_block(lens, causal)
    if q_block not in (128, 256):
        raise ValueError(f"q_block {q_block} (128 or 256)")
    blk_tok0, blk_ntok, blk_kr, blk_vb, qblk, qlen = [], [], [], [], [], []
    tok = 0
    nvb = 0
    for s, n in enumerate(lens):
        n = int(n)
        for j in range(0, n, 64):
            blk_tok0.append(tok + j)
            blk_ntok.append(min(64, n - j))
            blk_kr.append(int(k_row0[s]) + j)
            blk_vb.append(int(vt_blk0[s]) + j // 64)
        for j in range(0, n, q_block):
            nq = min(q_block, n - j)
            qblk.append((tok + j, nq, int(k_row0[s]), int(vt_blk0[s])))
            qlen.append((n, j))
        tok += n
        nvb += (n + 63) // 64
    # heaviest query blocks first: the kernel walks this table in order (head fastest), so a causal launch — whose
    # blocks see 1 .. n/64 key tiles — is scheduled longest-job-first and its tail is one light block, not a heavy one
    work = [min(n, j + nq) if causal else n for (_, nq, _, _), (n, j) in zip(qblk, qlen)]
    order = sorted(range(len(qblk)), key=lambda i: -((work[i] + 63) // 64))   # stable: ties keep token order
    qblk = [qblk[i] for i in order]
    qlen = [qlen[i] for i in order]
    return AttnPlan(
        np.asarray(blk_tok0, np.int32), np.asarray(blk_ntok, np.int32), np.asarray(blk_kr, np.int64),
        np.asarray(blk_vb, np.int64), np.asarray(qblk, np.int32).reshape(-1, 4),
        np.asarray(qlen, np.int32).reshape(-1, 2), tok, nvb, q_block)


def vit_attn_plan(grid_thw: Sequence[Sequence[int]]) -> AttnPlan:
    """One full-attention segment per image (cu_seqlens of TF:708, :399-418): K rows are the
    token rows themselves; V^T blocks are numbered consecutively, each image starting a new block."""
    lens = [int(t * h * w) for t, h, w in grid_thw]
    k_row0 = np.concatenate([[0], np.cumsum(lens)[:-1]]) if lens else []
    nb = [(n + 63) // 64 for n in lens]
    vt0 = np.concatenate([[0], np.cumsum(nb)[:-1]]) if lens else []
    return make_attn_plan(lens, k_row0, vt0, causal=False)


def vision_window_order(grid_thw: Sequence[Sequence[int]], merge: int, window_size: int, patch_size: int):
    """Qwen2.5-VL window attention (TF25:modeling_qwen2_5_vl.py:430-446, TF:vision_utils.py:130-188): the order in
    which the merged (merge x merge patch) units are visited window by window, and the window lengths in patches.
    Windows are window_size // merge // patch_size units on a side, the grid padded right / bottom to whole windows
    (by a full window when it already divides — those windows are empty and drop out)."""
    ws = window_size // merge // patch_size
    if ws < 1:
        raise ValueError(f"window_size {window_size} is smaller than one merged unit")
    order: List[np.ndarray] = []
    win_lens: List[int] = []
    base = 0
    for t, h, w in grid_thw:
        lh, lw = int(h) // merge, int(w) // merge
        ids = np.arange(int(t) * lh * lw, dtype=np.int64).reshape(int(t), lh, lw)
        for ti in range(int(t)):
            for y0 in range(0, lh, ws):          # windows in row-major order, units inside a window row-major
                for x0 in range(0, lw, ws):
                    blk = ids[ti, y0:y0 + ws, x0:x0 + ws].reshape(-1)
                    order.append(blk + base)
                    win_lens.append(int(blk.size) * merge * merge)
        base += int(t) * lh * lw
    return (np.concatenate(order) if order else np.zeros(0, np.int64)), win_lens


def segments_attn_plan(lens: Sequence[int]) -> AttnPlan:
    """Full (non-causal) attention inside consecutive token segments of the given lengths; every segment starts a
    new 64-token V^T block."""
    lens = [int(n) for n in lens]
    k_row0 = np.concatenate([[0], np.cumsum(lens)[:-1]]) if lens else []
    nb = [(n + 63) // 64 for n in lens]
    vt0 = np.concatenate([[0], np.cumsum(nb)[:-1]]) if lens else []
    return make_attn_plan(lens, k_row0, vt0, causal=False)


def prefill_attn_plan(prompt_lens: Sequence[int], slots: Sequence[int], kv_heads: int, s_max: int) -> AttnPlan:
    """Causal segments = sequences; keys/values live in the KV cache of the sequence's slot:
    K row0 = slot*kv_heads*s_max (+ head*s_max via the head stride), V^T block0 likewise / 64."""
    k_row0 = [int(sl) * kv_heads * s_max for sl in slots]
    vt0 = [int(sl) * kv_heads * (s_max // 64) for sl in slots]
    return make_attn_plan(prompt_lens, k_row0, vt0, causal=True)


# ----------------------------------------------------------------------------- V^T block layout (host view, tests and tools)
def vt_blocks(v_rows: np.ndarray) -> np.ndarray:
    """V rows ``[..., S, hd]`` (S a multiple of 64) -> the V^T block layout of the KV cache / the ViT's V^T buffer, shaped
    ``[..., S/64, hd, 64]`` as the engine's tensors are: a block holds 64 keys of one head, transposed, as TWO CONTIGUOUS
    32-KEY HALVES ``[2][hd][32]`` (csrc/kr_common.h, kr_vt_off) — a decode-attention wave reads one half as whole cache lines."""
    v_rows = np.asarray(v_rows)
    *lead, S, hd = v_rows.shape
    a = np.swapaxes(v_rows.reshape(*lead, S // 64, 2, 32, hd), -1, -2)
    return np.ascontiguousarray(a).reshape(*lead, S // 64, hd, 64)


def vt_rows(vt: np.ndarray) -> np.ndarray:
    """Inverse of :func:`vt_blocks`: ``[..., nb, hd, 64]``-shaped block storage -> V rows ``[..., nb * 64, hd]``."""
    vt = np.asarray(vt)
    *lead, nb, hd, _ = vt.shape
    a = np.swapaxes(vt.reshape(*lead, nb, 2, hd, 32), -1, -2)
    return np.ascontiguousarray(a).reshape(*lead, nb * 64, hd)
