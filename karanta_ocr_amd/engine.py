"""MI355X inference engine for the karanta OCR hot path: Qwen2-VL ViT -> scatter -> prefill ->
greedy decode, every arithmetic step a hand-written HIP kernel behind include/karanta_hip.h.

This is the thing the reference reaches over HTTP (``vllm serve``,
/root/reference/karanta/pipeline.py:707-742, :317-319) or through Hugging Face ``generate``
(/root/reference/karanta/training/test_trained_model.py:76-99).  PyTorch is used for device
memory and streams only; there is no torch arithmetic and no CPU fallback on this path.

Batching model: one `generate` call = one static batch of <= max_batch pages (one image +
prompt each, any mix of sizes); pages of different requests never interact.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import time
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import image_processing as IP
from . import positions as POS
from ._lib import (DEC_ARGMAX, DEC_OUT_XP, DEC_PLAIN, DEC_ROPE_KV, DEC_SILU8, Dec32, EPI_GELU_ERF, EPI_NONE, EPI_QUICK_GELU, EPI_SILU_MUL8,
                   KarantaHipError, lib, narrow_opts, ptr)
from .config import ModelConfig
from .weights import pack_w16x64, to_bf16_bits

BF16 = torch.bfloat16
GEMM_SCRATCH_BYTES = 512 * 65536   # include/karanta_hip.h: KR_GEMM_SCRATCH_BYTES
FP8_ACT_DEFAULT = False            # fp8 engines: W8A8 prefill on by default? (measured error and speed)


def _bits(w: np.ndarray) -> np.ndarray:
    return w if w.dtype == np.uint16 else to_bf16_bits(np.asarray(w, dtype=np.float32))


def _align(n: int, a: int = 256) -> int:
    return (n + a - 1) // a * a


# =============================================================================
# weights on the device: one packed arena (this is what kr_bcast_weights sends)
# =============================================================================
class DeviceWeights:
    """All parameters in one contiguous HBM arena, laid out for the kernels:

    * ViT Linears keep their [out, in] row-major layout (K contiguous = MFMA fragment order);
    * decoder q/k/v are fused into one [q+2kv, d] matrix, gate/up into one [2*ff, d] matrix with
      rows interleaved in groups of 8 (KR_EPI_SILU_MUL8: one 16-row MFMA tile = 8 gate rows + their 8 up rows);
    * every decoder Linear and the lm_head are stored PACKED as [N/16][K/64][16][64] tiles
      (weights.pack_w16x64): decode streams them linearly from HBM, prefill reads the same copy
      through kr_gemm_bf16(w_packed=1).  A tied lm_head gets its own packed copy (the embedding
      table itself stays row-major for the gather);
    * the patch-embed kernel matrix is zero-padded from K=1176 to 1216 (GEMM BK=64).
    """

    def __init__(self, cfg: ModelConfig, device: torch.device, weight_dtype: str = "bf16"):
        if weight_dtype not in ("bf16", "fp8"):
            raise ValueError(f"weight_dtype {weight_dtype!r} (bf16 or fp8)")
        self.cfg = cfg
        self.device = device
        # fp8 (BASELINE.json config 5): the decoder Linears are ALSO kept as e4m3fn codes + one f32 scale per output
        # row for the decode kernels (half the bytes per step); the bf16 entries then hold the dequantised values
        # (what the prefill GEMMs read).  lm_head, embeddings, norms, biases and the ViT stay bf16.
        self.weight_dtype = weight_dtype
        self.layout: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        self.nbytes = 0
        self.arena: Optional[torch.Tensor] = None
        self._plan()

    def _add(self, name: str, shape: Tuple[int, ...], itemsize: int = 2):
        self.layout[name] = (self.nbytes, tuple(shape))
        self.nbytes = _align(self.nbytes + itemsize * int(np.prod(shape)))

    def _plan(self):
        v, t = self.cfg.vision, self.cfg.text
        self._add("vit.patch", (v.embed_dim, v.patch_dim_padded))
        v25 = v.variant == "qwen2_5"
        for i in range(v.depth):
            p = f"vit.{i}."
            self._add(p + "ln1.w", (v.embed_dim,))
            self._add(p + "qkv.w", (3 * v.embed_dim, v.embed_dim)); self._add(p + "qkv.b", (3 * v.embed_dim,))
            self._add(p + "proj.w", (v.embed_dim, v.embed_dim)); self._add(p + "proj.b", (v.embed_dim,))
            self._add(p + "ln2.w", (v.embed_dim,))
            if v25:
                # biased SwiGLU (TF25:85-97): gate / up fused with rows AND biases interleaved in groups of 8
                # (KR_EPI_SILU_MUL8), width zero-padded to a multiple of 64 (3420 -> 3456)
                fp = v.mlp_dim_padded
                self._add(p + "gate_up.w", (2 * fp, v.embed_dim)); self._add(p + "gate_up.b", (2 * fp,))
                self._add(p + "down.w", (v.embed_dim, fp)); self._add(p + "down.b", (v.embed_dim,))
            else:
                self._add(p + "ln1.b", (v.embed_dim,)); self._add(p + "ln2.b", (v.embed_dim,))
                self._add(p + "fc1.w", (v.mlp_dim, v.embed_dim)); self._add(p + "fc1.b", (v.mlp_dim,))
                self._add(p + "fc2.w", (v.embed_dim, v.mlp_dim)); self._add(p + "fc2.b", (v.embed_dim,))
        self._add("vit.merger.ln.w", (v.embed_dim,))
        if not v25:
            self._add("vit.merger.ln.b", (v.embed_dim,))
        self._add("vit.merger.fc1.w", (v.merge_dim, v.merge_dim)); self._add("vit.merger.fc1.b", (v.merge_dim,))
        self._add("vit.merger.fc2.w", (v.hidden_size, v.merge_dim)); self._add("vit.merger.fc2.b", (v.hidden_size,))
        self._add("llm.embed", (t.vocab_size, t.hidden_size))
        for i in range(t.num_layers):
            p = f"llm.{i}."
            self._add(p + "ln1.w", (t.hidden_size,))
            self._add(p + "qkv.w", (t.qkv_dim, t.hidden_size)); self._add(p + "qkv.b", (t.qkv_dim,))
            self._add(p + "o.w", (t.hidden_size, t.q_dim))
            self._add(p + "ln2.w", (t.hidden_size,))
            self._add(p + "gate_up.w", (2 * t.intermediate_size, t.hidden_size))
            self._add(p + "down.w", (t.hidden_size, t.intermediate_size))
            if self.weight_dtype == "fp8":
                for n, shape in (("qkv", (t.qkv_dim, t.hidden_size)), ("o", (t.hidden_size, t.q_dim)),
                                 ("gate_up", (2 * t.intermediate_size, t.hidden_size)), ("down", (t.hidden_size, t.intermediate_size))):
                    self._add(p + n + ".w8", shape, itemsize=1)
                    self._add(p + n + ".s", (shape[0],), itemsize=4)
        self._add("llm.norm.w", (t.hidden_size,))
        self._add("llm.lm_head", (t.vocab_size, t.hidden_size))

    def allocate(self):
        self.arena = torch.zeros(self.nbytes, dtype=torch.uint8, device=self.device)

    def view(self, name: str) -> torch.Tensor:
        off, shape = self.layout[name]
        n = int(np.prod(shape))
        return self.arena[off:off + 2 * n].view(BF16).view(*shape)

    def view_u8(self, name: str) -> torch.Tensor:
        off, shape = self.layout[name]
        return self.arena[off:off + int(np.prod(shape))].view(*shape)

    def view_f32(self, name: str) -> torch.Tensor:
        off, shape = self.layout[name]
        return self.arena[off:off + 4 * int(np.prod(shape))].view(torch.float32).view(*shape)

    def has(self, name: str) -> bool:
        return name in self.layout

    def _put(self, name: str, bits: np.ndarray):
        off, shape = self.layout[name]
        assert tuple(bits.shape) == shape, (name, bits.shape, shape)
        src = torch.from_numpy(np.ascontiguousarray(bits).view(np.uint8).reshape(-1))
        self.arena[off:off + src.numel()].copy_(src, non_blocking=False)

    def load(self, w: Dict[str, np.ndarray]):
        """Fill the arena from a HF-named state dict (fp32 arrays or bf16 bit patterns)."""
        if self.arena is None:
            self.allocate()
        v, t = self.cfg.vision, self.cfg.text
        V, Lm = "model.visual.", "model.language_model."
        pe = _bits(w[V + "patch_embed.proj.weight"]).reshape(v.embed_dim, -1)
        pad = np.zeros((v.embed_dim, v.patch_dim_padded), np.uint16)
        pad[:, :pe.shape[1]] = pe
        self._put("vit.patch", pad)
        v25 = v.variant == "qwen2_5"
        for i in range(v.depth):
            s, d = f"{V}blocks.{i}.", f"vit.{i}."
            names = [("norm1.weight", "ln1.w"), ("attn.qkv.weight", "qkv.w"), ("attn.qkv.bias", "qkv.b"),
                     ("attn.proj.weight", "proj.w"), ("attn.proj.bias", "proj.b"), ("norm2.weight", "ln2.w")]
            if not v25:
                names += [("norm1.bias", "ln1.b"), ("norm2.bias", "ln2.b"), ("mlp.fc1.weight", "fc1.w"),
                          ("mlp.fc1.bias", "fc1.b"), ("mlp.fc2.weight", "fc2.w"), ("mlp.fc2.bias", "fc2.b")]
            for a, b in names:
                self._put(d + b, _bits(w[s + a]))
            if v25:
                ffv, fp, D = v.mlp_dim, v.mlp_dim_padded, v.embed_dim
                def padded(a, rows):            # zero rows up to the padded width
                    out = np.zeros((rows,) + a.shape[1:], np.uint16)
                    out[:a.shape[0]] = a
                    return out
                g = padded(_bits(w[s + "mlp.gate_proj.weight"]), fp).reshape(fp // 8, 8, D)
                u = padded(_bits(w[s + "mlp.up_proj.weight"]), fp).reshape(fp // 8, 8, D)
                self._put(d + "gate_up.w", np.stack([g, u], 1).reshape(2 * fp, D))
                gb = padded(_bits(w[s + "mlp.gate_proj.bias"]), fp).reshape(fp // 8, 8)
                ub = padded(_bits(w[s + "mlp.up_proj.bias"]), fp).reshape(fp // 8, 8)
                self._put(d + "gate_up.b", np.stack([gb, ub], 1).reshape(2 * fp))
                dw = np.zeros((D, fp), np.uint16)
                dw[:, :ffv] = _bits(w[s + "mlp.down_proj.weight"])
                self._put(d + "down.w", dw)
                self._put(d + "down.b", _bits(w[s + "mlp.down_proj.bias"]))
        merger = [("merger.ln_q.weight", "ln.w"), ("merger.mlp.0.weight", "fc1.w"), ("merger.mlp.0.bias", "fc1.b"),
                  ("merger.mlp.2.weight", "fc2.w"), ("merger.mlp.2.bias", "fc2.b")]
        if not v25:
            merger.append(("merger.ln_q.bias", "ln.b"))
        for a, b in merger:
            self._put("vit.merger." + b, _bits(w[V + a]))
        self._put("llm.embed", _bits(w[Lm + "embed_tokens.weight"]))
        ff = t.intermediate_size
        if ff % 8:
            raise KarantaHipError(f"intermediate_size {ff} must be a multiple of 8")
        fp8 = self.weight_dtype == "fp8"
        for i in range(t.num_layers):
            s, d = f"{Lm}layers.{i}.", f"llm.{i}."
            self._put(d + "ln1.w", _bits(w[s + "input_layernorm.weight"]))
            self._put(d + "ln2.w", _bits(w[s + "post_attention_layernorm.weight"]))
            self._put(d + "qkv.b", np.concatenate([_bits(w[s + f"self_attn.{n}_proj.bias"]) for n in "qkv"], 0))
            if fp8:
                # quantise every original matrix row-wise (scale = max|row| / 448), then fuse / interleave codes,
                # scales and the dequantised bf16 copy alike
                from .weights import as_f32, fp8_e4m3_to_f32, pack_w16x64_fp8, quantize_fp8_rows
                qs = {n: quantize_fp8_rows(as_f32(w[s + n + ".weight"])) for n in
                      ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj",
                       "mlp.up_proj", "mlp.down_proj")}
                il = lambda a, b: np.stack([a.reshape((ff // 8, 8) + a.shape[1:]), b.reshape((ff // 8, 8) + b.shape[1:])], 1) \
                    .reshape((2 * ff,) + a.shape[1:])
                fused = {"qkv": tuple(np.concatenate([qs[f"self_attn.{n}_proj"][k] for n in "qkv"], 0) for k in (0, 1)),
                         "o": qs["self_attn.o_proj"],
                         "gate_up": (il(qs["mlp.gate_proj"][0], qs["mlp.up_proj"][0]), il(qs["mlp.gate_proj"][1], qs["mlp.up_proj"][1])),
                         "down": qs["mlp.down_proj"]}
                for n, (q, sc) in fused.items():
                    off8, _ = self.layout[d + n + ".w8"]
                    src = torch.from_numpy(np.ascontiguousarray(pack_w16x64_fp8(q)).reshape(-1))
                    self.arena[off8:off8 + src.numel()].copy_(src, non_blocking=False)
                    offs, _ = self.layout[d + n + ".s"]
                    srcs = torch.from_numpy(np.ascontiguousarray(sc, np.float32).view(np.uint8).reshape(-1))
                    self.arena[offs:offs + srcs.numel()].copy_(srcs, non_blocking=False)
                    self._put(d + n + ".w", pack_w16x64(to_bf16_bits(fp8_e4m3_to_f32(q) * sc[:, None])))
                continue
            self._put(d + "qkv.w", pack_w16x64(np.concatenate([_bits(w[s + f"self_attn.{n}_proj.weight"]) for n in "qkv"], 0)))
            self._put(d + "o.w", pack_w16x64(_bits(w[s + "self_attn.o_proj.weight"])))
            g = _bits(w[s + "mlp.gate_proj.weight"]).reshape(ff // 8, 8, -1)   # 8-row interleave: KR_EPI_SILU_MUL8
            u = _bits(w[s + "mlp.up_proj.weight"]).reshape(ff // 8, 8, -1)
            self._put(d + "gate_up.w", pack_w16x64(np.stack([g, u], 1).reshape(2 * ff, -1)))
            self._put(d + "down.w", pack_w16x64(_bits(w[s + "mlp.down_proj.weight"])))
        self._put("llm.norm.w", _bits(w[Lm + "norm.weight"]))
        head = w[Lm + "embed_tokens.weight"] if (t.tie_word_embeddings or "lm_head.weight" not in w) else w["lm_head.weight"]
        self._put("llm.lm_head", pack_w16x64(_bits(head)))
        torch.cuda.synchronize(self.device)


# =============================================================================
# request / result types
# =============================================================================
@dataclass
class PageRequest:
    """One page = one sequence: prompt token ids (image placeholders included) + its images."""
    input_ids: np.ndarray                      # int64 [P]
    pixel_values: Optional[np.ndarray] = None  # fp32 [n_patches, 1176] (all images concatenated)
    grids: List[Tuple[int, int, int]] = field(default_factory=list)
    temperature: float = 0.0                   # 0: greedy; > 0: Gumbel-max sampling (kr_gumbel_argmax)
    seed: int = 0                              # the sampler is counter-based: (seed, token index) fixes every draw
    images: Optional[List[Any]] = None         # instead of pixel_values: HWC uint8 RGB pages for the GPU front end
    #                                            (numpy arrays, or torch uint8 tensors already resident in HBM)
    guide: Any = None                          # guided.Guide / DeviceGuide: the output must match this pattern
    logprobs: Optional[int] = None             # None: off; k >= 0: log-prob of every token + the k most probable (<= 20)


@dataclass
class GenerateResult:
    tokens: List[np.ndarray]          # per page: generated ids (EOS included, nothing after it)
    finish_reasons: List[str]         # "stop" | "length"
    prompt_tokens: List[int]
    timings: Dict[str, float]
    logits: Optional[np.ndarray] = None  # [B, steps, V] when return_logits
    logprobs: Optional[List[Optional[Dict[str, np.ndarray]]]] = None   # per page (None where not asked): "token" [n],
    #                                                                    "top_ids" [n, k], "top" [n, k]


class DeviceGuide:
    """A compiled pattern resident in HBM: DFA transitions [S, 256] uint16 and one allowed-token bit row per state
    (kr_guide_build_masks).  Built by :meth:`Engine.compile_guide`."""

    def __init__(self, guide, trans: torch.Tensor, masks: torch.Tensor):
        self.guide, self.trans, self.masks = guide, trans, masks
        self.start = int(guide.start)


class Engine:
    def __init__(self, cfg: ModelConfig, device: str = "cuda:0", max_batch: int = 8, s_max: int = 4096,
                 max_patches: int = 8 * 5476, max_prompt_tokens: int = 8 * 2048, decode_splits: int = 16,
                 weight_dtype: str = "bf16", fp8_activations: Optional[bool] = None, admission_cus: Optional[int] = None):
        if not torch.cuda.is_available():
            raise KarantaHipError("no HIP device: the karanta MI355X engine has no CPU fallback")
        self.L = lib()
        self.cfg = cfg
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.s = self.stream.cuda_stream
        self.B = max_batch
        if max_batch > 32:
            raise KarantaHipError("max_batch > 32: the decode kernels take at most two 16-row column tiles")
        self.s_max = _align(s_max, 64)
        self.max_patches = max_patches
        self.max_tokens = max_prompt_tokens
        self.n_split = decode_splits
        self._ignore_eos = self._freeze_finished = self._want_logits = self._sampling = False
        self._guided = False          # the decode graph masks logits by the slots' DFA states and advances them
        # what the current mode ALLOWS (begin_slots(sampling=, guided=); generate(): what its pages need): _sampling / _guided say
        # what the next decode step RUNS — in slot mode they follow the requests that are actually in the slots (set_step_features)
        self._cap_sampling = self._cap_guided = False
        self._logprobs = None         # None or k: the decode graph records log-probabilities (kr_logprobs_topk)
        self._adm_stream = None                          # second stream of the overlapped admission (slot mode)
        # overlapped admissions run on a stream restricted to this many compute units (kr_stream_create_cu_mask), so the
        # decode graph on the main stream always finds free CUs beside the admission's long-running workgroups; 0 / None: an
        # ordinary second stream (round 1: no gain, the streams serialise).  KARANTA_ADMIT_CUS overrides.
        env_cus = os.environ.get("KARANTA_ADMIT_CUS")
        self.admission_cus = int(env_cus) if env_cus is not None else (int(admission_cus) if admission_cus else 0)
        self._adm_stream_handle = None
        # ... and, while an admission is in flight on it, the decode graph replays on a stream restricted to the OTHER compute units
        # (kr_stream_create_cu_range): disjoint sets, so neither side waits for a workgroup slot of the other.  Measured alone
        # (tools/decode_ab.py --cus): a 32-row step keeps 94 / 70 / 47 % of its speed on 75 / 50 / 25 % of the CUs.
        self.disjoint_decode = os.environ.get("KARANTA_DISJOINT_DECODE", "1") != "0"
        self._dec_stream, self._dec_stream_handle, self._adm_inflight = None, None, 0
        self.n_cus = int(torch.cuda.get_device_properties(self.device).multi_processor_count)
        self._resample_cache: Dict[tuple, tuple] = {}   # (h, w, rh, rw) -> device tables of the GPU image front end
        self.persist_blocks = int(os.environ.get("KARANTA_PERSIST_BLOCKS", "512"))  # 2 persistent workgroups per CU (swept: 256..1024)
        if type(self) is Engine and any(os.environ.get(k, "0") not in ("", "0") for k in
                                        ("KARANTA_PREFETCH", "KARANTA_FAST_RESIDUAL", "KARANTA_ATTN_FUSED", "KARANTA_MERGE_IN_OPROJ")):
            raise KarantaHipError("KARANTA_PREFETCH / KARANTA_FAST_RESIDUAL / KARANTA_ATTN_FUSED / KARANTA_MERGE_IN_OPROJ are decode "
                                  "experiments: they run on csrc/tools/experiment_engine.ExperimentEngine with a -DKR_EXPERIMENTS "
                                  "library (csrc/tools/build_variant.py, KARANTA_HIP_LIB), not on the product engine")
        v, t = cfg.vision, cfg.text
        if v.head_dim not in (80, 128) or t.head_dim != 128:
            raise KarantaHipError(f"unsupported head dims vit={v.head_dim} llm={t.head_dim}")
        self.w = DeviceWeights(cfg, self.device, weight_dtype)
        self.fp8 = weight_dtype == "fp8"
        # fp8 engine: prefill GEMMs read the fp8 codes (kr_gemm_fp8); KARANTA_FP8_PREFILL=0: their dequantised bf16 copy
        self.fp8_prefill_gemm = os.environ.get("KARANTA_FP8_PREFILL", "1") == "1"
        # W8A8 prefill (fp8 engine): the activations entering the decoder's prefill GEMMs are quantised per token to e4m3
        # (kr_quantize_rows_fp8) and the products run on the fp8 matrix instruction (kr_gemm_fp8a) — what vLLM does for the
        # reference's OLMO_7B_0725_FP8.  Decode keeps bf16 activations (weight-only fp8).  KARANTA_FP8_ACT=0/1 overrides.
        env = os.environ.get("KARANTA_FP8_ACT")
        self.fp8_act = self.fp8 and self.fp8_prefill_gemm and (bool(fp8_activations) if fp8_activations is not None
                                                              else (env == "1" if env is not None else FP8_ACT_DEFAULT))
        self._graphs: Dict[Tuple[int, bool], int] = {}
        self._prof_on = False
        self._prof_events: List[Tuple[C.c_void_p, C.c_void_p]] = []
        self._prof_next = 0
        self._last_batch = max_batch
        self._alloc()

    # ------------------------------------------------------------------ buffers
    def _alloc(self):
        v, t, dev = self.cfg.vision, self.cfg.text, self.device
        N, M, B = self.max_patches, self.max_tokens, self.B
        z = lambda *shape, dtype=BF16: torch.zeros(*shape, dtype=dtype, device=dev)
        nvb = N // 64 + 64  # V^T blocks: every image may add one partial block
        if v.variant == "qwen2_5":  # windowed blocks: every window starts a V^T block, edge windows are partial ones
            per_win = (v.window_merge_units * v.spatial_merge_size) ** 2
            nvb = max(nvb, 2 * (N // max(1, min(64, per_win))) + 64)
        # ViT
        self.v_pix = z(N, v.patch_dim, dtype=torch.float32)
        self.v_in = z(N, v.patch_dim_padded)
        self.v_x = z(N, v.embed_dim)
        self.v_h = z(N, v.embed_dim)
        self.v_qkv = z(N, 3 * v.embed_dim)
        self.v_q = z(v.num_heads, N, v.head_dim)
        self.v_k = z(v.num_heads, N, v.head_dim)
        self.v_vt = z(v.num_heads, nvb, v.head_dim, 64)
        self.v_o = z(N, v.embed_dim)
        self.v_f = z(N, v.mlp_dim_padded)
        self.v_perm = None  # Qwen2.5-VL: window-order gather indices live in the per-geometry cache
        self.v_m1 = z(N // 4 + 1, v.merge_dim)
        self._vit_cache = {}
        self._vit_rot_cache = {}   # (t, h, w) -> (cos, sin) of one image, resident in HBM (_vit_tables)
        self.img_embeds = z(N // 4 + 1, t.hidden_size)
        # decoder prefill
        self.p_x = z(M, t.hidden_size)
        self.p_h = z(M, t.hidden_size)
        self.p_qkv = z(M, t.qkv_dim)
        self.p_q = z(t.num_heads, M, t.head_dim)
        self.p_o = z(M, t.q_dim)
        self.p_act = z(M, t.intermediate_size)
        self.p_cos = z(M, t.head_dim, dtype=torch.float32)
        self.p_sin = z(M, t.head_dim, dtype=torch.float32)
        self.p_src = z(M, dtype=torch.int32)
        if self.fp8_act:   # W8A8 prefill: the quantised A operand (row stride = the widest K) and its per-token scales
            self.p_a8 = z(M, (max(t.intermediate_size, t.hidden_size, t.q_dim) + 15) // 16 * 16, dtype=torch.uint8)
            self.p_as = z(M, dtype=torch.float32)
        self.gemm_scratch = z(GEMM_SCRATCH_BYTES // 4, dtype=torch.float32)   # kr_gemm_bf16_ws (KR_GEMM_SCRATCH_BYTES)
        # KV cache (zero-initialised: masked keys must be finite)
        self.kcache = z(t.num_layers, B, t.num_kv_heads, self.s_max, t.head_dim)
        self.vtcache = z(t.num_layers, B, t.num_kv_heads, self.s_max // 64, t.head_dim, 64)
        # decode state
        self.d_x = z(B, t.hidden_size)
        self.d_x2 = z(B, t.hidden_size)  # the other residual buffer (deferred split-K ping-pong)
        # down_proj slabs of the deferred split: [2][B][hidden]; > 16 rows with the group-split down_proj: two PAIRS (layer parity)
        self.d_part = z(4 if B > 16 else 2, B, t.hidden_size, dtype=torch.float32)
        # ONE-slab form of the deferred split (KARANTA_ATOMIC_SLAB, default on): down_proj's two K ranges add into one f32
        # accumulator with float atomics (a + b onto zero: order-free, reproducible); two accumulators alternate by layer
        # parity, layer L's qkv launch zeroes the one layer L's down_proj adds into.  d_part doubles as the pair.
        self.atomic_slab = os.environ.get("KARANTA_ATOMIC_SLAB", "1") == "1"
        # down_proj at >= 192 weight tiles (7B widths) runs two tiles per workgroup (kr_decode.hip, launch_narrow_direct):
        # 8-wave workgroups with a 5-deep ring then beat 16-wave ones (7B step 2.992 -> 2.975 ms)
        if t.hidden_size // 16 >= 192:
            self.down_waves_small = 8
        self.d_qkv = z(B, t.qkv_dim)
        # batches above 16 rows keep the inputs of their decode linears in the PACKED layout of kr_linear_decode32 (32 row
        # slots whatever the batch): d_h (normalised rows of kr_decode_resnorm32), d_o (merged heads), d_act (SiLU * up)
        Bp = 32 if B > 16 else B
        self.d_h = z(Bp, t.hidden_size)
        self.d_q = z(B, t.num_heads, t.head_dim)
        self.d_o = z(Bp, t.q_dim)
        self.d_act = z(Bp, t.intermediate_size)
        self.d_logits = z(B, t.vocab_size, dtype=torch.float32)
        self.d_ws = z(B * t.num_heads * self.n_split * (t.head_dim + 4), dtype=torch.float32)
        # waves per workgroup of the narrow decode linears: enough waves that every wave still
        # streams >= 2 K-chunks, no cross-workgroup reduction (each fence/atomic hop costs microseconds)
        self.wv_qkv = self._waves(t.hidden_size // 64)
        self.wv_o = self._waves(t.q_dim // 64)
        self.wv_down = self._waves(t.intermediate_size // 64)
        self.wv_wide = 4
        for name in ("qkv", "o", "down", "wide"):  # tuning overrides (sweeps): KARANTA_WV_DOWN=8 ...
            v_ = os.environ.get("KARANTA_WV_" + name.upper())
            if v_:
                setattr(self, "wv_" + name, int(v_))
        # qkv / o_proj / down_proj: kr_linear_decode_narrow; down_proj split over 2 workgroups per tile with the
        # reduction deferred to the next layer's qkv prologue (K = 1536 / 2048 / 3584 only)
        self.narrow_mode = os.environ.get("KARANTA_NARROW", "1") == "1"
        self.narrow_o = self.narrow_mode and os.environ.get("KARANTA_NARROW_O", "1") == "1"
        self.defer_down = (self.narrow_mode and os.environ.get("KARANTA_DEFER_DOWN", "1") == "1"
                           and t.hidden_size in (1536, 2048, 3584))
        # gate/up and lm_head: one wave per 16-row tile (kr_linear_decode_wide) when K allows it
        self.wide_mode = os.environ.get("KARANTA_WIDE", "1") == "1" and t.hidden_size % 512 == 0 and t.hidden_size <= 4096
        self.wide_blocks = int(os.environ.get("KARANTA_WIDE_BLOCKS", "256"))
        self.wide_waves = int(os.environ.get("KARANTA_WIDE_WAVES", "0"))  # 0: ceil(tiles / blocks), at most 8
        self.wide_spread32 = os.environ.get("KARANTA_WIDE_SPREAD32", "1") == "1"   # > 16 rows: tiles dealt as at <= 16 rows (_wide_geometry)
        if self.B > 16 and not (self.wide_mode and self.narrow_mode and self.narrow_o and t.intermediate_size % 64 == 0):
            raise KarantaHipError("max_batch > 16 needs the wide / narrow decode kernels (hidden_size % 512 == 0) and hidden_size <= 2048 or == 3584")
        # Above 16 rows the decode linears hold two 16-row column tiles per weight fragment, with all 32 normalised x rows
        # in LDS — which fits up to hidden_size 2048 (Qwen2-VL-2B, Qwen2.5-VL-3B).  At the 7B width (32 x 3584 bf16 = 229 KB
        # against 160 KB of LDS) the wide launches (gate/up, lm_head) stage K in two halves (dec_wide_kh_kernel: weights
        # still read once) and the qkv launch, whose norm prologue keeps whole rows per wave, runs ONCE PER 16-ROW RANGE
        # (33 MB of 14 GB streamed twice); o_proj / down_proj read their x fragments straight from L2 on two column tiles.
        self.row_split = self.B > 16 and t.hidden_size > 2048
        # > 16 rows: kr_decode_resnorm + ONE direct qkv launch.  On where the fused launch would have to run per 16-row range
        # (the 7B width: 4.73 -> 4.47 ms per step); at widths whose 32 rows fit the fused launch's LDS it is 1 % slower
        # (Qwen2-VL-2B, same-box A/B: 1.957 vs 1.979 ms) and stays off.  KARANTA_RESNORM_QKV = 0 / 1 forces either.
        env_rn = os.environ.get("KARANTA_RESNORM_QKV")
        self.resnorm_qkv = (env_rn == "1") if env_rn is not None else self.row_split
        # > 16 rows: the packed-activation family (kr_linear_decode32) for qkv (behind kr_decode_resnorm32), o_proj and
        # down_proj; KARANTA_DEC32=0: round 3's two-column-tile instantiations of the narrow kernel (row-major x fragments)
        # ... with kr_decode_resnorm32 + the packed qkv launch also at the widths whose fused qkv launch fits (KARANTA_RESNORM32_QKV)
        self.resnorm32_qkv = os.environ.get("KARANTA_RESNORM32_QKV", "1") == "1"
        self.group_split_down = int(os.environ.get("KARANTA_DOWN_GS", "1"))   # 0 off, 1 at 16-atom partitions (2B widths), 2 always
        self.family32 = (self.B > 16 and self.narrow_mode and self.narrow_o and t.intermediate_size % 64 == 0 and t.q_dim % 64 == 0
                         and t.hidden_size % 64 == 0 and os.environ.get("KARANTA_DEC32", "1") == "1")
        if self.row_split and t.hidden_size != 3584:
            raise KarantaHipError("max_batch > 16: hidden_size <= 2048 or == 3584 (the 7B width) only")
        if self.fp8 and not (self.wide_mode and self.narrow_mode):
            raise KarantaHipError("fp8 weights need the wide / narrow decode kernels: hidden_size % 512 == 0 and <= 4096")
        # one argmax partial per wave of the lm_head launch; the launch geometry depends on the row count (8 waves above
        # 16 rows), so the buffers are sized for the larger one and the sampler is told the launch's own count
        self.n_amax = (max(self._amax_parts(0), self._amax_parts(32)) if self.wide_mode
                       else (t.vocab_size // 16 + 1) // 2)
        self.d_amax_v = z(B, self.n_amax, dtype=torch.float32)
        self.d_amax_i = z(B, self.n_amax, dtype=torch.int32)
        self.d_plen = z(B, dtype=torch.int32)
        self.d_cs = None  # [B][max_new][128] rotary table of the decode positions, built per request
        self.d_ctx = z(B, dtype=torch.int32)
        self.d_delta = z(B, dtype=torch.int32)
        self.d_tok = z(B, dtype=torch.int32)
        self.d_fin = z(B, dtype=torch.int32)
        self._snap_ring, self._snap_next, self._snap_event, self._copy_stream = None, 0, None, None   # snapshot_slots()
        self.d_temp = z(B, dtype=torch.float32)   # per-slot sampling temperature and seed
        self.d_seed = z(B, dtype=torch.int32)
        # guided decoding: per slot the device addresses of its pattern's tables (0 = unconstrained) + its DFA state
        self.d_gtrans = z(B, dtype=torch.int64)
        self.d_gmasks = z(B, dtype=torch.int64)
        self.d_gstate = z(B, dtype=torch.int32)
        self.mask_words = 2 * ((t.vocab_size + 63) // 64)
        self.d_voc_off = self.d_voc_bytes = None   # set_vocab()
        self._guides: Dict[str, DeviceGuide] = {}
        self._slot_guides: Dict[int, DeviceGuide] = {}   # keeps the tables of the running requests alive
        # log-probabilities: [hist][B][1 + 20] / [hist][B][20], allocated with the token history when asked for
        self.lp_part = max(64 if t.vocab_size > 4096 else 1, -(-t.vocab_size // 4096))   # slices of <= 4096 logits
        self.d_lp = self.d_lpi = None
        self.d_lp_pv = z(B, self.lp_part, 20, dtype=torch.float32)
        self.d_lp_pi = z(B, self.lp_part, 20, dtype=torch.int32)
        self.d_lp_ms = z(B, self.lp_part, 2, dtype=torch.float32)
        self.max_new = 0          # size of the token history / rotary tables (only grows: graphs hold their addresses)
        self._req_max_new = 0     # what the current generate() / begin_slots() asked for (capacity checks use this)
        self.d_hist = None
        self.d_eos = torch.tensor(list(self.cfg.eos_token_ids), dtype=torch.int32, device=dev)
        self.d_invfreq = torch.from_numpy(POS.rope_inv_freq(t.head_dim, t.rope_theta)).to(dev)
        self.d_last = z(B, dtype=torch.int32)
        torch.cuda.synchronize(dev)

    @staticmethod
    def _waves(nchunks: int) -> int:
        return 16 if nchunks >= 64 else 8 if nchunks >= 16 else 4

    def load_weights(self, weights: Dict[str, np.ndarray]):
        self.w.load(weights)

    # ------------------------------------------------------------------ small launch helpers
    def _gemm(self, A, W, C_, M, bias=None, res=None, epi=EPI_NONE, packed=False, w8=None, w_scale=None):
        N, K = W.shape
        if w8 is not None and self.fp8_act:
            # W8A8: per-token e4m3 codes of A (scale = max|row| / 448), both operands through v_mfma_f32_16x16x32_fp8_fp8
            ldq = self.p_a8.stride(0)
            self.L.kr_quantize_rows_fp8(ptr(A), A.stride(0), ptr(self.p_a8), ldq, ptr(self.p_as), M, K, self.s)
            self.L.kr_gemm_fp8a(ptr(self.p_a8), ldq, ptr(self.p_as), ptr(w8), ptr(w_scale), ptr(bias), ptr(res),
                                res.stride(0) if res is not None else 0, ptr(C_), C_.stride(0), M, N, K, epi, self.s)
            return
        if w8 is not None and self.fp8_prefill_gemm:
            # fp8 engine: the prefill streams the SAME e4m3 codes + row scales the decode kernels read (kr_gemm_fp8)
            self.L.kr_gemm_fp8(ptr(A), A.stride(0), ptr(w8), ptr(w_scale), ptr(bias), ptr(res),
                               res.stride(0) if res is not None else 0, ptr(C_), C_.stride(0), M, N, K, epi, self.s)
            return
        # split-K scratch of long-K tail rounds: the engine's own buffer (its GEMMs never run on two streams at once)
        self.L.kr_gemm_bf16_ws(ptr(A), A.stride(0), ptr(W), ptr(bias), ptr(res), res.stride(0) if res is not None else 0,
                               ptr(C_), C_.stride(0), M, N, K, epi, 1 if packed else 0, ptr(self.gemm_scratch),
                               self.gemm_scratch.numel() * 4, self.s)

    def _dec(self, mode, x, W, M, out=None, out_f32=None, bias=None, norm_w=None, res=None, waves=4, kc=0, vc=0,
             attn_partials=None):
        """kr_linear_decode on packed weights (no cross-workgroup split on the engine's path)."""
        t = self.cfg.text
        N, K = W.shape
        o = out if out is not None else out_f32
        self.L.kr_linear_decode(mode, ptr(x), x.stride(0) if x is not None else 0, ptr(W), ptr(bias), ptr(norm_w),
                                t.rms_norm_eps, ptr(res), res.stride(0) if res is not None else 0, ptr(out), ptr(out_f32),
                                o.stride(0) if o is not None else 0, M, N, K, waves, self.persist_blocks, 1, 0, 0,
                                ptr(attn_partials), self.n_split, ptr(self.d_cs), self.max_new, ptr(self.d_plen),
                                ptr(self.d_ctx), ptr(self.d_q), kc, vc, t.num_heads, t.num_kv_heads, self.s_max,
                                ptr(self.d_amax_v), ptr(self.d_amax_i), self.s)

    def _dec_narrow(self, mode, x, W, M, out=None, out_f32=None, bias=None, norm_w=None, res=None, waves=8, ksplit=1,
                    part_in=None, x_out=None, kc=0, vc=0, w8=None, w_scale=None, part_rows=0, row0=0,
                    zero=None, atomic_out=False, **experiment):
        """kr_linear_decode_narrow: one workgroup per tile (pair); ksplit > 1 = deferred split-K slabs in out_f32.
        w8 / w_scale: the fp8 copy of W and its row scales (kr_linear_decode_narrow_fp8).  zero (an f32 tensor the launch
        also zeroes), atomic_out, part_rows: the launch's kr_narrow_opts."""
        t = self.cfg.text
        N, K = W.shape
        o = out if out is not None else out_f32
        ldc = o.stride(-2) if o is not None else 0  # slabs of the deferred split are [ksplit][M][ldc] with the CURRENT M, packed in d_part
        head = (mode, ptr(x), x.stride(0), ptr(part_in), int(part_in.shape[0]) if part_in is not None else 0, ptr(x_out),
                x_out.stride(0) if x_out is not None else 0)
        opts = narrow_opts(ptr(zero) if zero is not None else 0, zero.numel() * 4 if zero is not None else 0, atomic_out, part_rows)
        # row0: the launch covers batch rows row0 .. row0 + M - 1 (every per-sequence array is handed over from that row)
        tail = (ptr(bias), ptr(norm_w), t.rms_norm_eps, ptr(res), res.stride(0) if res is not None else 0, ptr(out),
                ptr(out_f32), ldc, M, N, K, waves, ksplit, ptr(self.d_cs[row0:]) if self.d_cs is not None else 0, self.max_new,
                ptr(self.d_plen[row0:]), ptr(self.d_ctx[row0:]), ptr(self.d_q[row0:]),
                kc + 2 * row0 * t.num_kv_heads * self.s_max * t.head_dim if kc else 0,
                vc + 2 * row0 * t.num_kv_heads * self.s_max * t.head_dim if vc else 0, t.num_heads, t.num_kv_heads, self.s_max, opts)
        if experiment:      # csrc/tools/experiment_engine.py (x_out_f32, prefetch: kr_linear_decode_narrow_x32)
            self._dec_narrow_experiment(head, tail, W, w8, w_scale, **experiment)
        elif w8 is not None:
            self.L.kr_linear_decode_narrow_fp8(*head, ptr(w8), ptr(w_scale), *tail, self.s)
        else:
            self.L.kr_linear_decode_narrow(*head, ptr(W), *tail, self.s)

    def _dec32(self, mode, xp, W, M, waves_ref, out=None, out_f32=None, bias=None, res=None, ksplit=1, atomic_out=False, zero=None,
               kc=0, vc=0, w8=None, w_scale=None, tiles_per_wg=0, group_split=False):
        """kr_linear_decode32: the decode linears of a 17..32-row batch on PACKED activations (xp: kr_pack_rows32 layout, written
        by kr_decode_resnorm32 / kr_attn_decode_merge32 / the gate/up launch with DEC_OUT_XP), with the K partition of the
        <= 16-row launch of the same layer (waves_ref, ksplit): row for row the bits that launch produces."""
        t = self.cfg.text
        N, K = W.shape
        o = out if out is not None else out_f32
        a = Dec32(ptr(xp), ptr(w8 if w8 is not None else W), ptr(w_scale), ptr(bias), ptr(res), res.stride(0) if res is not None else 0,
                  ptr(out), ptr(out_f32), o.stride(-2) if o is not None else 0, M, N, K, waves_ref, ksplit, 1 if atomic_out else 0,
                  tiles_per_wg, 1 if group_split else 0, 0, ptr(zero) if zero is not None else None,
                  zero.numel() * 4 if zero is not None else 0,
                  ptr(self.d_cs), self.max_new, ptr(self.d_plen), ptr(self.d_ctx), ptr(self.d_q), kc or None, vc or None,
                  t.num_heads, t.num_kv_heads, self.s_max)
        self.L.kr_linear_decode32(mode, C.byref(a), self.s)

    down_waves_small = 16         # waves per down_proj workgroup at <= 16 rows (instance attribute for sweeps)
    down_gs_tiles = 0             # weight tiles per workgroup of the group-split down_proj (0: 4 at 8-atom partitions, else 2)
    o_waves = 8                   # waves per o_proj workgroup at <= 16 rows

    def _down_waves(self, B: int) -> int:
        return self.down_waves_small if B <= 16 else 8   # two batch column tiles double the x fragments: 8-wave workgroups only

    def _w8kw(self, name: str) -> dict:
        w8, sc = self._w8(name)
        return {} if w8 is None else {"w8": w8, "w_scale": sc}

    def _w8(self, name: str):
        """(fp8 codes, row scales) of a decoder Linear when the engine runs on fp8 weights, else (None, None)."""
        if self.fp8 and self.w.has(name + "8"):
            return self.w.view_u8(name + "8"), self.w.view_f32(name[:-1] + "s")
        return None, None

    def _wide_geometry(self, N: int, M: int = 0):
        """(workgroups, waves) of a wide launch.  No idle waves: W = tiles per CU (at most 8), then just enough
        workgroups for one tile per wave (gate/up of the 2B decoder: 1120 tiles = 224 workgroups x 5 waves);
        beyond that the waves loop over their tiles.  More than 16 rows: 8 waves, so that the branch-free prologue
        (4 rows per wave) stages all 32 rows — with 5 waves 12 rows went the slow way, behind the weight ring
        (gate/up at B = 32: 16.7 -> 14.7 us)."""
        tiles = N // 16
        if M > 16 and not self.wide_waves and self.wide_spread32:
            # r4: 8 waves stage the 32 rows, but the TILES are dealt as at <= 16 rows — to as many workgroups as there are compute
            # units to pull them (2B gate/up: 1120 tiles = 224 workgroups whose waves 0..4 own one tile each and waves 5..7 only
            # help with the prologue; 140 x 8 left 116 compute units without a weight stream)
            own = min(8, -(-tiles // self.wide_blocks))
            return min(self.wide_blocks, -(-tiles // own)), 8
        waves = self.wide_waves or (8 if M > 16 else min(8, -(-tiles // self.wide_blocks)))
        return min(self.wide_blocks, -(-tiles // waves)), waves

    def _row_ranges(self, B: int):
        """(first row, rows) of the qkv launches: the whole batch, or 16-row ranges when 32 x rows do not fit (row_split)."""
        if self.row_split and B > 16:
            return [(0, 16), (16, B - 16)]
        return [(0, B)]

    def _amax_parts(self, M: int) -> int:
        """Argmax partials the lm_head launch writes per row at batch M (= its stride in d_amax_*): workgroups x waves."""
        wb, ww = self._wide_geometry(self.cfg.text.vocab_size, M)
        return wb * ww

    def _dec_wide(self, mode, x, W, M, out=None, out_f32=None, norm_w=None, w8=None, w_scale=None, amax_row0: int = 0):
        """kr_linear_decode_wide: `wide_blocks` workgroups (one per CU), each wave an independent weight stream."""
        N, K = W.shape
        blocks, waves = self._wide_geometry(N, M)
        o = out if out is not None else out_f32
        av, ai = self.d_amax_v.view(-1)[amax_row0 * blocks * waves:], self.d_amax_i.view(-1)[amax_row0 * blocks * waves:]
        tail = (0, ptr(norm_w), self.cfg.text.rms_norm_eps, 0, 0, ptr(out), ptr(out_f32), o.stride(0) if o is not None else 0,
                M, N, K, blocks, waves, ptr(av), ptr(ai), self.s)
        if w8 is not None:
            self.L.kr_linear_decode_wide_fp8(mode, ptr(x), x.stride(0), ptr(w8), ptr(w_scale), *tail)
        else:
            self.L.kr_linear_decode_wide(mode, ptr(x), x.stride(0), ptr(W), *tail)

    def _h2d(self, dst: torch.Tensor, arr: np.ndarray):
        src = torch.from_numpy(np.ascontiguousarray(arr))
        dst.view(-1)[:src.numel()].copy_(src.view(-1), non_blocking=False)

    # ------------------------------------------------------------------ vision tower
    def _vit_tables(self, grids):
        """Rotary tables and attention work lists depend only on the image grids: build once per
        distinct batch geometry and keep them resident in HBM."""
        key = tuple(tuple(int(x) for x in g) for g in grids)
        hit = self._vit_cache.get(key)
        if hit is None:
            v, dev = self.cfg.vision, self.device
            plan = POS.vit_attn_plan(key)
            t_ = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
            # the rotary tables of an image depend on its own grid only: kept per grid, so that a batch of a new COMPOSITION (every
            # admission of a corpus with mixed page sizes) costs a concatenation on the device, not ~0.7 ms of numpy per page with the
            # GPU waiting (the work lists below are a few hundred integers)
            per_img = []
            for g in key:
                hit_i = self._vit_rot_cache.get(g)
                if hit_i is None:
                    c_i, s_i = POS.vision_rotary_tables((g,), v.head_dim, v.spatial_merge_size)
                    hit_i = (t_(c_i), t_(s_i))
                    if len(self._vit_rot_cache) > 64:
                        self._vit_rot_cache.clear()
                    self._vit_rot_cache[g] = hit_i
                per_img.append(hit_i)
            cos = per_img[0][0] if len(per_img) == 1 else torch.cat([c for c, _ in per_img])
            sin = per_img[0][1] if len(per_img) == 1 else torch.cat([s_ for _, s_ in per_img])
            dplan = lambda pl: (pl, t_(pl.blk_tok0), t_(pl.blk_ntok), t_(pl.blk_k_row0), t_(pl.blk_vt_blk), t_(pl.qblk),
                                t_(pl.qblk_len))
            extra = None
            if v.variant == "qwen2_5":
                # window order (TF25:430-446): patches move in groups of merge^2; rotary tables move with them;
                # two attention work lists: windows, and whole images for the fullatt_block_indexes blocks
                unit = v.spatial_merge_size ** 2
                order, win_lens = POS.vision_window_order(key, v.spatial_merge_size, v.window_size, v.patch_size)
                perm = (order[:, None] * unit + np.arange(unit)[None, :]).reshape(-1)      # patch-level gather
                perm_long = t_(perm.astype(np.int64))
                cos, sin = cos.index_select(0, perm_long), sin.index_select(0, perm_long)
                extra = (t_(perm.astype(np.int32)), t_(np.argsort(order).astype(np.int32)), dplan(POS.segments_attn_plan(win_lens)))
            hit = (plan, cos, sin, t_(plan.blk_tok0), t_(plan.blk_ntok), t_(plan.blk_k_row0),
                   t_(plan.blk_vt_blk), t_(plan.qblk), t_(plan.qblk_len), extra)
            if len(self._vit_cache) > 16:
                self._vit_cache.clear()
            self._vit_cache[key] = hit
        return hit

    # ------------------------------------------------------------------ GPU image front end
    def patches_from_images(self, images: Sequence[np.ndarray], min_pixels: int = IP.MIN_PIXELS,
                            max_pixels: int = IP.MAX_PIXELS_CLASS_DEFAULT,
                            grids: Optional[Sequence[Sequence[int]]] = None):
        """HWC uint8 RGB pages -> (pixel_values fp32 [n, 1176] resident in HBM, grids): smart_resize on the host
        (integers), PIL-identical bicubic resize, normalisation and patch order on the GPU
        (kr_image_resize_bicubic_u8 / kr_image_normalize_patchify).  Same numbers as
        image_processing.image_to_patches, bit for bit, without the host resample and with 3 bytes per pixel
        crossing PCIe instead of 2 x 1176 floats per patch.  ``grids`` (one (1, gh, gw) per image): resize to exactly
        gh x gw patches — what the prompt's placeholders were counted for — instead of running smart_resize here."""
        v, L, s, dev = self.cfg.vision, self.L, self.s, self.device
        unit = v.patch_size * v.spatial_merge_size
        metas, total = [], 0
        if grids is not None and len(grids) != len(images):
            raise KarantaHipError(f"{len(images)} images but {len(grids)} grids")
        for k, im in enumerate(images):
            if isinstance(im, torch.Tensor):       # a page already resident in HBM (uint8 HWC): no copy at all
                if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3 or not im.is_contiguous():
                    raise KarantaHipError("device images must be contiguous HWC uint8 RGB tensors")
            else:
                im = np.asarray(im)
                if im.ndim == 2:
                    im = np.stack([im] * 3, axis=-1)
                if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
                    raise KarantaHipError("images must be HWC uint8 RGB arrays")
                im = np.ascontiguousarray(im)
            h, w = int(im.shape[0]), int(im.shape[1])
            if grids is not None:
                g = [int(x) for x in grids[k]]
                if g[0] != 1 or g[1] % v.spatial_merge_size or g[2] % v.spatial_merge_size or min(g[1:]) < 1:
                    raise KarantaHipError(f"image grid {tuple(g)} is not (1, even, even)")
                rh, rw = g[1] * v.patch_size, g[2] * v.patch_size
            else:
                rh, rw = IP.smart_resize(h, w, unit, min_pixels, max_pixels)
            metas.append((im, h, w, rh, rw))
            total += (rh // v.patch_size) * (rw // v.patch_size)
        out = torch.empty(total, v.patch_dim, dtype=torch.float32, device=dev)
        mean = (C.c_float * 3)(*[float(x) for x in IP.CLIP_MEAN])
        std = (C.c_float * 3)(*[float(x) for x in IP.CLIP_STD])
        grids, off = [], 0
        with torch.cuda.stream(self.stream):
            for im, h, w, rh, rw in metas:
                if isinstance(im, torch.Tensor):
                    src = im if im.device == dev else im.to(dev)
                else:
                    src = torch.from_numpy(im if im.flags.writeable else im.copy()).to(dev)   # PIL-backed arrays are read-only
                key = (h, w, rh, rw)
                tabs = self._resample_cache.get(key)
                if tabs is None:
                    t_ = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
                    hb, hk = IP.resample_tables(w, rw) if rw != w else (None, None)
                    vb, vk = IP.resample_tables(h, rh) if rh != h else (None, None)
                    tabs = tuple(None if a is None else t_(a) for a in (hb, hk, vb, vk))
                    if len(self._resample_cache) > 64:
                        self._resample_cache.clear()
                    self._resample_cache[key] = tabs
                hb, hk, vb, vk = tabs
                dst = torch.empty(rh, rw, 3, dtype=torch.uint8, device=dev)
                tmp = torch.empty(h, rw, 3, dtype=torch.uint8, device=dev) if (rw != w and rh != h) else None
                L.kr_image_resize_bicubic_u8(ptr(src), h, w, ptr(dst), rh, rw, ptr(tmp), ptr(hb), ptr(hk),
                                             hk.shape[1] if hk is not None else 0, ptr(vb), ptr(vk),
                                             vk.shape[1] if vk is not None else 0, s)
                n = (rh // v.patch_size) * (rw // v.patch_size)
                L.kr_image_normalize_patchify(ptr(dst), rh, rw, mean, std, v.patch_size, v.spatial_merge_size,
                                              v.temporal_patch_size, ptr(out[off:]), s)
                grids.append((1, rh // v.patch_size, rw // v.patch_size))
                off += n
        return out, grids

    def _pixels_for(self, pages: Sequence["PageRequest"], pixel_values_device):
        """The pixel source of a batch of pages: caller-resident patches, the GPU front end (pages with `images`),
        or the pages' host arrays."""
        if pixel_values_device is not None:
            return pixel_values_device
        with_images = [p for p in pages if p.images]
        if with_images:
            if any(p.pixel_values is not None and len(p.pixel_values) for p in pages):
                raise KarantaHipError("a batch mixes pages with `images` and pages with `pixel_values`")
            for p in pages:
                if len(p.images or []) != len(p.grids):
                    raise KarantaHipError(f"a page has {len(p.images or [])} images but {len(p.grids)} grids")
            pix, _ = self.patches_from_images([im for p in pages for im in (p.images or [])],
                                              grids=[g for p in pages for g in p.grids])
            return pix
        pvs = [p.pixel_values for p in pages if p.pixel_values is not None and len(p.pixel_values)]
        if not pvs:
            return None
        return np.concatenate(pvs, 0) if len(pvs) > 1 else pvs[0]

    def vit_forward(self, pixel_values, grids: Sequence[Sequence[int]]) -> torch.Tensor:
        """Qwen2VisionTransformerPretrainedModel.forward (TF:700-731).  ``pixel_values`` is fp32
        ``[n, 1176]`` — a numpy array (copied to the device here) or a torch tensor already resident
        in HBM.  Returns a view of the merged image embeddings ``[T, d]`` (bf16, device)."""
        v, L, s, w = self.cfg.vision, self.L, self.s, self.w
        n = int(pixel_values.shape[0])
        if n == 0:
            return self.img_embeds[:0]
        if n > self.max_patches:
            raise KarantaHipError(f"{n} patches > max_patches {self.max_patches}")
        with torch.cuda.stream(self.stream):
            plan, cos_d, sin_d, blk_tok0, blk_ntok, blk_kr, blk_vb, qblk, qlen, extra = self._vit_tables(grids)
            assert plan.n_tokens == n, (plan.n_tokens, n)
            if isinstance(pixel_values, torch.Tensor):
                pix = pixel_values
                if pix.dtype != torch.float32 or not pix.is_cuda or not pix.is_contiguous():
                    raise KarantaHipError("device pixel_values must be a contiguous fp32 CUDA tensor")
            else:
                self._h2d(self.v_pix, np.asarray(pixel_values, dtype=np.float32))
                pix = self.v_pix
            D, H, hd = v.embed_dim, v.num_heads, v.head_dim
            L.kr_cast_pad_f32_bf16(ptr(pix), ptr(self.v_in), n, v.patch_dim, v.patch_dim_padded, s)
            self._gemm(self.v_in, w.view("vit.patch"), self.v_x, n)
            nvb_total = self.v_vt.shape[1]
            if plan.n_vt_blocks > nvb_total or (extra is not None and extra[2][0].n_vt_blocks > nvb_total):
                raise KarantaHipError("too many image segments for the V^T buffer")
            if v.variant == "qwen2_5":
                return self._vit_blocks_qwen2_5(n, (plan, blk_tok0, blk_ntok, blk_kr, blk_vb, qblk, qlen), cos_d, sin_d, extra)
            for i in range(v.depth):
                p = f"vit.{i}."
                L.kr_layernorm(ptr(self.v_x), ptr(w.view(p + "ln1.w")), ptr(w.view(p + "ln1.b")), ptr(self.v_h), n, D, 1e-6, s)
                self._gemm(self.v_h, w.view(p + "qkv.w"), self.v_qkv, n, bias=w.view(p + "qkv.b"))
                L.kr_qkv_prep(ptr(self.v_qkv), 3 * D, 0, D, 2 * D, ptr(cos_d), ptr(sin_d),
                              ptr(blk_tok0), ptr(blk_ntok), ptr(blk_kr), ptr(blk_vb), len(plan.blk_tok0),
                              ptr(self.v_q), self.v_q.stride(0), ptr(self.v_k), self.v_k.stride(0),
                              ptr(self.v_vt), self.v_vt.stride(0), H, H, hd, s)
                L.kr_attn_varlen_q(ptr(self.v_q), ptr(self.v_k), ptr(self.v_vt), ptr(self.v_o), ptr(qblk), ptr(qlen),
                                   plan.qblk.shape[0], self.v_q.shape[1], H, H, hd, self.v_k.stride(0),
                                   self.v_vt.stride(0), hd ** -0.5, 0, plan.q_block, s)
                self._gemm(self.v_o, w.view(p + "proj.w"), self.v_x, n, bias=w.view(p + "proj.b"), res=self.v_x)
                L.kr_layernorm(ptr(self.v_x), ptr(w.view(p + "ln2.w")), ptr(w.view(p + "ln2.b")), ptr(self.v_h), n, D, 1e-6, s)
                self._gemm(self.v_h, w.view(p + "fc1.w"), self.v_f, n, bias=w.view(p + "fc1.b"), epi=EPI_QUICK_GELU)
                self._gemm(self.v_f, w.view(p + "fc2.w"), self.v_x, n, bias=w.view(p + "fc2.b"), res=self.v_x)
            # PatchMerger (TF:277-290): LN -> view [n/4, 4D] -> Linear+GELU -> Linear
            L.kr_layernorm(ptr(self.v_x), ptr(w.view("vit.merger.ln.w")), ptr(w.view("vit.merger.ln.b")), ptr(self.v_h), n, D, 1e-6, s)
            T = n // (v.spatial_merge_size ** 2)
            merged_in = self.v_h.view(-1)[: T * v.merge_dim].view(T, v.merge_dim)
            self._gemm(merged_in, w.view("vit.merger.fc1.w"), self.v_m1, T, bias=w.view("vit.merger.fc1.b"), epi=EPI_GELU_ERF)
            self._gemm(self.v_m1, w.view("vit.merger.fc2.w"), self.img_embeds, T, bias=w.view("vit.merger.fc2.b"))
        return self.img_embeds[:T]

    def _vit_blocks_qwen2_5(self, n: int, full, cos_d, sin_d, extra) -> torch.Tensor:
        """Qwen2_5_VisionTransformerPretrainedModel.forward after the patch embedding (TF25:430-472): tokens gathered
        into window order, RMSNorm blocks with attention inside the windows (whole images in the
        fullatt_block_indexes blocks) and the biased SwiGLU MLP, RMSNorm merger, merged tokens scattered back."""
        v, L, s, w = self.cfg.vision, self.L, self.s, self.w
        D, H, hd = v.embed_dim, v.num_heads, v.head_dim
        perm_d, inv_d, win = extra
        # v_x (image order) -> v_h (window order) -> the blocks run on v_x again
        L.kr_embed_scatter(ptr(perm_d), ptr(self.v_x), 0, ptr(self.v_h), n, D, s)
        self.v_x, self.v_h = self.v_h, self.v_x
        for i in range(v.depth):
            p = f"vit.{i}."
            plan, blk_tok0, blk_ntok, blk_kr, blk_vb, qblk, qlen = full if i in v.fullatt_block_indexes else win
            L.kr_rmsnorm(ptr(self.v_x), D, ptr(w.view(p + "ln1.w")), ptr(self.v_h), n, D, 1e-6, s)
            self._gemm(self.v_h, w.view(p + "qkv.w"), self.v_qkv, n, bias=w.view(p + "qkv.b"))
            L.kr_qkv_prep(ptr(self.v_qkv), 3 * D, 0, D, 2 * D, ptr(cos_d), ptr(sin_d),
                          ptr(blk_tok0), ptr(blk_ntok), ptr(blk_kr), ptr(blk_vb), len(plan.blk_tok0),
                          ptr(self.v_q), self.v_q.stride(0), ptr(self.v_k), self.v_k.stride(0),
                          ptr(self.v_vt), self.v_vt.stride(0), H, H, hd, s)
            L.kr_attn_varlen_q(ptr(self.v_q), ptr(self.v_k), ptr(self.v_vt), ptr(self.v_o), ptr(qblk), ptr(qlen),
                               plan.qblk.shape[0], self.v_q.shape[1], H, H, hd, self.v_k.stride(0),
                               self.v_vt.stride(0), hd ** -0.5, 0, plan.q_block, s)
            self._gemm(self.v_o, w.view(p + "proj.w"), self.v_x, n, bias=w.view(p + "proj.b"), res=self.v_x)
            L.kr_rmsnorm(ptr(self.v_x), D, ptr(w.view(p + "ln2.w")), ptr(self.v_h), n, D, 1e-6, s)
            self._gemm(self.v_h, w.view(p + "gate_up.w"), self.v_f, n, bias=w.view(p + "gate_up.b"), epi=EPI_SILU_MUL8)
            self._gemm(self.v_f, w.view(p + "down.w"), self.v_x, n, bias=w.view(p + "down.b"), res=self.v_x)
        L.kr_rmsnorm(ptr(self.v_x), D, ptr(w.view("vit.merger.ln.w")), ptr(self.v_h), n, D, 1e-6, s)
        T = n // (v.spatial_merge_size ** 2)
        merged_in = self.v_h.view(-1)[: T * v.merge_dim].view(T, v.merge_dim)
        self._gemm(merged_in, w.view("vit.merger.fc1.w"), self.v_m1, T, bias=w.view("vit.merger.fc1.b"), epi=EPI_GELU_ERF)
        # second GEMM into v_m1's neighbour, then back to image order (reverse_indices, TF25:466-468)
        tmp = self.v_o.view(-1)[: T * self.cfg.text.hidden_size].view(T, self.cfg.text.hidden_size)
        self._gemm(self.v_m1, w.view("vit.merger.fc2.w"), tmp, T, bias=w.view("vit.merger.fc2.b"))
        L.kr_embed_scatter(ptr(inv_d), ptr(tmp), 0, ptr(self.img_embeds), T, self.cfg.text.hidden_size, s)
        return self.img_embeds[:T]

    # ------------------------------------------------------------------ prefill
    # ------------------------------------------------------------------ guided decoding
    def set_vocab(self, token_bytes: Sequence[bytes]):
        """Byte string of every token id (b"" for special tokens) — what a pattern is matched against.  Needed once
        before any guided request (serving.ChatFrontend hands over its tokenizer's table)."""
        from .guided import pack_vocab
        V = self.cfg.text.vocab_size
        tb = list(token_bytes)[:V] + [b""] * max(0, V - len(token_bytes))
        off, flat = pack_vocab(tb)
        self.d_voc_off = torch.from_numpy(off).to(self.device)
        self.d_voc_bytes = torch.from_numpy(flat).to(self.device)
        self._guides.clear()

    def compile_guide(self, guide) -> DeviceGuide:
        """guided.Guide (or a regex string) -> device tables; cached by pattern."""
        from .guided import Guide, compile_regex
        if isinstance(guide, DeviceGuide):
            return guide
        if isinstance(guide, str):
            hit = self._guides.get(guide)
            if hit is not None:
                return hit
            guide = compile_regex(guide)
        if not isinstance(guide, Guide):
            raise KarantaHipError(f"not a guide: {type(guide).__name__}")
        if guide.pattern and guide.pattern in self._guides:
            return self._guides[guide.pattern]
        if self.d_voc_off is None:
            raise KarantaHipError("guided decoding needs the vocabulary's byte strings: call Engine.set_vocab() first")
        S = guide.n_states
        trans = torch.from_numpy(np.ascontiguousarray(guide.trans).view(np.int16)).to(self.device)
        accept = torch.from_numpy(np.ascontiguousarray(guide.accept).astype(np.uint8)).to(self.device)
        masks = torch.empty(S, self.mask_words, dtype=torch.int32, device=self.device)
        torch.cuda.current_stream(self.device).synchronize()
        with torch.cuda.stream(self.stream):
            self.L.kr_guide_build_masks(ptr(trans), ptr(accept), S, ptr(self.d_voc_off), ptr(self.d_voc_bytes),
                                        self.cfg.text.vocab_size, ptr(self.d_eos), self.d_eos.numel(), ptr(masks),
                                        self.mask_words, self.s)
        self.stream.synchronize()
        dg = DeviceGuide(guide, trans, masks)
        if guide.pattern:
            if len(self._guides) >= 64:           # bounded cache: drop the oldest pattern
                self._guides.pop(next(iter(self._guides)))
            self._guides[guide.pattern] = dg
        return dg

    def _guide_rows(self, pages):
        """Per page: (trans address, masks address, start state, DeviceGuide | None)."""
        rows = []
        for p in pages:
            g = getattr(p, "guide", None)
            if g is None:
                rows.append((0, 0, 0, None))
                continue
            if not self._cap_guided:
                raise KarantaHipError("a page carries a guide but the engine is not in its guided configuration "
                                      "(generate() decides from its pages; begin_slots(guided=True) for slot mode)")
            self._guided = self._sampling = True      # from this admission on the steps carry the masked sampling pass
            dg = self.compile_guide(g)
            rows.append((dg.trans.data_ptr(), dg.masks.data_ptr(), dg.start, dg))
        return rows

    def _ensure_history(self, max_new_tokens: int):
        """Token history, rotary table and (when asked for) log-prob history sized for max_new_tokens; their
        addresses are baked into captured graphs, so a reallocation drops the graphs."""
        grow = self.d_hist is None or self.max_new < max_new_tokens
        if grow:
            self.max_new = max_new_tokens
            self.d_hist = torch.zeros(max_new_tokens + 1, self.B, dtype=torch.int32, device=self.device)
            self.d_cs = torch.zeros(self.B, max_new_tokens, self.cfg.text.head_dim, dtype=torch.float32, device=self.device)
        if self._logprobs is not None and (self.d_lp is None or self.d_lp.shape[0] < self.max_new + 1):
            self.d_lp = torch.zeros(self.max_new + 1, self.B, 21, dtype=torch.float32, device=self.device)
            self.d_lpi = torch.zeros(self.max_new + 1, self.B, 20, dtype=torch.int32, device=self.device)
            grow = True
        if grow:
            for g in self._graphs.values():
                self.L.kr_graph_destroy(g)
            self._graphs.clear()

    def seq_room(self) -> int:
        """Cache rows one sequence may use: prompt + generated tokens (slot mode keeps the last row as the parking row)."""
        return self.s_max - (1 if self._freeze_finished else 0)

    def _prefill_prepare(self, pages: Sequence[PageRequest], n_image_tokens_total: int, slots: Optional[Sequence[int]] = None,
                         budgets: Optional[Sequence[int]] = None):
        """Host side of `prefill` (numpy only: token sources, M-RoPE tables of the prompt and of every decode position,
        the varlen attention plan).  `generate` runs it while the ViT launches are still executing.
        ``budgets``: per page, the number of tokens it may generate (slot mode: its max_tokens + the scheduler's chunk
        overshoot); default: the current request's max_new_tokens.  The bound is each page's OWN prompt + budget — not
        the engine-wide history size, which only grows."""
        cfg, t = self.cfg, self.cfg.text
        B = len(pages)
        whole_batch = slots is None
        slots = list(range(B)) if whole_batch else [int(j) for j in slots]
        if len(slots) != B or len(set(slots)) != B or min(slots) < 0 or max(slots) >= self.B:
            raise KarantaHipError(f"slots {slots} do not name {B} distinct slots below {self.B}")
        lens = [int(len(p.input_ids)) for p in pages]
        M = sum(lens)
        if M > self.max_tokens:
            raise KarantaHipError(f"{M} prompt tokens > max_prompt_tokens {self.max_tokens}")
        room = self.seq_room()
        if budgets is None:
            budgets = [self._req_max_new] * B
        if len(budgets) != B or max(budgets) > self.max_new:
            raise KarantaHipError(f"budgets {list(budgets)} do not fit {B} pages / the history of {self.max_new} tokens")
        for n, bud in zip(lens, budgets):
            if n + int(bud) > room:
                raise KarantaHipError(f"prompt {n} + max_new_tokens {int(bud)} exceeds s_max {room}")
        src = np.empty(M, np.int32)
        cos = np.empty((M, t.head_dim), np.float32)
        sin = np.empty((M, t.head_dim), np.float32)
        deltas = np.zeros(self.B, np.int32)
        off, img_off = 0, 0
        for b, pg in enumerate(pages):
            ids = np.asarray(pg.input_ids).reshape(-1)
            is_img = ids == cfg.image_token_id
            k = int(is_img.sum())
            row = ids.astype(np.int64).copy()
            row[is_img] = -(np.arange(img_off, img_off + k) + 1)
            if row.max(initial=0) >= t.vocab_size:
                raise KarantaHipError("token id out of vocabulary")
            src[off:off + len(ids)] = row
            pos3, delta = POS.rope_index_one(ids, pg.grids, cfg.image_token_id, cfg.vision.spatial_merge_size)
            c, sn = POS.mrope_tables(pos3, t.head_dim, t.rope_theta, t.mrope_section)
            cos[off:off + len(ids)], sin[off:off + len(ids)] = c, sn
            deltas[b] = delta
            off += len(ids)
            img_off += k
        if img_off != n_image_tokens_total:
            raise KarantaHipError(f"Image features and image tokens do not match, tokens: {img_off}, "
                                  f"features: {n_image_tokens_total}")
        plan = POS.prefill_attn_plan(lens, slots, t.num_kv_heads, self.s_max)
        last_rows = (np.cumsum(lens) - 1).astype(np.int32)
        # rotary table of every decode position of every sequence: pos = P + k + delta (all three
        # M-RoPE axes equal for generated text, TF:1124-1136), cos/sin rounded to bf16 (TF:169)
        kk = np.arange(self.max_new, dtype=np.int64)
        cs = np.zeros((B, self.max_new, t.head_dim), np.float32)
        for b in range(B):
            p1 = lens[b] + kk + int(deltas[b])
            c_, s_ = POS.mrope_tables(np.stack([p1, p1, p1]), t.head_dim, t.rope_theta, t.mrope_section)
            cs[b, :, : t.head_dim // 2], cs[b, :, t.head_dim // 2:] = c_[:, : t.head_dim // 2], s_[:, : t.head_dim // 2]
        return {"B": B, "whole_batch": whole_batch, "slots": slots, "lens": lens, "M": M, "src": src, "cos": cos, "sin": sin,
                "deltas": deltas, "plan": plan, "last_rows": last_rows, "cs": cs, "n_img": n_image_tokens_total}

    def prefill(self, pages: Sequence[PageRequest], n_image_tokens_total: int,
                slots: Optional[Sequence[int]] = None, defer_activation: bool = False, prep=None):
        """embed+scatter, M-RoPE, 28 x decoder layer over the flattened prompts (causal varlen
        attention writing the KV cache), last-token logits -> first greedy token.
        Leaves the decode state (d_x, d_ctx, d_delta, history row 0) ready.  Returns prompt lengths.
        ``slots`` (slot scheduler): the cache / state slots the pages go to; only those slots' state is
        touched, the other sequences keep decoding from where they are.  ``defer_activation`` (overlapped
        admission): only the layers run here (they fill the slots' KV rows); the slot state writes and the first
        sampling step are returned as a record for `_activate` to apply on the decode stream."""
        cfg, t, L, s, w, dev = self.cfg, self.cfg.text, self.L, self.s, self.w, self.device
        if prep is None or prep["n_img"] != n_image_tokens_total:
            prep = self._prefill_prepare(pages, n_image_tokens_total, slots)
        B, whole_batch, slots, lens, M = prep["B"], prep["whole_batch"], prep["slots"], prep["lens"], prep["M"]
        src, cos, sin, deltas, plan, last_rows, cs = (prep[k] for k in ("src", "cos", "sin", "deltas", "plan", "last_rows", "cs"))
        with torch.cuda.stream(self.stream):
            self._h2d(self.p_src, src)
            self._h2d(self.p_cos, cos)
            self._h2d(self.p_sin, sin)
            temps = np.asarray([float(getattr(p, "temperature", 0.0) or 0.0) for p in pages], np.float32)
            seeds = np.asarray([int(getattr(p, "seed", 0) or 0) & 0xFFFFFFFF for p in pages], np.uint32).view(np.int32)
            if temps.max(initial=0.0) > 0:
                if not self._cap_sampling:
                    raise KarantaHipError("a page asks for temperature > 0 but the engine is in its greedy configuration "
                                          "(generate() decides from its pages; begin_slots(sampling=True) for slot mode)")
                self._sampling = True
            grows = self._guide_rows(pages)
            if whole_batch:
                tb, sb = np.zeros(self.B, np.float32), np.zeros(self.B, np.int32)
                tb[:B], sb[:B] = temps, seeds
                self._h2d(self.d_temp, tb)
                self._h2d(self.d_seed, sb)
                gt, gm, gs = np.zeros(self.B, np.int64), np.zeros(self.B, np.int64), np.zeros(self.B, np.int32)
                for b, (a_t, a_m, st, dg) in enumerate(grows):
                    gt[b], gm[b], gs[b] = a_t, a_m, st
                self._slot_guides = {b: r[3] for b, r in enumerate(grows) if r[3] is not None}
                self._h2d(self.d_gtrans, gt)
                self._h2d(self.d_gmasks, gm)
                self._h2d(self.d_gstate, gs)
                ctx0 = np.zeros(self.B, np.int32)
                ctx0[:B] = np.asarray(lens, np.int32) - 1  # kr_sample_greedy adds 1 -> number of cached tokens
                plen = np.zeros(self.B, np.int32)
                plen[:B] = lens
                self._h2d(self.d_delta, deltas)
                self._h2d(self.d_ctx, ctx0)
                self._h2d(self.d_plen, plen)
                self._h2d(self.d_cs, cs)
                self.d_fin.zero_()
            elif not defer_activation:
                self._write_slot_state(slots, lens, deltas, cs, temps, seeds, grows)
            self._h2d(self.d_last, last_rows)
            t_ = lambda a: torch.from_numpy(a).to(dev)
            blk_tok0, blk_ntok, blk_kr, blk_vb = t_(plan.blk_tok0), t_(plan.blk_ntok), t_(plan.blk_k_row0), t_(plan.blk_vt_blk)
            qblk, qlen = t_(plan.qblk), t_(plan.qblk_len)
            d, H, KVH, hd = t.hidden_size, t.num_heads, t.num_kv_heads, t.head_dim
            L.kr_embed_scatter(ptr(self.p_src), ptr(w.view("llm.embed")), ptr(self.img_embeds), ptr(self.p_x), M, d, s)
            k_head_stride = self.s_max * hd
            vt_head_stride = (self.s_max // 64) * hd * 64
            for i in range(t.num_layers):
                p = f"llm.{i}."
                L.kr_rmsnorm(ptr(self.p_x), d, ptr(w.view(p + "ln1.w")), ptr(self.p_h), M, d, t.rms_norm_eps, s)
                self._gemm(self.p_h, w.view(p + "qkv.w"), self.p_qkv, M, bias=w.view(p + "qkv.b"), packed=True, **self._w8kw(p + "qkv.w"))
                kc, vc = self.kcache[i], self.vtcache[i]
                L.kr_qkv_prep(ptr(self.p_qkv), t.qkv_dim, 0, t.q_dim, t.q_dim + t.kv_dim, ptr(self.p_cos), ptr(self.p_sin),
                              ptr(blk_tok0), ptr(blk_ntok), ptr(blk_kr), ptr(blk_vb), len(plan.blk_tok0),
                              ptr(self.p_q), self.p_q.stride(0), ptr(kc), k_head_stride, ptr(vc), vt_head_stride,
                              H, KVH, hd, s)
                L.kr_attn_varlen_q(ptr(self.p_q), ptr(kc), ptr(vc), ptr(self.p_o), ptr(qblk), ptr(qlen),
                                   plan.qblk.shape[0], self.p_q.shape[1], H, KVH, hd, k_head_stride, vt_head_stride,
                                   hd ** -0.5, 1, plan.q_block, s)
                self._gemm(self.p_o, w.view(p + "o.w"), self.p_x, M, res=self.p_x, packed=True, **self._w8kw(p + "o.w"))
                L.kr_rmsnorm(ptr(self.p_x), d, ptr(w.view(p + "ln2.w")), ptr(self.p_h), M, d, t.rms_norm_eps, s)
                self._gemm(self.p_h, w.view(p + "gate_up.w"), self.p_act, M, epi=EPI_SILU_MUL8, packed=True, **self._w8kw(p + "gate_up.w"))
                self._gemm(self.p_act, w.view(p + "down.w"), self.p_x, M, res=self.p_x, packed=True, **self._w8kw(p + "down.w"))
            # last position of every sequence -> final norm (fused) -> lm_head -> greedy token
            if whole_batch:
                L.kr_embed_scatter(ptr(self.d_last), ptr(self.p_x), 0, ptr(self.d_x), B, d, s)
                self._lm_head_and_sample(B)
            elif not defer_activation:
                self._first_tokens(slots)
        if defer_activation:
            return {"slots": slots, "lens": lens, "deltas": deltas, "cs": cs, "temps": temps, "seeds": seeds, "guides": grows}
        return lens

    def _write_slot_state(self, slots, lens, deltas, cs, temps, seeds, guides=None):
        for b, j in enumerate(slots):
            a_t, a_m, st, dg = guides[b] if guides is not None else (0, 0, 0, None)
            self._h2d(self.d_gtrans[j:j + 1], np.asarray([a_t], np.int64))
            self._h2d(self.d_gmasks[j:j + 1], np.asarray([a_m], np.int64))
            self._h2d(self.d_gstate[j:j + 1], np.asarray([st], np.int32))
            if dg is None:
                self._slot_guides.pop(j, None)
            else:
                self._slot_guides[j] = dg
            self._h2d(self.d_delta[j:j + 1], deltas[b:b + 1])
            self._h2d(self.d_ctx[j:j + 1], np.asarray([lens[b] - 1], np.int32))
            self._h2d(self.d_plen[j:j + 1], np.asarray([lens[b]], np.int32))
            self._h2d(self.d_cs[j], cs[b])
            self._h2d(self.d_temp[j:j + 1], temps[b:b + 1])
            self._h2d(self.d_seed[j:j + 1], seeds[b:b + 1])
            self.d_fin[j:j + 1].zero_()

    def _first_tokens(self, slots):
        """Last prompt position of every prefilled sequence (p_x rows d_last) -> its slot's x -> lm_head -> first token."""
        d = self.cfg.text.hidden_size
        for b, j in enumerate(slots):
            self.L.kr_embed_scatter(ptr(self.d_last[b:]), ptr(self.p_x), 0, ptr(self.d_x[j:]), 1, d, self.s)
            self._lm_head_and_sample(1, slot0=j)

    def _lm_head_and_sample(self, B: int, x=None, slot0: int = 0):
        """final RMSNorm (fused) -> lm_head with per-workgroup argmax partials -> greedy token,
        bookkeeping and the next step's rotary table (TF:839, :1320-1323; generate(do_sample=False)).
        x = the residual buffer holding the last layer's output (d_x unless the decode step ended on the
        other buffer); the next step's input embedding always goes to d_x."""
        t, L, w, s = self.cfg.text, self.L, self.w, self.s
        x = self.d_x if x is None else x
        j = slot0  # rows j .. j+B-1 of every per-sequence array (the slot scheduler prefills single slots)
        logits = self.d_logits[j:] if (self._want_logits or self._sampling or self._logprobs is not None) else None
        if self.wide_mode:
            self._dec_wide(DEC_ARGMAX, x[j:], w.view("llm.lm_head"), B, norm_w=w.view("llm.norm.w"), out_f32=logits)
        else:
            self._dec(DEC_ARGMAX, x[j:], w.view("llm.lm_head"), B, norm_w=w.view("llm.norm.w"), out_f32=logits,
                      waves=self.wv_wide)
        n_part = self._amax_parts(B) if self.wide_mode else self.n_amax   # the stride the lm_head launch wrote with
        if self._sampling:
            # temperature > 0 somewhere in the batch: the partial argmax is redone on logits / T + Gumbel noise
            # (rows with T = 0 get their plain argmax back)
            n_part = min(64, n_part)
            if self._guided:   # guided slots: only the tokens their DFA state allows take part
                L.kr_gumbel_argmax_guided(ptr(logits), self.d_logits.stride(0), t.vocab_size, ptr(self.d_temp[j:]),
                                          ptr(self.d_seed[j:]), ptr(self.d_ctx[j:]), ptr(self.d_plen[j:]), ptr(self.d_amax_v),
                                          ptr(self.d_amax_i), n_part, B, ptr(self.d_gmasks[j:]), ptr(self.d_gstate[j:]),
                                          self.mask_words, int(self.cfg.eos_token_ids[0]), s)
            else:
                L.kr_gumbel_argmax(ptr(logits), self.d_logits.stride(0), t.vocab_size, ptr(self.d_temp[j:]), ptr(self.d_seed[j:]),
                                   ptr(self.d_ctx[j:]), ptr(self.d_plen[j:]), ptr(self.d_amax_v), ptr(self.d_amax_i), n_part, B, s)
        flags = (1 if self._ignore_eos else 0) | (2 if self._freeze_finished else 0)
        L.kr_sample_greedy(ptr(self.d_amax_v), ptr(self.d_amax_i), n_part, ptr(w.view("llm.embed")), t.hidden_size,
                           ptr(self.d_tok[j:]), ptr(self.d_hist[:, j:]), self.d_hist.stride(0), ptr(self.d_plen[j:]),
                           ptr(self.d_ctx[j:]), ptr(self.d_fin[j:]), ptr(self.d_eos), self.d_eos.numel(),
                           self.cfg.pad_token_id, flags, ptr(self.d_x[j:]), B, s)
        if self._guided:
            L.kr_guide_advance(ptr(self.d_tok[j:]), ptr(self.d_fin[j:]), ptr(self.d_gtrans[j:]), ptr(self.d_gstate[j:]),
                               ptr(self.d_voc_off), ptr(self.d_voc_bytes), t.vocab_size, B, s)
        if self._logprobs is not None:
            L.kr_logprobs_topk(ptr(logits), self.d_logits.stride(0), t.vocab_size, int(self._logprobs), self.lp_part,
                               ptr(self.d_lp_pv), ptr(self.d_lp_pi), ptr(self.d_lp_ms), ptr(self.d_tok[j:]), ptr(self.d_ctx[j:]),
                               ptr(self.d_plen[j:]), ptr(self.d_fin[j:]), ptr(self.d_lp[:, j:]), ptr(self.d_lpi[:, j:]),
                               self.d_lp.shape[0], self.B, 20, B, s)

    # ------------------------------------------------------------------ decode
    def _decode_step_launches(self, B: int):
        """One decode step = 6 launches per layer + 2 (Qwen2VLDecoderLayer TF:559-624, final norm TF:839, lm_head
        TF:1320-1323): [(down_proj slab +) RMSNorm + QKV + bias + M-RoPE + KV append] -> attention partials -> merge ->
        [o_proj + residual] -> [RMSNorm + gate/up + SiLU*mul] -> [down_proj (+ residual | slab)].  Above 16 rows the residual
        sum + RMSNorm may run as a launch of their own (kr_decode_resnorm32) and the narrow linears read PACKED activations
        (kr_linear_decode32): same sums, row for row.  The measured-and-not-adopted variants of this sequence (prefetch
        branches, fast-residual mode, in-launch merges) live in csrc/tools/experiment_engine.py."""
        t, L, w, s = self.cfg.text, self.L, self.w, self.s
        H, KVH, hd = t.num_heads, t.num_kv_heads, t.head_dim
        nl = t.num_layers
        f32 = B > 16 and self.family32            # packed-activation family of 17..32-row batches
        x, x_other = self.d_x, self.d_x2          # residual stream: swaps buffers at every deferred reduction
        pending = False                           # down_proj's split-K sums of the previous layer waiting in d_part
        slabs = self.d_part.view(-1)[: 2 * B * t.hidden_size].view(2, B, t.hidden_size)   # as down_proj packs them
        one_slab = self.atomic_slab and self.defer_down
        # > 16 rows, bf16 weights: down_proj's K ranges each in TWO workgroups (kr_dec32.group_split) adding into a slab per range;
        # the pair of a layer parity replaces the one slab, and kr_decode_resnorm32 adds x + (slab 0 + slab 1): the same bits
        # (measured, profiles/r04_down_group_split.txt: 2B / 32 rows 1.691 -> 1.678 ms per step; at the 7B width, where two tiles
        # per workgroup already share the x fragments, 3.968 -> 3.983: there it stays off unless KARANTA_DOWN_GS=2)
        gs = (f32 and one_slab and self.resnorm32_qkv and not self.fp8
              and (self.group_split_down == 2 or self.group_split_down == 1 and self.down_waves_small == 16)
              and (t.hidden_size // 16) % (4 if self.down_waves_small == 8 else 2) == 0)
        pairs = self.d_part.view(-1)[: 4 * B * t.hidden_size].view(2, 2, B, t.hidden_size) if gs else None
        for i in range(nl):
            p = f"llm.{i}."
            kc, vc = ptr(self.kcache[i]), ptr(self.vtcache[i])     # the cache tensors are [layers, max_batch, ...]
            # ---- 1. (x += down_proj sums of layer i - 1) -> RMSNorm -> QKV + bias + M-RoPE + KV append
            if not self.narrow_mode:
                self._dec(DEC_ROPE_KV, x, w.view(p + "qkv.w"), B, bias=w.view(p + "qkv.b"), norm_w=w.view(p + "ln1.w"),
                          waves=self.wv_qkv, kc=kc, vc=vc)
            else:
                # this layer's down_proj will ADD into accumulator (i + 1) & 1: the qkv launch zeroes it (it was last read by
                # layer i - 1's qkv launch, which is complete)
                zero = slabs[(i + 1) & 1] if (one_slab and i + 1 < nl) else None
                pin = ((slabs[i & 1:(i & 1) + 1] if one_slab else slabs) if pending else None)
                if gs:
                    zero = pairs[(i + 1) & 1] if i + 1 < nl else None
                    pin = pairs[i & 1] if pending else None
                if B > 16 and (self.resnorm_qkv or f32 and self.resnorm32_qkv):
                    # the residual sum + RMSNorm ONCE for the batch (bit-identical rows), then ONE qkv launch over all rows
                    args = (ptr(x), x.stride(0), ptr(pin), int(pin.shape[0]) if pin is not None else 0, B, ptr(x_other),
                            x_other.stride(0), ptr(w.view(p + "ln1.w")), t.rms_norm_eps, ptr(self.d_h))
                    if f32:
                        L.kr_decode_resnorm32(*args, B, t.hidden_size, 1 if (gs and pin is not None) else 0, s)
                        self._dec32(DEC_ROPE_KV, self.d_h, w.view(p + "qkv.w"), B, 8, bias=w.view(p + "qkv.b"), kc=kc, vc=vc,
                                    zero=zero, **self._w8kw(p + "qkv.w"))
                    else:
                        L.kr_decode_resnorm(*args, self.d_h.stride(0), B, t.hidden_size, s)
                        self._dec_narrow(DEC_ROPE_KV, self.d_h, w.view(p + "qkv.w"), B, bias=w.view(p + "qkv.b"), kc=kc, vc=vc,
                                         zero=zero, **self._w8kw(p + "qkv.w"))
                else:
                    for k, (r0, m) in enumerate(self._row_ranges(B)):     # one launch, or one per 16-row range (row_split)
                        self._dec_narrow(DEC_ROPE_KV, x[r0:], w.view(p + "qkv.w"), m, bias=w.view(p + "qkv.b"),
                                         norm_w=w.view(p + "ln1.w"), part_in=pin[:, r0:] if pin is not None else None,
                                         x_out=x_other[r0:] if pin is not None else None, kc=kc, vc=vc,
                                         part_rows=B if (pin is not None and self.row_split and B > 16) else 0, row0=r0,
                                         zero=zero if k == 0 else None, **self._w8kw(p + "qkv.w"))
                if pending:
                    x, x_other = x_other, x
                    pending = False
            # ---- 2. split-KV attention partials, 3. their merge
            L.kr_attn_decode_slots(ptr(self.d_q), kc, vc, ptr(self.d_ctx), ptr(self.d_fin), ptr(self.d_ws), B, H, KVH, hd, self.s_max,
                                   self.n_split, hd ** -0.5, s)
            if f32:
                L.kr_attn_decode_merge32(ptr(self.d_ws), ptr(self.d_o), B, H, hd, self.n_split, s)
            else:
                L.kr_attn_decode_merge(ptr(self.d_ws), ptr(self.d_o), B, H, hd, self.n_split, s)
            # ---- 4. o_proj + residual
            if f32:
                self._dec32(DEC_PLAIN, self.d_o, w.view(p + "o.w"), B, self.o_waves, out=x, res=x, **self._w8kw(p + "o.w"))
            elif self.narrow_o:
                self._dec_narrow(DEC_PLAIN, self.d_o, w.view(p + "o.w"), B, out=x, res=x, waves=self.o_waves if B <= 16 else 8,
                                 **self._w8kw(p + "o.w"))
            else:
                self._dec(DEC_PLAIN, self.d_o, w.view(p + "o.w"), B, out=x, res=x, waves=self.wv_o)
            # ---- 5. RMSNorm + gate/up + SiLU*mul (the kernel bench.py's roofline object is measured on)
            if self._prof_on:
                # [e0][e1] gate/up [e2]: the empty bracket e0..e1 measures what two back-to-back event
                # packets cost by themselves; it is subtracted from the bracket around the launch
                (e0, e1), (e2, _) = self._prof_event_pair(), self._prof_event_pair()
                L.kr_event_record(e0, s)
                L.kr_event_record(e1, s)
            if self.wide_mode:
                self._dec_wide(DEC_SILU8 | (DEC_OUT_XP if f32 else 0), x, w.view(p + "gate_up.w"), B, out=self.d_act,
                               norm_w=w.view(p + "ln2.w"), **self._w8kw(p + "gate_up.w"))
            else:
                self._dec(DEC_SILU8, x, w.view(p + "gate_up.w"), B, out=self.d_act, norm_w=w.view(p + "ln2.w"), waves=self.wv_wide)
            if self._prof_on:
                L.kr_event_record(e2, s)
            # ---- 6. down_proj: two K ranges per tile whose sums the next layer's first launch adds to x (deferred split-K),
            # or (last layer / widths without the deferral) down_proj + residual
            defer = self.defer_down and i + 1 < nl
            acc = slabs[(i + 1) & 1] if (defer and self.atomic_slab) else (self.d_part if defer else None)
            if f32 and gs and defer:
                self._dec32(DEC_PLAIN, self.d_act, w.view(p + "down.w"), B, self.down_waves_small, ksplit=2, out_f32=pairs[(i + 1) & 1],
                            atomic_out=True, group_split=True, tiles_per_wg=self.down_gs_tiles)
            elif f32:
                self._dec32(DEC_PLAIN, self.d_act, w.view(p + "down.w"), B, self.down_waves_small, ksplit=2 if defer else 1,
                            out_f32=acc, atomic_out=defer and self.atomic_slab, out=None if defer else x, res=None if defer else x,
                            **self._w8kw(p + "down.w"))
            elif defer:
                self._dec_narrow(DEC_PLAIN, self.d_act, w.view(p + "down.w"), B, out_f32=acc, waves=self._down_waves(B), ksplit=2,
                                 atomic_out=self.atomic_slab, **self._w8kw(p + "down.w"))
            elif self.narrow_mode:
                self._dec_narrow(DEC_PLAIN, self.d_act, w.view(p + "down.w"), B, out=x, res=x, waves=self._down_waves(B),
                                 **self._w8kw(p + "down.w"))
            else:
                self._dec(DEC_PLAIN, self.d_act, w.view(p + "down.w"), B, out=x, res=x, waves=self.wv_down)
            pending = defer
        self._lm_head_and_sample(B, x)

    # ------------------------------------------------------------------ live kernel timing (bench.py roofline)
    def _prof_event_pair(self):
        if self._prof_next == len(self._prof_events):
            e0, e1 = C.c_void_p(), C.c_void_p()
            self.L.kr_event_create(C.byref(e0))
            self.L.kr_event_create(C.byref(e1))
            self._prof_events.append((e0, e1))
        pair = self._prof_events[self._prof_next]
        self._prof_next += 1
        return pair

    def kernel_profile(self, reset: bool = True) -> Dict[str, float]:
        """Durations of the decode gate/up projection (the kernel that moves half of the decoder's bytes)
        measured with HIP events on the launch stream during the profiled eager steps.  Each sample is a HIP-event bracket around the one launch, minus an empty bracket
        (two events, nothing between) recorded right before it — the event packets' own cost.  Returns {launches, avg_us, min_us, bracket_us, null_bracket_us, bytes_per_launch}."""
        self.stream.synchronize()
        ms = C.c_float()
        vals, nulls = [], []
        ev = self._prof_events[: self._prof_next]
        for (e0, e1), (e2, _) in zip(ev[0::2], ev[1::2]):
            self.L.kr_event_elapsed_ms(e0, e1, C.byref(ms))
            nulls.append(ms.value * 1e3)
            self.L.kr_event_elapsed_ms(e1, e2, C.byref(ms))
            vals.append(ms.value * 1e3)
        if reset:
            self._prof_next = 0
        t = self.cfg.text
        B = self._last_batch
        nbytes = 2 * (2 * t.intermediate_size * t.hidden_size + t.hidden_size + B * t.hidden_size + B * t.intermediate_size)
        raw = float(np.mean(vals)) if vals else 0.0
        null = float(np.mean(nulls)) if nulls else 0.0
        return {"launches": len(vals), "bracket_us": raw, "null_bracket_us": null, "avg_us": max(raw - null, 0.0),
                "min_us": float(np.min(vals) - null) if vals else 0.0, "bytes_per_launch": nbytes}

    def gate_up_chain_profile(self, B: int, reps: int = 8) -> Dict[str, float]:
        """Average launch duration of the decode gate/up kernel, HIP events on the launch stream around a replayed
        graph of reps x num_layers back-to-back launches, each on its own layer's weights (so no launch finds its
        weights in a cache: 28 x 55 MB against 32 MB of L2 and 256 MB of Infinity Cache).  The per-launch figure
        includes the dispatch gap between dependent launches, as a rocprofv3 kernel span in a graph replay does."""
        t, w, L = self.cfg.text, self.w, self.L
        e0, e1 = self._prof_event_pair()
        self._prof_next -= 1
        def chain(n):
            for _ in range(n):
                for i in range(t.num_layers):
                    p = f"llm.{i}."
                    if self.wide_mode:
                        self._dec_wide(DEC_SILU8, self.d_x, w.view(p + "gate_up.w"), B, out=self.d_act, norm_w=w.view(p + "ln2.w"),
                                       **self._w8kw(p + "gate_up.w"))
                    else:
                        self._dec(DEC_SILU8, self.d_x, w.view(p + "gate_up.w"), B, out=self.d_act, norm_w=w.view(p + "ln2.w"),
                                  waves=self.wv_wide)
        chain(1)                               # eager once (function attributes), then captured: a replayed
        self.stream.synchronize()              # graph issues the launches at the GPU's pace, not Python's
        g = C.c_void_p()
        L.kr_graph_begin_capture(self.s)
        try:
            chain(reps)
        finally:
            L.kr_graph_end_capture(self.s, C.byref(g))
        ms, best = C.c_float(), None
        for _ in range(3):
            L.kr_event_record(e0, self.s)
            L.kr_graph_launch(g, self.s)
            L.kr_event_record(e1, self.s)
            L.kr_event_synchronize(e1)
            L.kr_event_elapsed_ms(e0, e1, C.byref(ms))
            best = ms.value if best is None else min(best, ms.value)
        L.kr_graph_destroy(g)
        ms.value = best
        n = reps * t.num_layers
        wbytes = 2 * t.intermediate_size * t.hidden_size * (1 if self.fp8 else 2) + (8 * t.intermediate_size if self.fp8 else 0)
        nbytes = wbytes + 2 * (t.hidden_size + B * t.hidden_size + B * t.intermediate_size)
        return {"launches": n, "avg_us": ms.value * 1e3 / n, "bytes_per_launch": nbytes}

    def _graph_key(self, B: int):
        return (B, self._ignore_eos, self._freeze_finished, self._sampling, self._guided, self._logprobs)

    def _graph_for(self, B: int) -> int:
        key = self._graph_key(B)
        assert not self._want_logits
        g = self._graphs.get(key)
        if g is None:
            L = self.L
            L.kr_graph_begin_capture(self.s)
            try:
                self._decode_step_launches(B)
            finally:
                ge = C.c_void_p()
                L.kr_graph_end_capture(self.s, C.byref(ge))
            g = ge.value
            self._graphs[key] = g
        return g

    # ------------------------------------------------------------------ public API
    def generate(self, pages: Sequence[PageRequest], max_new_tokens: int, ignore_eos: bool = False,
                 use_graph: bool = True, return_logits: bool = False, sync_every: int = 32,
                 pixel_values_device: Optional[torch.Tensor] = None, profile_every: int = 0,
                 force_tokens: Optional[np.ndarray] = None) -> GenerateResult:
        """Greedy generation for a static batch of pages (temperature 0 — the reference's
        ``build_page_query`` default, /root/reference/karanta/pipeline.py:166-171).

        ``pixel_values_device``: all pages' patches already resident in HBM (fp32 ``[n, 1176]``), used
        instead of the per-page host arrays.  ``profile_every`` > 0: every that many decode steps one
        step runs eagerly with HIP events around the dominant kernel (see :meth:`kernel_profile`).
        ``force_tokens`` (int [B, >= max_new_tokens - 1], eager path: a parity-test instrument): teacher forcing — after
        every sampling step the NEXT step's input embedding is replaced by that of the given token, while `tokens` /
        `logits` still report the engine's own argmax and logits.  With a reference's tokens as the forced sequence
        every step can be compared with the reference, also past a near-tie where free-running sequences part."""
        B = len(pages)
        if not 1 <= B <= self.B:
            raise KarantaHipError(f"batch {B} not in 1..{self.B}")
        if max_new_tokens < 1:
            raise ValueError("max_new_tokens must be >= 1")
        t0 = time.perf_counter()
        self._ignore_eos = bool(ignore_eos)
        self._freeze_finished = False
        self._want_logits = bool(return_logits)
        self._guided = any(getattr(p, "guide", None) is not None for p in pages)
        # a guided row is masked in the sampling pass, so that pass runs (rows with T = 0 stay a plain argmax)
        self._sampling = self._guided or any(float(getattr(p, "temperature", 0.0) or 0.0) > 0 for p in pages)
        self._cap_sampling, self._cap_guided = self._sampling, self._guided
        ks = [int(p.logprobs) for p in pages if getattr(p, "logprobs", None) is not None]
        if ks and not 0 <= max(ks) <= 20:
            raise KarantaHipError("logprobs must be in 0..20")
        self._logprobs = max(ks) if ks else None
        self._last_batch = B
        self._ensure_history(max_new_tokens)
        self._req_max_new = int(max_new_tokens)
        grids = [g for p in pages for g in p.grids]
        n_img_tok = 0
        pix = self._pixels_for(pages, pixel_values_device)
        if pix is not None:
            n_img_tok = self.vit_forward(pix, grids).shape[0]
        prep = self._prefill_prepare(pages, n_img_tok)   # host work of the prefill, under the ViT launches still running
        self.stream.synchronize()
        t1 = time.perf_counter()
        lens = self.prefill(pages, n_img_tok, prep=prep)
        forced = None
        if force_tokens is not None:
            ft = np.asarray(force_tokens, np.int64)
            if ft.ndim != 2 or ft.shape[0] != B or ft.shape[1] < max_new_tokens - 1 or ft.min() < 0 or ft.max() >= self.cfg.text.vocab_size:
                raise KarantaHipError(f"force_tokens must be [B={B}, >= {max_new_tokens - 1}] token ids")
            forced = torch.from_numpy(np.ascontiguousarray(ft.T.astype(np.int32))).to(self.device)   # [steps, B]
            use_graph = False

        def force(k):     # the input of decode step k + 1 becomes the embedding of forced token k
            if forced is not None and k < forced.shape[0]:
                self.L.kr_embed_scatter(ptr(forced[k]), ptr(self.w.view("llm.embed")), 0, ptr(self.d_x), B,
                                        self.cfg.text.hidden_size, self.s)

        with torch.cuda.stream(self.stream):
            force(0)
        logits_steps = []
        if return_logits:
            self.stream.synchronize()
            logits_steps.append(self.d_logits[:B].float().cpu().numpy().copy())
        self.stream.synchronize()
        t2 = time.perf_counter()
        steps_done = 1
        with torch.cuda.stream(self.stream):
            want_graph = use_graph and not return_logits
            graph = self._graphs.get(self._graph_key(B)) if want_graph else None
            while steps_done < max_new_tokens:
                if graph is not None and profile_every and steps_done % profile_every == 0:
                    self._prof_on = True
                    self._decode_step_launches(B)
                    self._prof_on = False
                elif graph is not None:
                    self.L.kr_graph_launch(graph, self.s)
                else:
                    # eager step; the first one also sets per-kernel attributes, so capture only
                    # after it (no attribute calls inside a stream capture)
                    self._decode_step_launches(B)
                    force(steps_done)
                    if want_graph:
                        graph = self._graph_for(B)
                steps_done += 1
                if return_logits:
                    self.stream.synchronize()
                    logits_steps.append(self.d_logits[:B].float().cpu().numpy().copy())
                if not ignore_eos and steps_done % sync_every == 0:
                    self.stream.synchronize()
                    if bool(self.d_fin[:B].all().item()):
                        break
        self.stream.synchronize()
        t3 = time.perf_counter()
        hist = self.d_hist[:steps_done, :B].cpu().numpy().T  # [B, steps]
        toks, reasons = [], []
        eos = set(int(e) for e in self.cfg.eos_token_ids)
        for b in range(B):
            row = hist[b]
            cut, reason = len(row), "length"
            if not ignore_eos:
                hit = np.flatnonzero(np.isin(row, list(eos)))
                if hit.size:
                    cut, reason = int(hit[0]) + 1, "stop"
            toks.append(row[:cut].astype(np.int64))
            reasons.append(reason)
        lps = None
        if self._logprobs is not None:
            lp = self.d_lp[:steps_done, :B].cpu().numpy()
            li = self.d_lpi[:steps_done, :B].cpu().numpy()
            lps = []
            for b, p in enumerate(pages):
                k = getattr(p, "logprobs", None)
                n = len(toks[b]) - (1 if reasons[b] == "stop" else 0)   # the EOS step records nothing
                lps.append(None if k is None else {"token": lp[:n, b, 0].copy(), "top": lp[:n, b, 1:1 + int(k)].copy(),
                                                   "top_ids": li[:n, b, :int(k)].astype(np.int64)})
        return GenerateResult(
            tokens=toks, finish_reasons=reasons, prompt_tokens=lens,
            timings={"vit_s": t1 - t0, "prefill_s": t2 - t1, "decode_s": t3 - t2, "total_s": t3 - t0,
                     "decode_steps": steps_done - 1},
            logits=np.stack(logits_steps, 1) if return_logits else None, logprobs=lps)

    # ------------------------------------------------------------------ slot scheduler API (continuous batching)
    # The decode graph always runs all `max_batch` slots; a slot whose sequence has finished idles in place
    # (kr_sample_greedy freeze bit) until `admit` prefills a new request into it.  See scheduler.SlotScheduler.
    def begin_slots(self, max_new_tokens: int, sampling: bool = False, guided: bool = False, logprobs: Optional[int] = None):
        """Enter slot mode: every slot idle, per-slot history / rotary tables sized for `max_new_tokens`.
        sampling=True: the decode graph carries the Gumbel-max pass, so requests may ask for temperature > 0.
        guided=True: that pass masks the logits of slots that carry a guide and their DFA state advances in the graph
        (needs set_vocab()).  logprobs=k: every step records log-probabilities of the token and of the k most probable."""
        if max_new_tokens < 1:
            raise ValueError("max_new_tokens must be >= 1")
        if logprobs is not None and not 0 <= int(logprobs) <= 20:
            raise KarantaHipError("logprobs must be in 0..20")
        if guided and self.d_voc_off is None:
            raise KarantaHipError("guided decoding needs the vocabulary's byte strings: call Engine.set_vocab() first")
        self._ignore_eos, self._freeze_finished, self._want_logits = False, True, False
        self._guided = bool(guided)
        self._sampling = bool(sampling) or self._guided
        self._cap_sampling, self._cap_guided = self._sampling, self._guided
        self._logprobs = None if logprobs is None else int(logprobs)
        self._last_batch = self.B
        self._ensure_history(max_new_tokens)
        self._req_max_new = int(max_new_tokens)
        # a scheduler rebuilt after an exception may have left an overlapped admission in flight: without this the counter stays
        # above zero and every later decode_steps() replays the graph on the complement CU subset
        if self._adm_stream is not None:
            self._adm_stream.synchronize()
        self._adm_inflight = 0
        self._snap_event = None
        with torch.cuda.stream(self.stream):
            self.d_fin.fill_(1)
            self.d_temp.zero_()
            self.d_gtrans.zero_()
            self.d_gmasks.zero_()
            self._slot_guides = {}
            self.d_ctx.zero_()
            self.d_plen.zero_()
            self.d_x.zero_()
        self.stream.synchronize()

    def admit(self, pages: Sequence[PageRequest], slots: Sequence[int], budgets: Optional[Sequence[int]] = None) -> List[int]:
        """ViT + prefill of new requests into idle slots; their first token is sampled.  Returns prompt lengths.
        ``budgets``: tokens each page may generate before the host retires it (its max_tokens + the chunk overshoot)."""
        self._check_budgets(pages, budgets)       # before any launch: an oversized page fails the call, nothing ran
        grids = [g for p in pages for g in p.grids]
        pix = self._pixels_for(pages, None)
        n_img_tok = self.vit_forward(pix, grids).shape[0] if pix is not None else 0
        prep = self._prefill_prepare(pages, n_img_tok, slots, budgets)   # host tables while the ViT launches execute
        return self.prefill(pages, n_img_tok, slots=slots, prep=prep)

    def _check_budgets(self, pages, budgets):
        room = self.seq_room()
        for i, p in enumerate(pages):
            bud = int(budgets[i]) if budgets is not None else self._req_max_new
            if len(p.input_ids) + bud > room:
                raise KarantaHipError(f"prompt {len(p.input_ids)} + max_new_tokens {bud} exceeds s_max {room}")

    # Overlapped admission (optional, SlotScheduler(overlap=True); measured: no gain on the ragged serving benchmark —
    # 6.56 vs 6.57 pages/s, with or without a high-priority decode stream: the ViT / prefill launches fill all 256 CUs
    # with long-running workgroups and the short decode launches wait for slots, so the two streams serialise in
    # practice).  ViT + prefill of the new requests run on a second stream while the decode graph keeps
    # stepping the other slots.  The target slots are first PARKED on the last cache row (their frozen per-step KV
    # write then cannot land inside the rows the prefill is filling); the slot state and the first sampling step are
    # applied on the decode stream once the admission stream has finished.
    def admit_begin(self, pages: Sequence[PageRequest], slots: Sequence[int], budgets: Optional[Sequence[int]] = None):
        self._check_budgets(pages, budgets)
        slots = [int(j) for j in slots]
        if self._adm_stream is None:
            if self.admission_cus > 0:
                h = C.c_void_p()
                self.L.kr_stream_create_cu_mask(C.byref(h), int(self.admission_cus))
                self._adm_stream_handle = h
                self._adm_stream = torch.cuda.ExternalStream(h.value, device=self.device)
            else:
                self._adm_stream = torch.cuda.Stream(device=self.device)
        park = np.asarray([self.s_max - 1], np.int32)
        with torch.cuda.stream(self.stream):
            for j in slots:
                self.d_fin[j:j + 1].fill_(1)
                self._h2d(self.d_ctx[j:j + 1], park)
                self._h2d(self.d_plen[j:j + 1], park)
            parked = torch.cuda.Event()
            parked.record(self.stream)
        self._adm_stream.wait_event(parked)
        main = (self.stream, self.s)
        self.stream, self.s = self._adm_stream, self._adm_stream.cuda_stream
        try:
            grids = [g for p in pages for g in p.grids]
            pix = self._pixels_for(pages, None)
            n_img_tok = self.vit_forward(pix, grids).shape[0] if pix is not None else 0
            rec = self.prefill(pages, n_img_tok, slots=slots, defer_activation=True,
                               prep=self._prefill_prepare(pages, n_img_tok, slots, budgets))
            done = torch.cuda.Event()
            done.record(self._adm_stream)
        finally:
            self.stream, self.s = main
        self._adm_inflight += 1
        return {"done": done, "rec": rec}

    def admit_ready(self, handle) -> bool:
        return bool(handle["done"].query())

    def admit_end(self, handle) -> List[int]:
        """Activate the admitted slots on the decode stream (waits for the admission stream there, not on the host)."""
        rec = handle["rec"]
        self._adm_inflight = max(0, self._adm_inflight - 1)
        self.stream.wait_event(handle["done"])
        with torch.cuda.stream(self.stream):
            self._write_slot_state(rec["slots"], rec["lens"], rec["deltas"], rec["cs"], rec["temps"], rec["seeds"],
                                   rec.get("guides"))
            self._first_tokens(rec["slots"])
        return rec["lens"]

    def set_step_features(self, sampling: bool, guided: bool):
        """Slot mode: which passes the NEXT decode steps carry, within what begin_slots() allowed.  The scheduler calls it with what
        the requests in the slots need: a server that accepts guided / sampled requests runs the plain argmax graph (no f32 logits
        written and re-read, no DFA advance) while none is decoding — rows with temperature 0 and no guide get the same token from
        either graph.  An admission that brings a guide or a temperature switches the passes on by itself (prefill)."""
        guided = bool(guided) and self._cap_guided
        self._guided = guided
        self._sampling = (bool(sampling) and self._cap_sampling) or guided

    def decode_steps(self, n: int):
        """n decode steps over all slots (asynchronous on the engine's stream).  While an admission is in flight on a CU-masked
        stream the captured graph replays on the complementary compute units (same kernels, same grids: the same tokens), ordered
        against the engine's stream by events on both sides."""
        graph = self._graphs.get(self._graph_key(self.B))
        self.last_decode_disjoint = False      # (observable for tests: which of the two paths the call took)
        if (graph is not None and self._adm_inflight > 0 and self.admission_cus > 0 and self.disjoint_decode
                and self.admission_cus < self.n_cus):
            self.last_decode_disjoint = True
            if self._dec_stream is None:
                h = C.c_void_p()
                self.L.kr_stream_create_cu_range(C.byref(h), int(self.admission_cus), int(self.n_cus - self.admission_cus))
                self._dec_stream_handle = h
                self._dec_stream = torch.cuda.ExternalStream(h.value, device=self.device)
            before = torch.cuda.Event()
            before.record(self.stream)
            self._dec_stream.wait_event(before)
            for _ in range(n):
                self.L.kr_graph_launch(graph, self._dec_stream.cuda_stream)
            after = torch.cuda.Event()
            after.record(self._dec_stream)
            self.stream.wait_event(after)
            return
        with torch.cuda.stream(self.stream):
            for _ in range(n):
                graph = self._graphs.get(self._graph_key(self.B))
                if graph is not None:
                    self.L.kr_graph_launch(graph, self.s)
                else:  # first step eager (kernel attributes), then captured
                    self._decode_step_launches(self.B)
                    self._graph_for(self.B)

    def poll_slots(self):
        """(finished[B], generated[B]): device EOS flags and tokens generated so far per slot (synchronises)."""
        self.stream.synchronize()
        self._snap_event = None
        fin = self.d_fin.cpu().numpy().astype(bool)
        gen = (self.d_ctx.cpu().numpy() + 1 - self.d_plen.cpu().numpy()).astype(np.int64)
        return fin, gen

    # poll_slots without draining the stream (SlotScheduler launch-ahead): the slot state as of THIS point of the engine's stream is
    # copied to pinned host memory behind an event; the scheduler queues the next decode chunk before it waits for that event, so
    # the GPU has work while the host harvests finished slots and prepares the next admission.
    def snapshot_slots(self):
        ring = self._snap_ring
        if ring is None:
            ring = self._snap_ring = [torch.empty(3, self.B, dtype=torch.int32).pin_memory() for _ in range(4)]
        buf = ring[self._snap_next % len(ring)]
        self._snap_next += 1
        with torch.cuda.stream(self.stream):
            buf[0].copy_(self.d_fin, non_blocking=True)
            buf[1].copy_(self.d_ctx, non_blocking=True)
            buf[2].copy_(self.d_plen, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return {"buf": buf, "event": ev}

    def read_snapshot(self, snap):
        """(finished[B], generated[B]) of a snapshot_slots() handle; waits for that point of the stream only.  Later slot_tokens /
        slot_logprobs reads run on a side stream ordered after the same point (history rows are append-only)."""
        snap["event"].synchronize()
        a = snap["buf"].numpy()
        fin = a[0].astype(bool)
        gen = (a[1].astype(np.int64) + 1 - a[2])
        self._snap_event = snap["event"]
        return fin, gen

    def _host_copy(self, t: "torch.Tensor") -> np.ndarray:
        """Device -> host of a slot's history.  After read_snapshot(): on the copy stream, behind the snapshot's event, so that the
        read does not wait for the decode chunk already queued on the engine's stream."""
        if self._snap_event is None:
            return t.cpu().numpy()
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=self.device)
        self._copy_stream.wait_event(self._snap_event)
        with torch.cuda.stream(self._copy_stream):
            return t.cpu().numpy()

    def slot_tokens(self, slot: int, n: int) -> np.ndarray:
        return self._host_copy(self.d_hist[:n, slot]).astype(np.int64)

    def slot_logprobs(self, slot: int, n: int, k: int) -> Dict[str, np.ndarray]:
        """Log-probabilities of the first n generated tokens of a slot (begin_slots(logprobs=...))."""
        if self.d_lp is None or self._logprobs is None:
            raise KarantaHipError("log-probabilities were not recorded: begin_slots(logprobs=k)")
        k = min(int(k), int(self._logprobs))
        lp = self._host_copy(self.d_lp[:n, slot])
        return {"token": lp[:, 0].copy(), "top": lp[:, 1:1 + k].copy(),
                "top_ids": self._host_copy(self.d_lpi[:n, slot, :k]).astype(np.int64)}

    def retire(self, slot: int):
        """Host-side stop (length limit): the slot idles from the next step on."""
        with torch.cuda.stream(self.stream):
            self.d_fin[slot:slot + 1].fill_(1)

    def close(self):
        for g in self._graphs.values():
            self.L.kr_graph_destroy(g)
        self._graphs.clear()
        if self._adm_stream_handle is not None:
            torch.cuda.synchronize(self.device)
            self._adm_stream = None
            self.L.kr_stream_destroy(self._adm_stream_handle)
            self._adm_stream_handle = None
        if self._dec_stream_handle is not None:
            torch.cuda.synchronize(self.device)
            self._dec_stream = None
            self.L.kr_stream_destroy(self._dec_stream_handle)
            self._dec_stream_handle = None
