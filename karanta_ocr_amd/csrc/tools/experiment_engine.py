"""The measured-and-not-adopted variants of the decode step (rounds 1-3), kept runnable for A/B timing
(csrc/tools/decode_ab.py) and for their tests — NOT part of the product path.

`ExperimentEngine` is `karanta_ocr_amd.engine.Engine` with the round-3 launch sequence of one decode step, which interleaves the
product launches with:
  KARANTA_PREFETCH=1..6        Infinity-Cache prefetch of a layer's MLP weights (serial / second graph branch / idle-CU workgroups
                               riding on the qkv launch)                                   DESIGN.md 5-r2: all slower
  fast_residual                o_proj split by attention head + float atomics into an f32 residual accumulator, no merge launch
  KARANTA_ATTN_FUSED=1         split-KV merge inside the attention launch (last-arriving workgroup)
  KARANTA_MERGE_IN_OPROJ=1     split-KV merge in the o_proj prologue (general dec_linear_kernel)
  KARANTA_EXTRA_NULLS=n        n empty launches per layer (the price of a launch in the chain)
Most of them need a library built with -DKR_EXPERIMENTS (csrc/tools/build_variant.py exp kr_decode.hip,kr_selftest.hip
-DKR_EXPERIMENTS; load it through KARANTA_HIP_LIB): include/karanta_hip_experiments.h.  The 17..32-row packed family
(kr_linear_decode32) is not wired in here: this sequence keeps round 3's row-major narrow launches at every batch size."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np
import torch

from karanta_ocr_amd._lib import DEC_ARGMAX, DEC_PLAIN, DEC_ROPE_KV, DEC_SILU8, KarantaHipError, ptr
from karanta_ocr_amd.engine import Engine


class ExperimentEngine(Engine):
    def __init__(self, *args, fast_residual: Optional[bool] = None, **kw):
        super().__init__(*args, **kw)
        t, dev, B = self.cfg.text, self.device, self.B
        self._extra_nulls = int(os.environ.get("KARANTA_EXTRA_NULLS", "0"))
        self._prefetch_mode = int(os.environ.get("KARANTA_PREFETCH", "0"))
        self._pf_blocks = int(os.environ.get("KARANTA_PREFETCH_BLOCKS", "256"))
        self._pf_stream = None
        self.merge_in_o_proj = os.environ.get("KARANTA_MERGE_IN_OPROJ", "0") == "1"
        self.attn_fused_merge = os.environ.get("KARANTA_ATTN_FUSED", "0") == "1"
        want_fast = (os.environ.get("KARANTA_FAST_RESIDUAL", "0") == "1") if fast_residual is None else bool(fast_residual)
        if not self.L.experiments and (self._prefetch_mode or want_fast or self.attn_fused_merge):
            raise KarantaHipError("KARANTA_PREFETCH / KARANTA_FAST_RESIDUAL / KARANTA_ATTN_FUSED need a library built with "
                                  "-DKR_EXPERIMENTS (csrc/tools/build_variant.py), loaded through KARANTA_HIP_LIB")
        self.family32 = False          # round 3's sequence: row-major narrow launches at every batch size
        self.d_xacc = torch.zeros(B, t.hidden_size, dtype=torch.float32, device=dev)      # fast-residual mode: f32 residual accumulator
        self.d_cnt = torch.zeros(B * t.num_kv_heads, dtype=torch.int32, device=dev)       # arrival counters of the in-launch merge
        self.fast_residual = want_fast and self.narrow_mode and self.wide_mode and not self.row_split

    def _pf_events(self, layer: int):
        """(fork, join) events of layer `layer`'s prefetch branch, and the side stream they need (created on first use)."""
        if self._pf_stream is None:
            self._pf_stream = torch.cuda.Stream(device=self.device)
            self._pf_ev = {}
        ev = self._pf_ev.get(layer)
        if ev is None:
            a, b = C.c_void_p(), C.c_void_p()
            self.L.kr_event_create(C.byref(a))
            self.L.kr_event_create(C.byref(b))
            ev = self._pf_ev[layer] = (a, b)
        return ev

    def _dec_narrow_experiment(self, head, tail, W, w8, w_scale, x_out_f32=None, prefetch=None):
        """kr_linear_decode_narrow_x32: workgroup 0 also stores x_new as the f32 accumulator's start value (fast-residual mode);
        prefetch = (address, bytes, blocks): prefetch workgroups ride on the launch."""
        if x_out_f32 is None and prefetch is None:
            if w8 is not None:
                self.L.kr_linear_decode_narrow_fp8(*head, ptr(w8), ptr(w_scale), *tail, self.s)
            else:
                self.L.kr_linear_decode_narrow(*head, ptr(W), *tail, self.s)
            return
        pf = prefetch or (0, 0, 0)
        self.L.kr_linear_decode_narrow_x32(*head, ptr(x_out_f32), x_out_f32.stride(0) if x_out_f32 is not None else 0,
                                           ptr(w8 if w8 is not None else W), ptr(w_scale), *tail, int(pf[0]), int(pf[1]), int(pf[2]), self.s)

    def _dec_wide_x32(self, mode, x_f32, x_out, W, M, out=None, out_f32=None, norm_w=None, w8=None, w_scale=None):
        """kr_linear_decode_wide_x32 (fast-residual mode): the rows come from the f32 residual accumulator; x_out receives their
        bf16 rounding (workgroup 0)."""
        N, K = W.shape
        blocks, waves = self._wide_geometry(N, M)
        o = out if out is not None else out_f32
        self.L.kr_linear_decode_wide_x32(mode, ptr(x_f32), x_f32.stride(0), ptr(x_out), x_out.stride(0) if x_out is not None else 0,
                                         ptr(w8 if w8 is not None else W), ptr(w_scale), ptr(norm_w), self.cfg.text.rms_norm_eps,
                                         ptr(out), ptr(out_f32), o.stride(0) if o is not None else 0, M, N, K, blocks, waves,
                                         ptr(self.d_amax_v), ptr(self.d_amax_i), self.s)

    def _decode_step_launches(self, B: int):
        """One decode step = 6 launches per layer + 2 (Qwen2VLDecoderLayer TF:559-624, final norm
        TF:839, lm_head TF:1320-1323): [(down_proj slabs +) RMSNorm+QKV+bias+M-RoPE+KV append] -> attention
        partials -> merge -> [o_proj+residual] -> [RMSNorm+gate/up+SiLU*mul] -> [down_proj (+residual | slabs)]."""
        t, L, w, s = self.cfg.text, self.L, self.w, self.s
        H, KVH, hd = t.num_heads, t.num_kv_heads, t.head_dim
        nl = t.num_layers
        x, x_other = self.d_x, self.d_x2   # residual stream: swaps buffers at every deferred reduction
        pending = False                    # down_proj slabs of the previous layer waiting in d_part
        for i in range(nl):
            p = f"llm.{i}."
            # the cache tensor is [layers, max_batch, ...]: hand the kernels layer i's base
            kc, vc = ptr(self.kcache[i]), ptr(self.vtcache[i])
            if self._prefetch_mode == 1:  # diagnostic: serial prefetch of this layer's weights into the Infinity Cache
                a0 = w.layout[p + "ln1.w"][0]
                a1 = w.layout[p + "down.w"][0] + 2 * int(np.prod(w.layout[p + "down.w"][1]))
                L.kr_prefetch(w.arena.data_ptr() + a0, a1 - a0, 512, s)
            joined = None
            if self._prefetch_mode in (2, 3):
                # SECOND GRAPH BRANCH: while this layer's latency-bound chain qkv -> attention ->
                # merge -> o_proj runs (25 us moving 25 MB), a side stream pulls the layer's MLP weights (mode 2: gate/up +
                # down, 82.5 MB of the 2B model; mode 3: down only) into the 256 MB Infinity Cache; the branch joins
                # before the gate/up launch.  In a stream capture the event pair forks / joins the graph.
                first = "gate_up.w" if self._prefetch_mode == 2 else "down.w"
                a0 = w.layout[p + first][0]
                a1 = w.layout[p + "down.w"][0] + 2 * int(np.prod(w.layout[p + "down.w"][1]))
                if self.fp8 and w.has(p + "down.w8"):
                    a0 = w.layout[p + ("gate_up.w8" if self._prefetch_mode == 2 else "down.w8")][0]
                    a1 = w.layout[p + "down.s"][0]
                ef, joined = self._pf_events(i)
                L.kr_event_record(ef, s)
                L.kr_stream_wait_event(self._pf_stream.cuda_stream, ef)
                L.kr_prefetch(w.arena.data_ptr() + a0, a1 - a0, self._pf_blocks, self._pf_stream.cuda_stream)
                L.kr_event_record(joined, self._pf_stream.cuda_stream)
            fast = self.fast_residual
            xacc = self.d_xacc if fast else None     # qkv's workgroup 0 leaves x_new there as f32; o_proj adds into it
            pf = None
            if self._prefetch_mode in (4, 5, 6) and self.narrow_mode:
                # PIGGYBACK PREFETCH (experiment builds): the qkv launch occupies 64 of the 256 CUs; extra workgroups of the
                # same launch pull this layer's down_proj weights (mode 4), gate/up + down (5) or gate/up (6) into the
                # Infinity Cache
                lo = "down.w" if self._prefetch_mode == 4 else "gate_up.w"
                hi = "gate_up.w" if self._prefetch_mode == 6 else "down.w"
                sfx = "8" if (self.fp8 and w.has(p + "down.w8")) else ""
                a0 = w.layout[p + lo + sfx][0]
                a1 = w.layout[p + hi + sfx][0] + (1 if sfx else 2) * int(np.prod(w.layout[p + hi + sfx][1]))
                pf = (w.arena.data_ptr() + a0, a1 - a0, self._pf_blocks)
            if self.narrow_mode:
                ranges = self._row_ranges(B)
                slabs = self.d_part.view(-1)[: 2 * B * t.hidden_size].view(2, B, t.hidden_size)   # as down_proj packs them
                one_slab = self.atomic_slab and self.defer_down
                # this layer's down_proj will ADD into accumulator (i + 1) & 1: the (first) qkv launch zeroes it (it was last
                # read by layer i - 1's qkv launch, which is complete)
                zero = slabs[(i + 1) & 1] if (one_slab and i + 1 < nl) else None
                if B > 16 and self.resnorm_qkv and xacc is None and pf is None:
                    # ABOVE 16 ROWS: the residual sum + RMSNorm run ONCE for the batch (kr_decode_resnorm: bit-identical rows),
                    # then ONE qkv launch over all rows reads its x fragments straight from L2 — instead of every one of the
                    # 64-144 workgroups staging 32 rows of x + slab (294 KB at the 2B width) and, at the 7B width, two launches
                    # over 16-row ranges that stream the weights twice (r3 kernel trace, 7B at 32 rows: 2 x 15.8 us per layer)
                    pin = (slabs[i & 1:(i & 1) + 1] if one_slab else slabs) if pending else None
                    L.kr_decode_resnorm(ptr(x), x.stride(0), ptr(pin), int(pin.shape[0]) if pin is not None else 0, B,
                                        ptr(x_other), x_other.stride(0), ptr(w.view(p + "ln1.w")), t.rms_norm_eps, ptr(self.d_h),
                                        self.d_h.stride(0), B, t.hidden_size, s)
                    if pending:
                        x, x_other = x_other, x
                        pending = False
                    self._dec_narrow(DEC_ROPE_KV, self.d_h, w.view(p + "qkv.w"), B, bias=w.view(p + "qkv.b"), kc=kc, vc=vc, zero=zero,
                                     **self._w8kw(p + "qkv.w"))
                elif pending:
                    for k, (r0, m) in enumerate(ranges):
                        pin = slabs[i & 1:(i & 1) + 1, r0:] if one_slab else slabs[:, r0:]
                        self._dec_narrow(DEC_ROPE_KV, x[r0:], w.view(p + "qkv.w"), m, bias=w.view(p + "qkv.b"),
                                         norm_w=w.view(p + "ln1.w"), part_in=pin, x_out=x_other[r0:], kc=kc, vc=vc,
                                         x_out_f32=None if xacc is None else xacc[r0:], part_rows=B if len(ranges) > 1 else 0,
                                         row0=r0, zero=zero if k == 0 else None, prefetch=pf if k == 0 else None,
                                         **self._w8kw(p + "qkv.w"))
                    x, x_other = x_other, x
                    pending = False
                else:
                    for k, (r0, m) in enumerate(ranges):
                        self._dec_narrow(DEC_ROPE_KV, x[r0:], w.view(p + "qkv.w"), m, bias=w.view(p + "qkv.b"),
                                         norm_w=w.view(p + "ln1.w"), kc=kc, vc=vc, x_out_f32=None if xacc is None else xacc[r0:],
                                         row0=r0, zero=zero if k == 0 else None, prefetch=pf if k == 0 else None,
                                         **self._w8kw(p + "qkv.w"))
            else:
                self._dec(DEC_ROPE_KV, x, w.view(p + "qkv.w"), B, bias=w.view(p + "qkv.b"), norm_w=w.view(p + "ln1.w"),
                          waves=self.wv_qkv, kc=kc, vc=vc)
            if self.attn_fused_merge and not fast:  # the last split workgroup of each (sequence, kv head) merges: no merge launch
                L.kr_attn_decode_fused(ptr(self.d_q), kc, vc, ptr(self.d_ctx), ptr(self.d_o), ptr(self.d_ws), ptr(self.d_cnt),
                                       B, H, KVH, hd, self.s_max, self.n_split, hd ** -0.5, s)
            else:
                L.kr_attn_decode_fused(ptr(self.d_q), kc, vc, ptr(self.d_ctx), 0, ptr(self.d_ws), 0, B, H, KVH, hd,
                                       self.s_max, self.n_split, hd ** -0.5, s)
            if fast:
                # [merge of head h's partials + W_o[:, head h] + atomic add into the f32 residual]: no merge launch
                w8o, sco = self._w8(p + "o.w")
                L.kr_oproj_heads(ptr(self.d_ws), self.n_split, ptr(w8o if w8o is not None else w.view(p + "o.w")), ptr(sco),
                                 ptr(self.d_xacc), self.d_xacc.stride(0), B, t.hidden_size, H, s)
            elif self.attn_fused_merge:
                if self.narrow_o:
                    self._dec_narrow(DEC_PLAIN, self.d_o, w.view(p + "o.w"), B, out=x, res=x, waves=self.o_waves if B <= 16 else 8, **self._w8kw(p + "o.w"))
                else:
                    self._dec(DEC_PLAIN, self.d_o, w.view(p + "o.w"), B, out=x, res=x, waves=self.wv_o)
            elif self.merge_in_o_proj:
                self._dec(DEC_PLAIN, None, w.view(p + "o.w"), B, out=x, res=x, waves=self.wv_o,
                          attn_partials=self.d_ws)
            else:
                L.kr_attn_decode_merge(ptr(self.d_ws), ptr(self.d_o), B, H, hd, self.n_split, s)
                if self.narrow_o:
                    self._dec_narrow(DEC_PLAIN, self.d_o, w.view(p + "o.w"), B, out=x, res=x, waves=self.o_waves if B <= 16 else 8, **self._w8kw(p + "o.w"))
                else:
                    self._dec(DEC_PLAIN, self.d_o, w.view(p + "o.w"), B, out=x, res=x, waves=self.wv_o)
            if joined is not None:
                L.kr_stream_wait_event(s, joined)
            if self._prof_on:
                # [e0][e1] gate/up [e2]: the empty bracket e0..e1 measures what two back-to-back event
                # packets cost by themselves; it is subtracted from the bracket around the launch
                (e0, e1), (e2, _) = self._prof_event_pair(), self._prof_event_pair()
                L.kr_event_record(e0, s)
                L.kr_event_record(e1, s)
            if fast:   # reads the accumulated f32 residual, rounds it once; workgroup 0 leaves the bf16 rows in x
                self._dec_wide_x32(DEC_SILU8, self.d_xacc, x, w.view(p + "gate_up.w"), B, out=self.d_act, norm_w=w.view(p + "ln2.w"),
                                   **self._w8kw(p + "gate_up.w"))
            elif self.wide_mode:
                self._dec_wide(DEC_SILU8, x, w.view(p + "gate_up.w"), B, out=self.d_act, norm_w=w.view(p + "ln2.w"),
                               **self._w8kw(p + "gate_up.w"))
            else:
                self._dec(DEC_SILU8, x, w.view(p + "gate_up.w"), B, out=self.d_act, norm_w=w.view(p + "ln2.w"),
                          waves=self.wv_wide)
            if self._prof_on:
                L.kr_event_record(e2, s)
            if self.defer_down and i + 1 < nl:
                # 2 workgroups per tile; the slabs are added to x by the next layer's qkv prologue
                if self.atomic_slab:
                    acc = self.d_part.view(-1)[: 2 * B * t.hidden_size].view(2, B, t.hidden_size)[(i + 1) & 1]
                    self._dec_narrow(DEC_PLAIN, self.d_act, w.view(p + "down.w"), B, out_f32=acc, waves=self._down_waves(B), ksplit=2,
                                     atomic_out=True, **self._w8kw(p + "down.w"))
                else:
                    self._dec_narrow(DEC_PLAIN, self.d_act, w.view(p + "down.w"), B, out_f32=self.d_part, waves=self._down_waves(B),
                                     ksplit=2, **self._w8kw(p + "down.w"))
                pending = True
            elif self.narrow_mode:
                self._dec_narrow(DEC_PLAIN, self.d_act, w.view(p + "down.w"), B, out=x, res=x, waves=self._down_waves(B),
                                 **self._w8kw(p + "down.w"))
            else:
                self._dec(DEC_PLAIN, self.d_act, w.view(p + "down.w"), B, out=x, res=x, waves=self.wv_down)
            for _ in range(self._extra_nulls):  # diagnostic: price of one more (empty) launch in the chain
                L.kr_launch_null(s)
        self._lm_head_and_sample(B, x)


    def set_fast_residual(self, on: bool):
        """Switch between the deterministic decode step (split-KV merge launch + slab reductions) and the fast-residual
        one (per-head o_proj with float atomics); captured graphs of the other mode are dropped."""
        on = bool(on) and self.narrow_mode and self.wide_mode and not self.row_split and self.L.experiments   # an experiment build only
        if on != self.fast_residual:
            self.stream.synchronize()
            for g in self._graphs.values():
                self.L.kr_graph_destroy(g)
            self._graphs.clear()
            self.fast_residual = on
        return self.fast_residual

