"""A/B timing of the decode step (one process, interleaved rounds: cdna_hip_programming.md §5.4 rule 24).

    python karanta_ocr_amd/csrc/tools/decode_ab.py [--model Qwen2-VL-2B] [--batch 8] [--ctx 1906] [--steps 64] \
        [--rounds 5] variant [variant ...]

A variant is a comma-separated list of Engine attribute overrides, e.g. `base`, `attn_fused_merge=1`,
`attn_fused_merge=1,n_split=4`.  The weight arena is filled with random bf16 values ON THE DEVICE (timing does not depend
on the values), the decode state is set up directly (every sequence at context `ctx`), and each variant's decode step
is captured in a hipGraph and replayed `steps` times between two HIP events; rounds alternate between the variants.
Prints per variant: median and min ms/step, and the HBM-roofline fraction of the step."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from karanta_ocr_amd._lib import lib  # noqa: E402
from karanta_ocr_amd.config import CONFIGS  # noqa: E402
from karanta_ocr_amd.engine import Engine  # noqa: E402
from karanta_ocr_amd.csrc.tools.experiment_engine import ExperimentEngine  # noqa: E402

EXPERIMENT_KEYS = {"fast_residual", "attn_fused_merge", "merge_in_o_proj", "_prefetch_mode", "_extra_nulls", "experiment"}
EXPERIMENT_ENV = {"KARANTA_PREFETCH", "KARANTA_FAST_RESIDUAL", "KARANTA_ATTN_FUSED", "KARANTA_MERGE_IN_OPROJ", "KARANTA_EXTRA_NULLS"}


def fill_random(arena: torch.Tensor):
    """Small random bf16 values in every 2-byte slot (fp8 codes / f32 scales of an fp8 arena come out as garbage of
    harmless magnitude: timing only)."""
    n = arena.numel() // 2
    view = arena[: 2 * n].view(torch.bfloat16)
    step = 1 << 26
    for i in range(0, n, step):
        m = min(step, n - i)
        view[i:i + m] = (torch.randn(m, device=arena.device) * 0.02).to(torch.bfloat16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="Qwen2-VL-2B")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--ctx", type=int, default=1906, help="cached tokens per sequence (bench mean: P + T_out / 2)")
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--weights", default="bf16")
    ap.add_argument("--chain", default=None,
                    help="instead of whole steps: time ONE kind of launch as a chain over the layers' weights (graph of reps x "
                         "layers launches; the per-launch figure includes the dependent-launch gap): comma-separated list of "
                         "qkv,qkv0,attn,merge,o,oheads,gateup,gateup32,down,merge+o; the 17..32-row packed family: resnorm32,qkv32,"
                         "merge32,o32,gateupxp,down32,down32t1,down32t2,down32a,down32gs,down32gs2,down32gs4; round 3's > 16-row forms: resnorm,qkvd")
    ap.add_argument("--cus", default=None,
                    help="comma-separated CU counts: replay each variant's graph on a stream masked to the first N compute units "
                         "(kr_stream_create_cu_mask) — what a decode step keeps when part of the chip is given to something else")
    ap.add_argument("--concurrent", action="store_true",
                    help="launch the variants' graphs CONCURRENTLY, each engine on its own stream (e.g. `--batch 16 --concurrent base base` "
                         "= two independent 16-row decode batches side by side), and report the wall time per step of the group")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    cfg = CONFIGS[a.model]
    L = lib()
    B = a.batch
    s_max = (a.ctx + a.steps * (a.rounds + 2) + 128) // 64 * 64
    engines = []
    base = None
    for spec in a.variants:
        over = {}
        env = {}
        for kv in (spec.split(",") if spec != "base" else []):
            k, v = kv.split("=")
            if k.startswith("KARANTA_"):
                env[k] = v
            else:
                over[k] = v
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        cls = ExperimentEngine if (EXPERIMENT_KEYS & set(over) or EXPERIMENT_ENV & set(env)) else Engine
        over.pop("experiment", None)
        eng = cls(cfg, max_batch=B, s_max=s_max, max_patches=64, max_prompt_tokens=64,
                  decode_splits=int(over.pop("n_split", 16)), weight_dtype=a.weights)
        over.pop("steps_per_graph", None)     # handled where the graph is captured
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        if base is None:
            eng.w.allocate()
            fill_random(eng.w.arena)
            base = eng
        else:
            eng.w = base.w          # same arena: one copy of the weights for every variant
        for k, v in over.items():
            cur = getattr(eng, k)
            setattr(eng, k, type(cur)(int(v)) if isinstance(cur, (bool, int)) else type(cur)(v))
        eng._ignore_eos, eng._freeze_finished, eng._want_logits = True, False, False
        eng._ensure_history(a.steps * (a.rounds + 2) + 8)
        eng._req_max_new = eng.max_new
        engines.append((spec, eng))

    def reset(eng):
        with torch.cuda.stream(eng.stream):
            eng.d_ctx.fill_(a.ctx)          # the state a decode step finds: ctx tokens cached, decode position ctx - plen = 0
            eng.d_plen.fill_(a.ctx)
            eng.d_fin.zero_()
            eng.d_x.copy_((torch.randn(eng.d_x.shape, device=eng.device) * 0.5).to(torch.bfloat16))
        eng.stream.synchronize()

    if a.chain:
        return chains(a, engines, reset)
    graphs, per_graph = {}, {}
    for spec, eng in engines:
        reset(eng)
        k = 1
        for kv in spec.split(","):
            if kv.startswith("steps_per_graph="):
                k = int(kv.split("=")[1])
        per_graph[spec] = k
        with torch.cuda.stream(eng.stream):
            eng._decode_step_launches(B)          # eager once (function attributes), then captured
            eng.stream.synchronize()
            if k == 1:
                graphs[spec] = eng._graph_for(B)
            else:                                 # k consecutive steps in ONE graph (the state lives on the device)
                L.kr_graph_begin_capture(eng.s)
                for _ in range(k):
                    eng._decode_step_launches(B)
                g = C.c_void_p()
                L.kr_graph_end_capture(eng.s, C.byref(g))
                graphs[spec] = g.value
    if a.concurrent:
        import time
        kvb = cfg.text.kv_bytes_per_token
        ts = []
        for r in range(a.rounds + 1):
            for _, eng in engines:
                reset(eng)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                for spec, eng in engines:
                    L.kr_graph_launch(graphs[spec], eng.s)
            torch.cuda.synchronize()
            if r:
                ts.append((time.perf_counter() - t0) / a.steps * 1e3)
        n = len(engines)
        roof_ms = (n * cfg.decoder_weight_bytes(a.weights) + n * B * (a.ctx + a.steps / 2) * kvb) / 8e12 * 1e3
        print(f"{n} x {B} rows side by side: median {np.median(ts):.4f} ms per step of the group, min {min(ts):.4f}  "
              f"({n * B} rows per step; {roof_ms / np.median(ts) * 100:.1f} % of 8 TB/s on the bytes the group streams)", flush=True)
        return
    if a.cus:
        e0, e1 = C.c_void_p(), C.c_void_p()
        L.kr_event_create(C.byref(e0)); L.kr_event_create(C.byref(e1))
        for n in [int(x) for x in a.cus.split(",")]:
            ms_ = C.c_void_p()
            L.kr_stream_create_cu_mask(C.byref(ms_), n)
            for spec, eng in engines:
                ts = []
                for r in range(a.rounds + 1):
                    reset(eng)
                    L.kr_event_record(e0, ms_)
                    for _ in range(a.steps // per_graph[spec]):
                        L.kr_graph_launch(graphs[spec], ms_)
                    L.kr_event_record(e1, ms_)
                    L.kr_event_synchronize(e1)
                    ms = C.c_float()
                    L.kr_event_elapsed_ms(e0, e1, C.byref(ms))
                    if r:
                        ts.append(ms.value / a.steps)
                print(f"{spec:30s} on {n:3d} CUs: median {np.median(ts):.4f} ms/step  min {min(ts):.4f}", flush=True)
            L.kr_stream_destroy(ms_)
        return
    e0, e1 = C.c_void_p(), C.c_void_p()
    L.kr_event_create(C.byref(e0)); L.kr_event_create(C.byref(e1))
    times = {spec: [] for spec, _ in engines}
    for r in range(a.rounds + 1):
        for spec, eng in engines:
            reset(eng)
            L.kr_event_record(e0, eng.s)
            for _ in range(a.steps // per_graph[spec]):
                L.kr_graph_launch(graphs[spec], eng.s)
            L.kr_event_record(e1, eng.s)
            L.kr_event_synchronize(e1)
            ms = C.c_float()
            L.kr_event_elapsed_ms(e0, e1, C.byref(ms))
            if r:                                  # round 0 warms up
                times[spec].append(ms.value / a.steps)
    kvb = cfg.text.kv_bytes_per_token
    roof_ms = (cfg.decoder_weight_bytes(a.weights) + B * (a.ctx + a.steps / 2) * kvb) / 8e12 * 1e3
    for spec, _ in engines:
        t = np.asarray(times[spec])
        print(f"{spec:50s} median {np.median(t):.4f} ms/step  min {t.min():.4f}  ({roof_ms / np.median(t) * 100:.1f} % of the 8 TB/s step roofline)",
              flush=True)


def chains(a, engines, reset):
    """Per-launch time of single kinds of decode launches (see --chain)."""
    from karanta_ocr_amd._lib import DEC_OUT_XP, DEC_PLAIN, DEC_ROPE_KV, DEC_SILU8, ptr
    L = lib()
    B = a.batch
    e0, e1 = C.c_void_p(), C.c_void_p()
    L.kr_event_create(C.byref(e0)); L.kr_event_create(C.byref(e1))
    for spec, eng in engines:
        t, w = eng.cfg.text, eng.w
        H, KVH, hd = t.num_heads, t.num_kv_heads, t.head_dim

        def one(kind, i):
            p = f"llm.{i}."
            kc, vc = ptr(eng.kcache[i]), ptr(eng.vtcache[i])
            if kind == "qkv":
                eng._dec_narrow(DEC_ROPE_KV, eng.d_x, w.view(p + "qkv.w"), B, bias=w.view(p + "qkv.b"), norm_w=w.view(p + "ln1.w"),
                                part_in=eng.d_part, x_out=eng.d_x2, kc=kc, vc=vc, **eng._w8kw(p + "qkv.w"))
            elif kind == "qkv0":                   # no pending slabs: the prologue reads x only
                eng._dec_narrow(DEC_ROPE_KV, eng.d_x, w.view(p + "qkv.w"), B, bias=w.view(p + "qkv.b"), norm_w=w.view(p + "ln1.w"),
                                kc=kc, vc=vc, **eng._w8kw(p + "qkv.w"))
            elif kind.startswith("down") and len(kind) > 4 and not kind.startswith("down32"):      # down<ksplit>[w<waves>]: e.g. down3, down4w8
                import re
                m = re.fullmatch(r"down(\d)(?:w(\d+))?", kind)
                ks, wv = int(m.group(1)), int(m.group(2) or 16)
                if not hasattr(eng, "_slabs_x"):
                    eng._slabs_x = torch.zeros(8, eng.B, t.hidden_size, dtype=torch.float32, device=eng.device)
                eng._dec_narrow(DEC_PLAIN, eng.d_act, w.view(p + "down.w"), B, out_f32=eng._slabs_x, waves=wv, ksplit=ks,
                                **eng._w8kw(p + "down.w"))
            elif kind in ("resnorm", "resnorm32"):
                args = (ptr(eng.d_x), eng.d_x.stride(0), ptr(eng.d_part[:1]), 1, B, ptr(eng.d_x2), eng.d_x2.stride(0), ptr(w.view(p + "ln1.w")),
                        t.rms_norm_eps, ptr(eng.d_h))
                if kind == "resnorm32":
                    L.kr_decode_resnorm32(*args, B, t.hidden_size, 0, eng.s)
                else:
                    L.kr_decode_resnorm(*args, eng.d_h.stride(0), B, t.hidden_size, eng.s)
            elif kind == "qkvd":
                eng._dec_narrow(DEC_ROPE_KV, eng.d_h, w.view(p + "qkv.w"), B, bias=w.view(p + "qkv.b"), kc=kc, vc=vc, **eng._w8kw(p + "qkv.w"))
            elif kind == "qkv32":
                eng._dec32(DEC_ROPE_KV, eng.d_h, w.view(p + "qkv.w"), B, 8, bias=w.view(p + "qkv.b"), kc=kc, vc=vc, **eng._w8kw(p + "qkv.w"))
            elif kind == "merge32":
                L.kr_attn_decode_merge32(ptr(eng.d_ws), ptr(eng.d_o), B, H, hd, eng.n_split, eng.s)
            elif kind in ("o32", "o32t1", "o32t2"):
                eng._dec32(DEC_PLAIN, eng.d_o, w.view(p + "o.w"), B, eng.o_waves, out=eng.d_x, res=eng.d_x,
                           tiles_per_wg={"o32": 0, "o32t1": 1, "o32t2": 2}[kind], **eng._w8kw(p + "o.w"))
            elif kind == "gateupxp":
                eng._dec_wide(DEC_SILU8 | DEC_OUT_XP, eng.d_x, w.view(p + "gate_up.w"), B, out=eng.d_act, norm_w=w.view(p + "ln2.w"),
                              **eng._w8kw(p + "gate_up.w"))
            elif kind in ("down32", "down32t1", "down32t2"):
                eng._dec32(DEC_PLAIN, eng.d_act, w.view(p + "down.w"), B, eng.down_waves_small, ksplit=2, out_f32=eng.d_part,
                           tiles_per_wg={"down32": 0, "down32t1": 1, "down32t2": 2}[kind], **eng._w8kw(p + "down.w"))
            elif kind in ("down32a", "down32gs", "down32gs2", "down32gs4"):   # atomic slab(s): the product form / group split
                gs = kind != "down32a"
                acc = eng.d_part.view(-1)[: 2 * B * t.hidden_size].view(2, B, t.hidden_size)
                eng._dec32(DEC_PLAIN, eng.d_act, w.view(p + "down.w"), B, eng.down_waves_small, ksplit=2, out_f32=acc if gs else acc[0],
                           atomic_out=True, group_split=gs, tiles_per_wg={"down32a": 0, "down32gs": 0, "down32gs2": 2, "down32gs4": 4}[kind],
                           **({} if gs else eng._w8kw(p + "down.w")))
            elif kind == "attn":
                L.kr_attn_decode_fused(ptr(eng.d_q), kc, vc, ptr(eng.d_ctx), 0, ptr(eng.d_ws), 0, B, H, KVH, hd, eng.s_max, eng.n_split,
                                       hd ** -0.5, eng.s)
            elif kind == "merge":
                L.kr_attn_decode_merge(ptr(eng.d_ws), ptr(eng.d_o), B, H, hd, eng.n_split, eng.s)
            elif kind == "o":
                eng._dec_narrow(DEC_PLAIN, eng.d_o, w.view(p + "o.w"), B, out=eng.d_x, res=eng.d_x, waves=8, **eng._w8kw(p + "o.w"))
            elif "+" in kind:                      # a group of launches per layer, e.g. attn+merge+o
                for k in kind.split("+"):
                    one(k, i)
            elif kind == "oheads":
                w8o, sco = eng._w8(p + "o.w")
                L.kr_oproj_heads(ptr(eng.d_ws), eng.n_split, ptr(w8o if w8o is not None else w.view(p + "o.w")), ptr(sco),
                                 ptr(eng.d_xacc), eng.d_xacc.stride(0), B, t.hidden_size, H, eng.s)
            elif kind == "gateup":
                eng._dec_wide(DEC_SILU8, eng.d_x, w.view(p + "gate_up.w"), B, out=eng.d_act, norm_w=w.view(p + "ln2.w"),
                              **eng._w8kw(p + "gate_up.w"))
            elif kind == "gateup32":
                eng._dec_wide(DEC_SILU8, None, w.view(p + "gate_up.w"), B, out=eng.d_act, norm_w=w.view(p + "ln2.w"), x_f32=eng.d_xacc,
                              x_out=eng.d_x, **eng._w8kw(p + "gate_up.w"))
            elif kind == "down":
                eng._dec_narrow(DEC_PLAIN, eng.d_act, w.view(p + "down.w"), B, out_f32=eng.d_part, waves=eng._down_waves(B), ksplit=2,
                                **eng._w8kw(p + "down.w"))
            else:
                raise SystemExit(f"unknown chain kind {kind}")

        reset(eng)
        for kind in a.chain.split(","):
            reps = 4
            with torch.cuda.stream(eng.stream):
                for i in range(t.num_layers):
                    one(kind, i)                      # eager once (function attributes)
                eng.stream.synchronize()
                g = C.c_void_p()
                L.kr_graph_begin_capture(eng.s)
                try:
                    for _ in range(reps):
                        for i in range(t.num_layers):
                            one(kind, i)
                finally:
                    L.kr_graph_end_capture(eng.s, C.byref(g))
                best = None
                for _ in range(a.rounds):
                    L.kr_event_record(e0, eng.s)
                    L.kr_graph_launch(g, eng.s)
                    L.kr_event_record(e1, eng.s)
                    L.kr_event_synchronize(e1)
                    ms = C.c_float()
                    L.kr_event_elapsed_ms(e0, e1, C.byref(ms))
                    best = ms.value if best is None else min(best, ms.value)
                L.kr_graph_destroy(g)
            print(f"{spec:30s} chain {kind:16s} {best * 1e3 / (reps * t.num_layers):8.2f} us per launch group", flush=True)


if __name__ == "__main__":
    main()
