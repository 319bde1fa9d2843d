"""Timing of the varlen attention kernel on page-sized segments: ViT (hd 80, full, 8 images of 70x70 patches) and decoder
prefill (hd 128, causal GQA 12/2, 8 prompts of 1394 tokens), 128- vs 256-query workgroups.
    python karanta_ocr_amd/csrc/tools/attn_microbench.py"""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from karanta_ocr_amd import positions as POS  # noqa: E402
from karanta_ocr_amd._lib import lib, ptr  # noqa: E402

L = lib()
dev = "cuda:0"
st = torch.cuda.Stream()
S = st.cuda_stream


def run(lens, H, KVH, hd, causal, q_block, reps=10):
    n = sum(lens)
    k_row0 = np.concatenate([[0], np.cumsum(lens)[:-1]])
    nb = [(x + 63) // 64 for x in lens]
    vt0 = np.concatenate([[0], np.cumsum(nb)[:-1]])
    plan = POS.make_attn_plan(lens, k_row0, vt0, causal, q_block=q_block)
    torch.manual_seed(1234)     # the same operands for both block sizes: their outputs must then be bit-identical
    q = (torch.randn(H, n, hd, device=dev) * 0.5).bfloat16()
    k = (torch.randn(KVH, n, hd, device=dev) * 0.5).bfloat16()
    vt = (torch.randn(KVH, plan.n_vt_blocks, hd, 64, device=dev)).bfloat16()
    o = torch.zeros(n, H * hd, device=dev).bfloat16()
    qb, ql = torch.from_numpy(plan.qblk).to(dev), torch.from_numpy(plan.qblk_len).to(dev)
    e0, e1 = C.c_void_p(), C.c_void_p()
    L.kr_event_create(C.byref(e0)); L.kr_event_create(C.byref(e1))

    def launch():
        L.kr_attn_varlen_q(ptr(q), ptr(k), ptr(vt), ptr(o), ptr(qb), ptr(ql), plan.qblk.shape[0], n, H, KVH, hd, n * hd,
                           plan.n_vt_blocks * hd * 64, hd ** -0.5, 1 if causal else 0, q_block, S)
    launch(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        L.kr_event_record(e0, S)
        for _ in range(reps):
            launch()
        L.kr_event_record(e1, S)
        L.kr_event_synchronize(e1)
        ms = C.c_float(); L.kr_event_elapsed_ms(e0, e1, C.byref(ms))
        best = min(best, ms.value / reps)
    flops = sum(4.0 * x * x * hd * H * (0.5 if causal else 1.0) for x in lens)
    stamps(q_block)
    return best, flops / best / 1e9, o


SEGMENTS = ("K reads", "QK^T + max", "exp + PV", "tile -> LDS", "barrier", "load issue")


def stamps(q_block):
    """Diagnostic builds only (tools/build_variant.py ... -DKR_ATTN_STAMPS, KARANTA_HIP_LIB=...): cycles per tile and per
    segment of the middle workgroup's waves.  SHARES are what to read, not totals: the stamps fence overlaps."""
    try:
        fn = C.CDLL(os.environ["KARANTA_HIP_LIB"]).kr_attn_debug_read
    except (KeyError, AttributeError, OSError):
        return
    buf = (C.c_ulonglong * 64)()
    if fn(buf) != 0:
        return
    for w in range(q_block // 32):
        nt = max(1, buf[w * 8 + 6])
        seg = [buf[w * 8 + i] / nt for i in range(6)]
        print(f"      wave {w}: {nt} tiles, s_memtime ticks / tile  " + "  ".join(f"{n} {v:.0f}" for n, v in zip(SEGMENTS, seg))
              + f"  sum {sum(seg):.0f}", flush=True)


def run_prep(lens, H, KVH, hd, reps=10):
    """kr_qkv_prep (rotary + re-layout + V transpose) on the same segments: useful bytes = qkv rows read + q/k/V^T written."""
    n = sum(lens)
    k_row0 = np.concatenate([[0], np.cumsum(lens)[:-1]])
    nb = [(x + 63) // 64 for x in lens]
    vt0 = np.concatenate([[0], np.cumsum(nb)[:-1]])
    plan = POS.make_attn_plan(lens, k_row0, vt0, False)
    ld = (H + 2 * KVH) * hd
    qkv = torch.randn(n, ld, device=dev).bfloat16()
    cos = torch.rand(n, hd, device=dev); sin = torch.rand(n, hd, device=dev)
    q = torch.empty(H, n, hd, device=dev).bfloat16(); k = torch.empty(KVH, n, hd, device=dev).bfloat16()
    vt = torch.empty(KVH, plan.n_vt_blocks, hd, 64, device=dev).bfloat16()
    t_ = lambda a: torch.from_numpy(a).to(dev)
    b0, bn, bk, bv = t_(plan.blk_tok0), t_(plan.blk_ntok), t_(plan.blk_k_row0), t_(plan.blk_vt_blk)
    e0, e1 = C.c_void_p(), C.c_void_p()
    L.kr_event_create(C.byref(e0)); L.kr_event_create(C.byref(e1))

    def launch():
        L.kr_qkv_prep(ptr(qkv), ld, 0, H * hd, (H + KVH) * hd, ptr(cos), ptr(sin), ptr(b0), ptr(bn), ptr(bk), ptr(bv),
                      plan.blk_tok0.shape[0], ptr(q), n * hd, ptr(k), n * hd, ptr(vt), plan.n_vt_blocks * hd * 64, H, KVH, hd, S)
    launch(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        L.kr_event_record(e0, S)
        for _ in range(reps):
            launch()
        L.kr_event_record(e1, S)
        L.kr_event_synchronize(e1)
        ms = C.c_float(); L.kr_event_elapsed_ms(e0, e1, C.byref(ms))
        best = min(best, ms.value / reps)
    return best, 2.0 * n * ld * 2 / best / 1e6


if __name__ == "__main__":
    for name, (lens, H, KVH, hd) in {"qkv_prep vit 8 x 4900, 16 heads x 80": ([4900] * 8, 16, 16, 80),
                                     "qkv_prep prefill 8 x 1394, 12/2 heads x 128": ([1394] * 8, 12, 2, 128)}.items():
        ms, gbs = run_prep(lens, H, KVH, hd)
        print(f"{name:45s}            : {ms:8.3f} ms  {gbs:7.0f} GB/s (activations in + out)", flush=True)
    for name, (lens, H, KVH, hd, causal) in {"vit 8 x 4900, 16 heads x 80": ([4900] * 8, 16, 16, 80, False),
                                              "prefill 8 x 1394, 12/2 heads x 128 causal": ([1394] * 8, 12, 2, 128, True),
                                              "vit 1 x 19276": ([19276], 16, 16, 80, False)}.items():
        outs = {}
        for qb in (128, 256, "256 (4 waves x 64 queries)"):
            # r4: the one-wave-per-SIMD form of the 256-query workgroup (hd 80 only; KARANTA_ATTN_Q64_NOW is read per call)
            os.environ["KARANTA_ATTN_Q64_NOW"] = "1" if isinstance(qb, str) else "0"
            if isinstance(qb, str) and hd != 80:
                continue
            ms, tf, o = run(lens, H, KVH, hd, causal, 256 if isinstance(qb, str) else qb)
            outs[qb] = o
            print(f"{name:45s} q_block {qb}: {ms:8.3f} ms  {tf:7.1f} TFLOP/s", flush=True)
        os.environ["KARANTA_ATTN_Q64_NOW"] = "0"
        print("    max |diff| between the block sizes:", max(float((outs[128].float() - o.float()).abs().max()) for o in outs.values()), flush=True)
