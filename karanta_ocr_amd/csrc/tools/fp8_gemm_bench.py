"""Prefill GEMMs of the 7B decoder on fp8 weights, three ways: bf16 activations (kr_gemm_fp8: codes converted to bf16 in
registers, bf16 MFMA), W8A8 on v_mfma_f32_16x16x32_fp8_fp8 (KARANTA_FP8_MX=0) and W8A8 on the block-scaled
v_mfma_scale_f32_32x32x64_f8f6f4 (default), with and without the per-token quantisation pass.
Run on the GPU box: python karanta_ocr_amd/csrc/tools/fp8_gemm_bench.py [M]"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from karanta_ocr_amd._lib import EPI_NONE, EPI_SILU_MUL8, lib, ptr  # noqa: E402

L = lib()
dev = "cuda:0"
st = torch.cuda.Stream()
S = st.cuda_stream
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4 * 4988      # BASELINE config 5: 4 pages of 4988 prompt tokens
SHAPES = [("qkv", 4608, 3584, EPI_NONE), ("o_proj", 3584, 3584, EPI_NONE), ("gate_up", 37888, 3584, EPI_SILU_MUL8),
          ("down", 3584, 18944, EPI_NONE)]


def timed(call, reps=4):
    e0, e1 = C.c_void_p(), C.c_void_p()
    L.kr_event_create(C.byref(e0)); L.kr_event_create(C.byref(e1))
    call(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        L.kr_event_record(e0, S)
        for _ in range(3):
            call()
        L.kr_event_record(e1, S); L.kr_event_synchronize(e1)
        ms = C.c_float(); L.kr_event_elapsed_ms(e0, e1, C.byref(ms)); best = min(best, ms.value / 3)
    return best * 1e3


tot = {}
for name, N, K, epi in SHAPES:
    a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
    w8 = torch.randint(0, 120, (N, K), device=dev, dtype=torch.uint8)       # timing only (finite codes)
    ws = torch.ones(N, device=dev, dtype=torch.float32)
    a8 = torch.zeros(M, K, device=dev, dtype=torch.uint8)
    a_s = torch.ones(M, device=dev, dtype=torch.float32)
    nc = N // 2 if epi == EPI_SILU_MUL8 else N
    c = torch.empty(M, nc, device=dev, dtype=torch.bfloat16)
    r = {}
    r["W8A16 (bf16 MFMA)"] = timed(lambda: L.kr_gemm_fp8(ptr(a), K, ptr(w8), ptr(ws), 0, 0, 0, ptr(c), nc, M, N, K, epi, S))
    r["quantise rows"] = timed(lambda: L.kr_quantize_rows_fp8(ptr(a), K, ptr(a8), K, ptr(a_s), M, K, S))
    for mx, two, label in ((0, 0, "W8A8 16x16x32 fp8"), (1, 0, "W8A8 scaled 32x32x64"), (1, 1, "scaled, 2 K-tiles per barrier pair")):
        os.environ["KARANTA_FP8_MX"], os.environ["KARANTA_FP8_MX2"] = str(mx), str(two)
        r[label] = timed(lambda: L.kr_gemm_fp8a(ptr(a8), K, ptr(a_s), ptr(w8), ptr(ws), 0, 0, 0, ptr(c), nc, M, N, K, epi, S))
    os.environ.pop("KARANTA_FP8_MX", None)
    os.environ.pop("KARANTA_FP8_MX2", None)
    fl = 2.0 * M * N * K
    print(f"{name:8s} M={M} N={N} K={K}: " + " | ".join(f"{k} {v:8.1f} us" + (f" {fl / v / 1e6:5.0f} TF/s" if "quant" not in k else "") for k, v in r.items()),
          flush=True)
    for k, v in r.items():
        tot[k] = tot.get(k, 0.0) + v
print("per layer: " + " | ".join(f"{k} {v:8.1f} us" for k, v in tot.items()))
