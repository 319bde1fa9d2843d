"""ViT GEMMs of 1..8-page admission batches (M = pages x 4900 rows): how the last round of 256x256 tiles is run.
    python karanta_ocr_amd/csrc/tools/gemm_tail_probe.py
Columns: tail rule as shipped | every tail of <= 128 tiles split into 128x128 quarters (KARANTA_GEMM_TAIL_MINK=0) | never (KARANTA_GEMM_TAIL=0)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from karanta_ocr_amd._lib import EPI_NONE, EPI_QUICK_GELU, lib, ptr  # noqa: E402
L = lib(); dev = "cuda:0"
SCRATCH = torch.zeros(512 * 65536 // 4, dtype=torch.float32, device=dev)
st = torch.cuda.Stream(); S = st.cuda_stream


def run(M, N, K, epi, env):
    for k in ("KARANTA_GEMM_TAIL", "KARANTA_GEMM_TAIL_MINK"):
        os.environ.pop(k, None)
    os.environ.update(env)
    a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16(); w = (torch.rand(N, K, device=dev) * 2 - 1).bfloat16()
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16); bias = torch.zeros(N, device=dev, dtype=torch.bfloat16)
    e0, e1 = C.c_void_p(), C.c_void_p(); L.kr_event_create(C.byref(e0)); L.kr_event_create(C.byref(e1))
    call = lambda: L.kr_gemm_bf16_ws(ptr(a), K, ptr(w), ptr(bias), 0, 0, ptr(c), N, M, N, K, epi, 0, ptr(SCRATCH), SCRATCH.numel() * 4, S)
    call(); torch.cuda.synchronize(); best = 1e9
    for _ in range(5):
        L.kr_event_record(e0, S)
        for _ in range(3):
            call()
        L.kr_event_record(e1, S); L.kr_event_synchronize(e1)
        ms = C.c_float(); L.kr_event_elapsed_ms(e0, e1, C.byref(ms)); best = min(best, ms.value / 3)
    return best * 1e3


for name, N, K, epi in (("proj", 1280, 1280, EPI_NONE), ("fc2", 1280, 5120, EPI_NONE), ("qkv", 3840, 1280, EPI_NONE), ("fc1", 5120, 1280, EPI_QUICK_GELU)):
    for pages in range(1, 9):
        M = pages * 4900
        tiles = -(-M // 256) * (N // 256)
        r = [run(M, N, K, epi, e) for e in ({}, {"KARANTA_GEMM_TAIL_MINK": "0"}, {"KARANTA_GEMM_TAIL": "0"})]
        print(f"{name:4s} {pages} pages M {M:6d}: {tiles:5d} tiles = {tiles / 256:5.2f} rounds (tail {tiles % 256:3d})   shipped {r[0]:7.1f} us | all tails split {r[1]:7.1f} | never {r[2]:7.1f}", flush=True)
