import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from karanta_ocr_amd._lib import EPI_NONE, EPI_QUICK_GELU, lib, ptr
L = lib(); dev = "cuda:0"
SCRATCH = torch.zeros(512 * 65536 // 4, dtype=torch.float32, device=dev)
st = torch.cuda.Stream(); S = st.cuda_stream
def run(M, N, K, epi, with_bias=True, with_res=False):
    a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16(); w = (torch.rand(N, K, device=dev) * 2 - 1).bfloat16()
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16); bias = torch.zeros(N, device=dev, dtype=torch.bfloat16)
    r = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    e0, e1 = C.c_void_p(), C.c_void_p(); L.kr_event_create(C.byref(e0)); L.kr_event_create(C.byref(e1))
    call = lambda: L.kr_gemm_bf16_ws(ptr(a), K, ptr(w), ptr(bias) if with_bias else 0, ptr(r) if with_res else 0, N if with_res else 0, ptr(c), N, M, N, K, epi, 0, ptr(SCRATCH), SCRATCH.numel() * 4, S)
    call(); torch.cuda.synchronize(); best = 1e9
    for _ in range(5):
        L.kr_event_record(e0, S)
        for _ in range(3): call()
        L.kr_event_record(e1, S); L.kr_event_synchronize(e1)
        ms = C.c_float(); L.kr_event_elapsed_ms(e0, e1, C.byref(ms)); best = min(best, ms.value / 3)
    return best
for M in (39200, 39168):
    for N in (1280, 2560, 3840, 5120):
        for epi, nm in ((EPI_NONE, "none"), (EPI_QUICK_GELU, "gelu")):
            for bias in (True, False):
                ms = run(M, N, 1280, epi, bias)
                tiles = ((M + 255) // 256) * (N // 256)
                print(f"M {M} N {N} K 1280 epi {nm} bias {int(bias)}: {ms*1e3:7.1f} us  {2.0*M*N*1280/ms/1e9:7.1f} TF/s  tiles {tiles} rounds {tiles/256:.2f}  us/round {ms*1e3/ -(-tiles//256):.1f}", flush=True)
print("residual epilogue, N 1280:", run(39200, 1280, 1280, EPI_NONE, True, True) * 1e3, "us")
