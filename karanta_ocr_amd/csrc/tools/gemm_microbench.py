"""Times kr_gemm_bf16 on the ViT / prefill shapes of the bench workload with both tile geometries.
Run on the GPU box: python karanta_ocr_amd/csrc/tools/gemm_microbench.py"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from karanta_ocr_amd._lib import EPI_NONE, EPI_QUICK_GELU, EPI_SILU_MUL8, lib, ptr  # noqa: E402

L = lib()
dev = "cuda:0"
SCRATCH = torch.zeros(512 * 65536 // 4, dtype=torch.float32, device=dev)
st = torch.cuda.Stream()
S = st.cuda_stream
SHAPES = [("vit qkv", 39200, 3840, 1280, EPI_NONE, 0), ("vit proj", 39200, 1280, 1280, EPI_NONE, 0),
          ("vit fc1", 39200, 5120, 1280, EPI_QUICK_GELU, 0), ("vit fc2", 39200, 1280, 5120, EPI_NONE, 0),
          ("merger fc1", 9800, 5120, 5120, EPI_NONE, 0), ("merger fc2", 9800, 1536, 5120, EPI_NONE, 0),
          ("prefill qkv", 11152, 2048, 1536, EPI_NONE, 1), ("prefill o", 11152, 1536, 1536, EPI_NONE, 1),
          ("prefill gate_up", 11152, 17920, 1536, EPI_SILU_MUL8, 1), ("prefill down", 11152, 1536, 8960, EPI_NONE, 1)]


def run(name, M, N, K, epi, packed, reps=5):
    a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
    w = (torch.rand(N, K, device=dev) * 2 - 1).bfloat16()   # timing only: the packed layout has the same footprint
    nc = N // 2 if epi == EPI_SILU_MUL8 else N
    c = torch.empty(M, nc, device=dev, dtype=torch.bfloat16)
    bias = torch.zeros(N, device=dev, dtype=torch.bfloat16)
    e0, e1 = C.c_void_p(), C.c_void_p()
    L.kr_event_create(C.byref(e0)); L.kr_event_create(C.byref(e1))
    out = {}
    # (label, KARANTA_GEMM_TILE, KARANTA_GEMM_TAIL, KARANTA_GEMM_GROUP_M); the last one is the default configuration
    variants = [("128-tile", 128, 0, 0), ("256-tile", 256, 0, 0), ("pipelined 256 m-major", 512, 0, 0),
                ("+ groups of 4", 512, 0, 4), ("+ groups of 8", 512, 0, 8), ("+ groups of 16", 512, 0, 16),
                ("groups + tail, 2-buffer 128 kernel", 512, 1, None, 2), ("groups + unsplit deep-ring tail", 512, 1, None, 4, 1), ("default (groups where >= 8 n tiles, tail split, deep ring)", 512, 1, None),
                ("automatic tile, 2-buffer", 0, 1, None, 2), ("automatic tile (default)", 0, 1, None),
                ("automatic tile, persistent tile loop", 0, 1, None, 4, 16, 1), ("automatic tile, one tile per workgroup", 0, 1, None, 4, 16, 0)]
    call = lambda: L.kr_gemm_bf16_ws(ptr(a), K, ptr(w), 0 if epi == EPI_SILU_MUL8 else ptr(bias), 0, 0, ptr(c), nc, M, N, K, epi, packed,
                                     ptr(SCRATCH), SCRATCH.numel() * 4, S)   # with the caller-owned split-K scratch, as the engine calls it
    for label, tile, tail, gm, *st in variants:
        if tile:
            os.environ["KARANTA_GEMM_TILE"] = str(tile)
        else:
            os.environ.pop("KARANTA_GEMM_TILE", None)
        if st and st[0] != 4:
            os.environ["KARANTA_GEMM_STAGES"] = str(st[0])
        else:
            os.environ.pop("KARANTA_GEMM_STAGES", None)
        os.environ["KARANTA_GEMM_TAIL_KSPLIT"] = str(st[1]) if len(st) > 1 else "16"
        if len(st) > 2:
            os.environ["KARANTA_GEMM_PERSIST"] = str(st[2])
        else:
            os.environ.pop("KARANTA_GEMM_PERSIST", None)
        os.environ["KARANTA_GEMM_TAIL"] = str(tail)
        if gm is None:
            os.environ.pop("KARANTA_GEMM_GROUP_M", None)
        else:
            os.environ["KARANTA_GEMM_GROUP_M"] = str(gm)
        call(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            L.kr_event_record(e0, S)
            for _ in range(3):
                call()
            L.kr_event_record(e1, S); L.kr_event_synchronize(e1)
            ms = C.c_float(); L.kr_event_elapsed_ms(e0, e1, C.byref(ms)); best = min(best, ms.value / 3)
        out[label] = best
    fl = 2.0 * M * N * K
    print(f"{name:16s} M={M:6d} N={N:6d} K={K:5d} ({-(-M // 256) * (N // 256)} tiles of 256x256): "
          + " | ".join(f"{lb} {t*1e3:7.1f} us {fl/t/1e9:5.0f} TF/s" for lb, t in out.items()), flush=True)


if __name__ == "__main__":
    torch.zeros(1, device=dev)
    if len(sys.argv) > 1 and sys.argv[1] == "stagger":   # KARANTA_GEMM_STAGGER sweep on the default configuration
        import itertools
        for name, M, N, K, epi, packed in SHAPES:
            a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16(); w = (torch.rand(N, K, device=dev) * 2 - 1).bfloat16()
            nc = N // 2 if epi == EPI_SILU_MUL8 else N
            c = torch.empty(M, nc, device=dev, dtype=torch.bfloat16); bias = torch.zeros(N, device=dev, dtype=torch.bfloat16)
            e0, e1 = C.c_void_p(), C.c_void_p(); L.kr_event_create(C.byref(e0)); L.kr_event_create(C.byref(e1))
            res = []
            for sg in (0, 2, 4, 6, 8, 0):
                os.environ["KARANTA_GEMM_STAGGER"] = str(sg)
                call = lambda: L.kr_gemm_bf16_ws(ptr(a), K, ptr(w), 0 if epi == EPI_SILU_MUL8 else ptr(bias), 0, 0, ptr(c), nc, M, N, K, epi,
                                                 packed, ptr(SCRATCH), SCRATCH.numel() * 4, S)
                call(); torch.cuda.synchronize()
                best = 1e9
                for _ in range(5):
                    L.kr_event_record(e0, S)
                    for _ in range(3):
                        call()
                    L.kr_event_record(e1, S); L.kr_event_synchronize(e1)
                    ms = C.c_float(); L.kr_event_elapsed_ms(e0, e1, C.byref(ms)); best = min(best, ms.value / 3)
                res.append(f"stagger {sg}: {best*1e3:7.1f} us {2.0*M*N*K/best/1e9:5.0f} TF/s")
            print(f"{name:16s} " + " | ".join(res), flush=True)
        os.environ.pop("KARANTA_GEMM_STAGGER", None)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "square":   # the shapes GEMM kernels are usually quoted on
        SHAPES = [("4096^3", 4096, 4096, 4096, EPI_NONE, 0), ("8192^3", 8192, 8192, 8192, EPI_NONE, 0),
                  ("16384x4096x4096", 16384, 4096, 4096, EPI_NONE, 0)]
    if len(sys.argv) > 1 and sys.argv[1] == "page":     # one page admitted alone (serving): M = 4900 patches / 1394 tokens
        SHAPES = [("1p vit qkv", 4900, 3840, 1280, EPI_NONE, 0), ("1p vit proj", 4900, 1280, 1280, EPI_NONE, 0),
                  ("1p vit fc1", 4900, 5120, 1280, EPI_QUICK_GELU, 0), ("1p vit fc2", 4900, 1280, 5120, EPI_NONE, 0),
                  ("1p prefill qkv", 1394, 2048, 1536, EPI_NONE, 1), ("1p prefill o", 1394, 1536, 1536, EPI_NONE, 1),
                  ("1p prefill gate_up", 1394, 17920, 1536, EPI_SILU_MUL8, 1), ("1p prefill down", 1394, 1536, 8960, EPI_NONE, 1)]
    for sh in SHAPES:
        run(*sh)
