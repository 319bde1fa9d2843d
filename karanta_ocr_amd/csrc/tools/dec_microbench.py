"""Chain-timing of the decode linears: `reps` back-to-back launches inside one hipGraph over rotating weight
copies (more bytes than the 256 MB Infinity Cache, so every launch streams from HBM), total / reps per launch.
Run on the GPU box: python karanta_ocr_amd/csrc/tools/dec_microbench.py [variant ...]"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from karanta_ocr_amd._lib import DEC_PLAIN, DEC_SILU8, DEC_ARGMAX, lib, ptr  # noqa: E402

L = lib()
dev = "cuda:0"
st = torch.cuda.Stream()
S = st.cuda_stream


def time_chain(launch, reps=64, rounds=5):
    g = C.c_void_p()
    launch(0)  # first launch eager (function attributes)
    torch.cuda.synchronize()
    L.kr_graph_begin_capture(S)
    for i in range(reps):
        launch(i)
    L.kr_graph_end_capture(S, C.byref(g))
    e0, e1 = C.c_void_p(), C.c_void_p()
    L.kr_event_create(C.byref(e0)); L.kr_event_create(C.byref(e1))
    best = 1e9
    for _ in range(rounds):
        L.kr_event_record(e0, S)
        L.kr_graph_launch(g, S)
        L.kr_event_record(e1, S)
        L.kr_event_synchronize(e1)
        ms = C.c_float()
        L.kr_event_elapsed_ms(e0, e1, C.byref(ms))
        best = min(best, ms.value)
    L.kr_graph_destroy(g)
    return best * 1e3 / reps


def wide(mode, M, N, K, blocks, waves, norm=True, copies=None):
    nbytes = N * K * 2
    copies = copies or max(2, int(600e6 // nbytes) + 1)
    W = [torch.randint(-3, 3, (N * K,), dtype=torch.int16, device=dev).view(torch.bfloat16) for _ in range(copies)]
    x = torch.randn(M, K, device=dev).bfloat16()
    nw = torch.ones(K, device=dev).bfloat16()
    nc = N // 2 if mode == DEC_SILU8 else N
    out = torch.zeros(M, nc, device=dev).bfloat16()
    av = torch.zeros(M, N // 16, device=dev); ai = torch.zeros(M, N // 16, dtype=torch.int32, device=dev)

    def launch(i):
        L.kr_linear_decode_wide(mode, ptr(x), K, ptr(W[i % copies]), 0, ptr(nw) if norm else 0, 1e-6, 0, 0,
                                0 if mode == DEC_ARGMAX else ptr(out), 0, nc, M, N, K, blocks, waves, ptr(av), ptr(ai), S)
    us = time_chain(launch)
    return us, nbytes / us / 1e6


def persist(mode, M, N, K, waves, max_blocks, norm=True, res=False):
    nbytes = N * K * 2
    copies = max(2, int(600e6 // nbytes) + 1)
    W = [torch.randint(-3, 3, (N * K,), dtype=torch.int16, device=dev).view(torch.bfloat16) for _ in range(copies)]
    x = torch.randn(M, K, device=dev).bfloat16()
    nw = torch.ones(K, device=dev).bfloat16()
    nc = N // 2 if mode == DEC_SILU8 else N
    out = torch.zeros(M, nc, device=dev).bfloat16()
    n_amax = (N // 16 + 1) // 2
    av = torch.zeros(M, n_amax, device=dev); ai = torch.zeros(M, n_amax, dtype=torch.int32, device=dev)

    def launch(i):
        L.kr_linear_decode(mode, ptr(x), K, ptr(W[i % copies]), 0, ptr(nw) if norm else 0, 1e-6, ptr(out) if res else 0,
                           nc if res else 0, 0 if mode == DEC_ARGMAX else ptr(out), 0, nc, M, N, K, waves, max_blocks, 1, 0, 0,
                           0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 64, ptr(av), ptr(ai), S)
    us = time_chain(launch)
    return us, nbytes / us / 1e6


def null_chain():
    def launch(i):
        L.kr_launch_null(S)
    return time_chain(launch)


if __name__ == "__main__":
    torch.zeros(1, device=dev)
    print(f"null kernel chain: {null_chain():.2f} us/launch")
    d, ff, V = 1536, 8960, 151936
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "wide"):
        for norm in (True, False):
            for (blocks, waves) in ((256, 8), (256, 5), (512, 4), (224, 5), (128, 8)):
                us, tb = wide(DEC_SILU8, 8, 2 * ff, d, blocks, waves, norm)
                print(f"wide gate_up  M=8 N={2*ff} K={d} blocks={blocks} waves={waves} norm={norm}: {us:7.2f} us  {tb:5.2f} TB/s", flush=True)
        for n_tiles in (256, 512, 1024, 2048, 4096):
            us, tb = wide(DEC_SILU8, 8, n_tiles * 16, d, 256, 8, True)
            us2, _ = wide(DEC_SILU8, 8, n_tiles * 16, d, 256, 8, False)
            print(f"wide N={n_tiles*16:6d} (tiles/CU {n_tiles/256:.0f}) norm: {us:7.2f} us {tb:5.2f} TB/s   no-norm: {us2:7.2f} us", flush=True)
        for M in (1, 4, 8, 16):
            us, tb = wide(DEC_SILU8, M, 2 * ff, d, 256, 8, True)
            print(f"wide gate_up M={M}: {us:7.2f} us {tb:5.2f} TB/s", flush=True)
        us, tb = wide(DEC_ARGMAX, 8, V, d, 256, 8, True)
        print(f"wide lm_head: {us:7.2f} us {tb:5.2f} TB/s", flush=True)
    if which in ("all", "persist"):
        for (waves, mb) in ((4, 512), (8, 512), (4, 1024)):
            us, tb = persist(DEC_SILU8, 8, 2 * ff, d, waves, mb)
            print(f"persist gate_up waves={waves} max_blocks={mb}: {us:7.2f} us {tb:5.2f} TB/s", flush=True)
        for (waves, mb) in ((16, 512), (8, 512), (16, 96)):
            us, tb = persist(DEC_PLAIN, 8, d, ff, waves, mb, norm=False, res=True)
            print(f"persist down waves={waves} max_blocks={mb}: {us:7.2f} us {tb:5.2f} TB/s", flush=True)
        us, tb = persist(DEC_PLAIN, 8, d, d, 8, 512, norm=False, res=True)
        print(f"persist o_proj waves=8: {us:7.2f} us {tb:5.2f} TB/s", flush=True)
    if which == "ramp":
        for n_tiles in (256, 1120):
            for norm in (True, False):
                us, tb = wide(DEC_SILU8, 8, n_tiles * 16, d, 256, 8, norm)
                print(f"ramp debug={os.environ.get('KARANTA_WIDE_DEBUG', '0')} tiles={n_tiles} norm={norm}: {us:7.2f} us", flush=True)
    if which in ("all", "split"):
        # what a 2- / 3-way K split of down_proj could reach: same bytes, more tiles, shorter K per tile
        for (n, k, waves) in ((d, ff, 16), (2 * d, ff // 2, 16), (2 * d, ff // 2, 8), (3 * d, 2944, 8), (4 * d, ff // 4, 8), (4 * d, ff // 4, 4)):
            us, tb = persist(DEC_PLAIN, 8, n, k, waves, 512, norm=False, res=False)
            print(f"down as N={n} K={k} waves={waves}: {us:7.2f} us {tb:5.2f} TB/s", flush=True)
        for (n, k, waves) in ((2048, d, 8), (2048, d, 4), (4096, 768, 4)):
            us, tb = persist(DEC_PLAIN, 8, n, k, waves, 512, norm=False, res=False)
            print(f"qkv-like N={n} K={k} waves={waves}: {us:7.2f} us {tb:5.2f} TB/s", flush=True)
