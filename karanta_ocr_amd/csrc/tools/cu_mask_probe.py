"""How the admission kernels scale on a CU-masked stream (kr_stream_create_cu_mask, first N CUs): the ViT qkv GEMM (M 39200, N 3840, K 1280)
and the ViT attention (8 x 4900 tokens, 16 heads x 80) alone on 256 / 192 / 128 / 64 compute units.
    python karanta_ocr_amd/csrc/tools/cu_mask_probe.py"""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from karanta_ocr_amd import positions as POS  # noqa: E402
from karanta_ocr_amd._lib import EPI_NONE, lib, ptr  # noqa: E402
L = lib(); dev = "cuda:0"
SCRATCH = torch.zeros(512 * 65536 // 4, dtype=torch.float32, device=dev)
M, N, K = 39200, 3840, 1280
a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16(); w = (torch.rand(N, K, device=dev) * 2 - 1).bfloat16()
c = torch.empty(M, N, device=dev, dtype=torch.bfloat16); bias = torch.zeros(N, device=dev, dtype=torch.bfloat16)
lens = [4900] * 8; H, hd = 16, 80; n = sum(lens)
k_row0 = np.concatenate([[0], np.cumsum(lens)[:-1]]); nb = [(x + 63) // 64 for x in lens]; vt0 = np.concatenate([[0], np.cumsum(nb)[:-1]])
plan = POS.make_attn_plan(lens, k_row0, vt0, False, q_block=256)
q = (torch.randn(H, n, hd, device=dev) * 0.5).bfloat16(); kk = (torch.randn(H, n, hd, device=dev) * 0.5).bfloat16()
vt = torch.randn(H, plan.n_vt_blocks, hd, 64, device=dev).bfloat16(); o = torch.zeros(n, H * hd, device=dev).bfloat16()
qb, ql = torch.from_numpy(plan.qblk).to(dev), torch.from_numpy(plan.qblk_len).to(dev)
torch.cuda.synchronize()
e0, e1 = C.c_void_p(), C.c_void_p(); L.kr_event_create(C.byref(e0)); L.kr_event_create(C.byref(e1))
for cus in (256, 192, 128, 64):
    h = C.c_void_p(); L.kr_stream_create_cu_mask(C.byref(h), cus); S = h.value
    gemm = lambda: L.kr_gemm_bf16_ws(ptr(a), K, ptr(w), ptr(bias), 0, 0, ptr(c), N, M, N, K, EPI_NONE, 0, ptr(SCRATCH), SCRATCH.numel() * 4, S)
    attn = lambda: L.kr_attn_varlen_q(ptr(q), ptr(kk), ptr(vt), ptr(o), ptr(qb), ptr(ql), plan.qblk.shape[0], n, H, H, hd, n * hd,
                                      plan.n_vt_blocks * hd * 64, hd ** -0.5, 0, 256, S)
    out = []
    for fn in (gemm, attn):
        fn(); L.kr_stream_synchronize(S); best = 1e9
        for _ in range(3):
            L.kr_event_record(e0, S)
            for _ in range(3):
                fn()
            L.kr_event_record(e1, S); L.kr_event_synchronize(e1)
            ms = C.c_float(); L.kr_event_elapsed_ms(e0, e1, C.byref(ms)); best = min(best, ms.value / 3)
        out.append(best * 1e3)
    print(f"{cus:3d} CUs: vit qkv GEMM {out[0]:8.1f} us   vit attention {out[1]:8.1f} us", flush=True)
    L.kr_stream_destroy(h)
