"""One GEMM shape under every tile geometry, a few launches each — the target of rocprofv3 --pmc passes
(python3 karanta_ocr_amd/csrc/tools/gemm_pmc.py [M N K]); summarise with pmc_sq_summary.py <csv> gemm."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from karanta_ocr_amd._lib import EPI_NONE, lib, ptr  # noqa: E402

M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (39200, 5120, 1280)
L = lib()
dev = "cuda:0"
a = (torch.rand(M, K, device=dev) * 2 - 1).bfloat16()
w = (torch.rand(N, K, device=dev) * 2 - 1).bfloat16()
c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
# launches in this order: 256-tile x3, pipelined m-major x3 (GROUP_M=1), pipelined grouped x3 (default order)
for tile, group in ((256, None), (512, "1"), (512, None)):
    os.environ["KARANTA_GEMM_TILE"] = str(tile)
    os.environ["KARANTA_GEMM_TAIL"] = "0"
    if group is None:
        os.environ.pop("KARANTA_GEMM_GROUP_M", None)
    else:
        os.environ["KARANTA_GEMM_GROUP_M"] = group
    for _ in range(3):
        L.kr_gemm_bf16(ptr(a), K, ptr(w), 0, 0, 0, ptr(c), N, M, N, K, EPI_NONE, 0, 0)
    torch.cuda.synchronize()
