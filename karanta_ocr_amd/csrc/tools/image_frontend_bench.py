"""Host (PIL) image front end + H2D of fp32 patches vs the GPU front end, 8 synthetic 1024x1024 pages.
Run on the GPU box: python karanta_ocr_amd/csrc/tools/image_frontend_bench.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from karanta_ocr_amd import image_processing as IP  # noqa: E402
from karanta_ocr_amd.config import CONFIGS  # noqa: E402
from karanta_ocr_amd.engine import Engine  # noqa: E402

cfg = CONFIGS["tiny"]   # only the front end runs: no weights needed
eng = Engine(cfg, max_batch=2, s_max=256, max_patches=64, max_prompt_tokens=64)
imgs = [IP.synthetic_page(i, 1024, 1024) for i in range(8)]
for rep in range(3):
    t0 = time.perf_counter()
    pvs = [IP.image_to_patches(im)[0] for im in imgs]
    t1 = time.perf_counter()
    dev = torch.from_numpy(np.concatenate(pvs, 0)).to("cuda:0")
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    pix, grids = eng.patches_from_images(imgs)
    eng.stream.synchronize()
    t3 = time.perf_counter()
    same = bool(torch.equal(pix, dev))
    print(f"host: PIL resize + normalise + patchify {1e3*(t1-t0):7.1f} ms, H2D of {dev.numel()*4/1e6:.0f} MB fp32 {1e3*(t2-t1):6.1f} ms"
          f" | GPU front end (H2D of {sum(i.nbytes for i in imgs)/1e6:.0f} MB uint8 + kernels) {1e3*(t3-t2):6.1f} ms | identical: {same}",
          flush=True)
