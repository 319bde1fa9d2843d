"""Build libkaranta_hip.so (gfx950) and the oracle's C helpers, in-tree.

    python -m karanta_ocr_amd.build [--force]

hipcc cross-compiles without a GPU.  Objects go to karanta_ocr_amd/csrc/_build/, the library
to karanta_ocr_amd/libkaranta_hip.so (git-ignored, but shipped to the GPU box by gpurun).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(CSRC, "_build")
LIB = os.path.join(HERE, "libkaranta_hip.so")
SOURCES = ["kr_api.hip", "kr_elementwise.hip", "kr_gemm.hip", "kr_attention.hip", "kr_decode.hip", "kr_decode32.hip", "kr_image.hip", "kr_guide.hip", "kr_comm.hip",
           "kr_selftest.hip"]
HEADERS = [os.path.join(CSRC, "kr_common.h"), os.path.join(os.path.dirname(HERE), "include", "karanta_hip.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc", "-ffp-contract=fast",
         # leading scalar kernel arguments arrive in SGPRs at wave start (kr_decode.hip: WideHot)
         "-mllvm", "-amdgpu-kernarg-preload-count=16",
         # MFMA results straight into VGPRs: gfx950's register file is unified, but by default the backend keeps MFMA
         # accumulators in the AGPR half and copies them (v_accvgpr_read / _write) for every VALU use — 1680 such copies in
         # kr_attention.hip (112 per 64-key tile of the ViT attention, a quarter of its VALU work), 3298 in kr_gemm.hip;
         # with this option there are none and the kernels need 25-30 % fewer registers (attn_varlen<128> 336 -> 254:
         # two waves per SIMD instead of one; the 128x128 GEMM 184 -> 126)
         "-mllvm", "-amdgpu-mfma-vgpr-form=1"]


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP extension cannot be built on this machine")


def _digest(paths) -> str:
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(BUILD, exist_ok=True)
    hipcc = _hipcc()
    hdr_digest = _digest(HEADERS)
    jobs = []
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(BUILD, src.replace(".hip", ".o"))
        stamp = obj + ".sha"
        dig = _digest([sp]) + hdr_digest
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
            continue
        jobs.append((sp, obj, stamp, dig))

    def compile_one(job):
        sp, obj, stamp, dig = job
        cmd = [hipcc, *FLAGS, "-c", sp, "-o", obj]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {sp}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip() and verbose:
            print(r.stderr, file=sys.stderr)
        with open(stamp, "w") as f:
            f.write(dig)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    if jobs or force or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-L/opt/rocm/lib", "-lrccl"]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
