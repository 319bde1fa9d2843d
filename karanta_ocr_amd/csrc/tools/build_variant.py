"""Build a VARIANT of libkaranta_hip.so with extra -D switches on one or more sources, for same-box A/B measurements.

    python karanta_ocr_amd/csrc/tools/build_variant.py NAME kr_attention.hip '-DKR_ATTN_VPRE(HD)=0' ...
    python karanta_ocr_amd/csrc/tools/build_variant.py exp kr_decode.hip,kr_selftest.hip -DKR_EXPERIMENTS
        (the decode experiments of rounds 1-2 and their entry points, include/karanta_hip_experiments.h)

writes karanta_ocr_amd/csrc/_build/variants/libkaranta_hip.NAME.so (built artefact: git-ignored, travels with gpurun);
run any tool against it with KARANTA_HIP_LIB=<that path>.  The other objects are the ones of the regular build.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..", "..")))
from karanta_ocr_amd import build as B  # noqa: E402


def main():
    name, srcs, defs = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
    B.build(verbose=False)
    csrc = os.path.abspath(os.path.join(HERE, ".."))
    bdir = os.path.join(csrc, "_build")
    vdir = os.path.join(bdir, "variants")
    os.makedirs(vdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    repl = {}
    for src in srcs:
        obj = os.path.join(vdir, f"{os.path.splitext(src)[0]}.{name}.o")
        subprocess.check_call([hipcc, *B.FLAGS, *defs, "-c", os.path.join(csrc, src), "-o", obj])
        repl[os.path.splitext(src)[0] + ".o"] = obj
    objs = []
    for f in sorted(os.listdir(bdir)):
        if f.endswith(".o"):
            objs.append(repl.get(f, os.path.join(bdir, f)))
    out = os.path.join(vdir, f"libkaranta_hip.{name}.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs, "-L/opt/rocm/lib", "-lrccl"])
    print(out)


if __name__ == "__main__":
    main()
