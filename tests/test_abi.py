"""The C-ABI library loads (no GPU needed) and exports exactly what include/karanta_hip.h declares;
the ctypes signatures in karanta_ocr_amd/_lib.py have the same arity as the header prototypes."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "karanta_hip.h")


def header_prototypes():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"^(?:int|const char\*)\s+(kr_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.M | re.S):
        name, args = m.group(1), m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        protos[name] = n
    return protos


@pytest.fixture(scope="module")
def built_lib():
    from karanta_ocr_amd import build

    return build.build(verbose=False)


def test_header_has_the_survey_operator_set():
    protos = header_prototypes()
    for name in ("kr_layernorm", "kr_rmsnorm", "kr_gemm_bf16", "kr_rope2d_vision", "kr_mrope", "kr_attn_varlen",
                 "kr_kv_append", "kr_attn_decode_gqa", "kr_embed_scatter", "kr_argmax", "kr_bcast_weights",
                 "kr_linear_decode", "kr_attn_decode_fused", "kr_attn_decode_slots", "kr_sample_greedy"):
        assert name in protos, name


def test_library_exports_every_declared_symbol(built_lib):
    dll = ctypes.CDLL(built_lib)
    for name in header_prototypes():
        assert hasattr(dll, name), f"{name} declared in the header but not exported"


def experiment_prototypes():
    src = open(os.path.join(ROOT, "include", "karanta_hip_experiments.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return {m.group(1): len([a for a in m.group(2).split(",") if a.strip()])
            for m in re.finditer(r"^int\s+(kr_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.M | re.S)}


def test_shipped_library_holds_the_product_path_only(built_lib):
    """the measured-and-not-adopted decode experiments (Infinity-Cache prefetch, fast-residual mode,
    in-launch attention merge) live behind -DKR_EXPERIMENTS with their own header; the shipped .so exports none of them,
    no thread-local one-shot setter is left in the ABI, and every exported kr_* symbol is declared in karanta_hip.h."""
    import subprocess
    from karanta_ocr_amd._lib import EXPERIMENT_SIGNATURES
    dll = ctypes.CDLL(built_lib)
    exp = experiment_prototypes()
    assert set(exp) == set(EXPERIMENT_SIGNATURES) and exp
    for name, n in exp.items():
        assert not hasattr(dll, name), f"{name} is an experiment entry point but the shipped library exports it"
        assert len(EXPERIMENT_SIGNATURES[name]) == n, name
    protos = header_prototypes()
    for gone in ("kr_decode_slab_next", "kr_decode_part_rows_next", "kr_decode_prefetch_next"):
        assert gone not in protos and not hasattr(dll, gone)
    out = subprocess.run(["nm", "-D", "--defined-only", built_lib], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("kr_")}
    assert exported == set(protos), (sorted(exported - set(protos)), sorted(set(protos) - exported))


def test_ctypes_signatures_match_header_arity(built_lib):
    from karanta_ocr_amd._lib import SIGNATURES

    protos = header_prototypes()
    assert set(SIGNATURES) == set(protos)
    for name, n in protos.items():
        assert len(SIGNATURES[name]) == n, f"{name}: header has {n} params, ctypes binding {len(SIGNATURES[name])}"


def test_loader_and_version(built_lib):
    from karanta_ocr_amd._lib import lib

    from karanta_ocr_amd._lib import ABI_MAJOR
    assert lib().kr_version() // 100 == ABI_MAJOR


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from karanta_ocr_amd import _lib

    with pytest.raises(_lib.KarantaHipError):
        _lib._Lib(str(tmp_path / "nope.so"))


def test_host_side_argument_validation_needs_no_gpu(built_lib):
    """Shape checks run on the host before any launch, so they are testable without a device."""
    from karanta_ocr_amd._lib import KarantaHipError, lib

    L = lib()
    with pytest.raises(KarantaHipError, match="multiple of 64"):
        L.kr_gemm_bf16(16, 100, 16, 0, 0, 0, 16, 128, 4, 128, 100, 0, 0, 0)
    with pytest.raises(KarantaHipError, match="1..16"):
        L.kr_gemv_bf16(16, 64, 16, 0, 0, 0, 16, 0, 64, 17, 64, 64, 0, 0, 0.0, 0)
    with pytest.raises(KarantaHipError, match="hd=64"):
        L.kr_attn_varlen(16, 16, 16, 16, 16, 16, 1, 1, 1, 1, 64, 0, 0, 1.0, 0, 0)
