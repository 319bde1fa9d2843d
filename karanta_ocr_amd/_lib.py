"""ctypes binding of libkaranta_hip.so (include/karanta_hip.h).

The product path has NO CPU fallback: if the shared library is missing or fails to load,
:func:`lib` raises :class:`KarantaHipError`.  `import torch` happens first on purpose — torch
ships its own libamdhip64.so.7 / librccl.so.1 and the extension must bind to the same runtime
instance so that torch's streams and device pointers are valid inside the library.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libkaranta_hip.so")

c_p = C.c_void_p
i32, i64, f32 = C.c_int, C.c_int64, C.c_float


class KarantaHipError(RuntimeError):
    pass


# name -> argtypes (restype is int unless listed in _RESTYPES)
SIGNATURES = {
    "kr_version": [],
    "kr_last_error": [],
    "kr_device_info": [i32, C.c_char_p, C.POINTER(i32), C.POINTER(C.c_size_t)],
    "kr_set_device": [i32],
    "kr_stream_synchronize": [c_p],
    "kr_stream_create_cu_mask": [C.POINTER(c_p), i32],
    "kr_stream_create_cu_range": [C.POINTER(c_p), i32, i32],
    "kr_stream_destroy": [c_p],
    "kr_event_create": [C.POINTER(c_p)],
    "kr_event_destroy": [c_p],
    "kr_event_record": [c_p, c_p],
    "kr_event_synchronize": [c_p],
    "kr_stream_wait_event": [c_p, c_p],
    "kr_event_elapsed_ms": [c_p, c_p, C.POINTER(f32)],
    "kr_graph_begin_capture": [c_p],
    "kr_graph_end_capture": [c_p, C.POINTER(c_p)],
    "kr_graph_launch": [c_p, c_p],
    "kr_graph_destroy": [c_p],
    "kr_cast_pad_f32_bf16": [c_p, c_p, i64, i32, i32, c_p],
    "kr_layernorm": [c_p, c_p, c_p, c_p, i64, i32, f32, c_p],
    "kr_rmsnorm": [c_p, i64, c_p, c_p, i64, i32, f32, c_p],
    "kr_gemm_bf16": [c_p, i64, c_p, c_p, c_p, i64, c_p, i64, i64, i32, i32, i32, i32, c_p],
    "kr_decode_resnorm": [c_p, i64, c_p, i32, i32, c_p, i64, c_p, f32, c_p, i64, i32, i32, c_p],
    "kr_decode_resnorm32": [c_p, i64, c_p, i32, i32, c_p, i64, c_p, f32, c_p, i32, i32, i32, c_p],
    "kr_pack_rows32": [c_p, i64, i32, i32, c_p, c_p],
    "kr_linear_decode32": [i32, c_p, c_p],
    "kr_attn_decode_merge32": [c_p, c_p, i32, i32, i32, i32, c_p],
    "kr_quantize_rows_fp8": [c_p, i64, c_p, i64, c_p, i64, i32, c_p],
    "kr_gemm_fp8a": [c_p, i64, c_p, c_p, c_p, c_p, c_p, i64, c_p, i64, i64, i32, i32, i32, c_p],
    "kr_gemm_bf16_ws": [c_p, i64, c_p, c_p, c_p, i64, c_p, i64, i64, i32, i32, i32, i32, c_p, C.c_size_t, c_p],
    "kr_gemv_bf16": [c_p, i64, c_p, c_p, c_p, i64, c_p, c_p, i64, i32, i32, i32, i32, c_p, f32, c_p],
    "kr_qkv_prep": [c_p, i64, i32, i32, i32, c_p, c_p, c_p, c_p, c_p, c_p, i32, c_p, i64, c_p, i64, c_p, i64,
                    i32, i32, i32, c_p],
    "kr_rope2d_vision": [c_p, c_p, c_p, i64, i32, i32, i64, c_p],
    "kr_attn_varlen": [c_p, c_p, c_p, c_p, c_p, c_p, i32, i64, i32, i32, i32, i64, i64, f32, i32, c_p],
    "kr_attn_varlen_q": [c_p, c_p, c_p, c_p, c_p, c_p, i32, i64, i32, i32, i32, i64, i64, f32, i32, i32, c_p],
    "kr_embed_scatter": [c_p, c_p, c_p, c_p, i64, i32, c_p],
    "kr_mrope": [c_p, c_p, c_p, i64, i32, i32, i64, c_p],
    "kr_kv_append": [c_p, c_p, i64, c_p, c_p, c_p, c_p, i64, i32, i32, i32, i32, i32, c_p],
    "kr_decode_qkv_prep": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, i32, i32, i32, i32, i32, i32, c_p],
    "kr_attn_decode_gqa": [c_p, c_p, c_p, c_p, c_p, c_p, i32, i32, i32, i32, i32, i32, i32, f32, c_p],
    "kr_argmax_embed": [c_p, i64, i32, c_p, i32, c_p, c_p, c_p, c_p, c_p, c_p, i32, i32, i32, c_p, i32, i32, c_p],
    "kr_argmax": [c_p, i64, i32, c_p, i32, c_p],
    "kr_linear_decode": [i32, c_p, i64, c_p, c_p, c_p, f32, c_p, i64, c_p, c_p, i64, i32, i32, i32, i32, i32, i32, c_p, c_p,
                         c_p, i32, c_p, i32, c_p, c_p, c_p, c_p, c_p, i32, i32, i32, c_p, c_p, c_p],
    "kr_linear_decode_wide": [i32, c_p, i64, c_p, c_p, c_p, f32, c_p, i64, c_p, c_p, i64, i32, i32, i32, i32, i32, c_p, c_p, c_p],
    "kr_image_resize_bicubic_u8": [c_p, i32, i32, c_p, i32, i32, c_p, c_p, c_p, i32, c_p, c_p, i32, c_p],
    "kr_image_normalize_patchify": [c_p, i32, i32, c_p, c_p, i32, i32, i32, c_p, c_p],
    "kr_linear_decode_wide_fp8": [i32, c_p, i64, c_p, c_p, c_p, c_p, f32, c_p, i64, c_p, c_p, i64, i32, i32, i32, i32, i32, c_p, c_p, c_p],
    "kr_linear_decode_narrow_fp8": [i32, c_p, i64, c_p, i32, c_p, i64, c_p, c_p, c_p, c_p, f32, c_p, i64, c_p, c_p, i64,
                                    i32, i32, i32, i32, i32, c_p, i32, c_p, c_p, c_p, c_p, c_p, i32, i32, i32, c_p, c_p],
    "kr_fp8_to_bf16": [c_p, c_p, i64, c_p],
    "kr_gemm_fp8": [c_p, i64, c_p, c_p, c_p, c_p, i64, c_p, i64, i64, i32, i32, i32, c_p],
    "kr_gumbel_argmax": [c_p, i64, i32, c_p, c_p, c_p, c_p, c_p, c_p, i32, i32, c_p],
    "kr_gumbel_argmax_guided": [c_p, i64, i32, c_p, c_p, c_p, c_p, c_p, c_p, i32, i32, c_p, c_p, i32, i32, c_p],
    "kr_guide_build_masks": [c_p, c_p, i32, c_p, c_p, i32, c_p, i32, c_p, i32, c_p],
    "kr_guide_advance": [c_p, c_p, c_p, c_p, c_p, c_p, i32, i32, c_p],
    "kr_logprobs_topk": [c_p, i64, i32, i32, i32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, i32, i32, i32, i32, c_p],
    "kr_linear_decode_narrow": [i32, c_p, i64, c_p, i32, c_p, i64, c_p, c_p, c_p, f32, c_p, i64, c_p, c_p, i64,
                                i32, i32, i32, i32, i32, c_p, i32, c_p, c_p, c_p, c_p, c_p, i32, i32, i32, c_p, c_p],
    "kr_attn_decode_fused": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, i32, i32, i32, i32, i32, i32, f32, c_p],
    "kr_attn_decode_slots": [c_p, c_p, c_p, c_p, c_p, c_p, i32, i32, i32, i32, i32, i32, f32, c_p],
    "kr_attn_decode_merge": [c_p, c_p, i32, i32, i32, i32, c_p],
    "kr_sample_greedy": [c_p, c_p, i32, c_p, i32, c_p, c_p, i32, c_p, c_p, c_p, c_p, i32, i32, i32, c_p, i32, c_p],
    "kr_comm_unique_id": [c_p],
    "kr_comm_init": [C.POINTER(c_p), i32, i32, c_p],
    "kr_comm_count": [c_p, C.POINTER(i32)],
    "kr_comm_destroy": [c_p],
    "kr_rccl_version": [C.POINTER(i32)],
    "kr_bcast_weights": [c_p, c_p, C.c_size_t, i32, c_p],
    "kr_selftest_mfma": [c_p],
    "kr_probe_launch_floor": [c_p, i32, i32, i32, C.POINTER(f32)],
    "kr_probe_stream_read": [c_p, C.c_size_t, i32, i32, c_p, C.POINTER(f32)],
    "kr_launch_null": [c_p],
}
_RESTYPES = {"kr_last_error": C.c_char_p}

# include/karanta_hip_experiments.h: exported only by -DKR_EXPERIMENTS builds (csrc/tools/build_variant.py, loaded through
# KARANTA_HIP_LIB); bound when present, absent from the shipped library
EXPERIMENT_SIGNATURES = {
    "kr_linear_decode_narrow_x32": [i32, c_p, i64, c_p, i32, c_p, i64, c_p, i64, c_p, c_p, c_p, c_p, f32, c_p, i64, c_p, c_p, i64,
                                    i32, i32, i32, i32, i32, c_p, i32, c_p, c_p, c_p, c_p, c_p, i32, i32, i32, c_p,
                                    c_p, C.c_size_t, i32, c_p],
    "kr_oproj_heads": [c_p, i32, c_p, c_p, c_p, i64, i32, i32, i32, c_p],
    "kr_linear_decode_wide_x32": [i32, c_p, i64, c_p, i64, c_p, c_p, c_p, f32, c_p, c_p, i64, i32, i32, i32, i32, i32, c_p, c_p, c_p],
    "kr_prefetch": [c_p, C.c_size_t, i32, c_p],
}


class NarrowOpts(C.Structure):
    """kr_narrow_opts (include/karanta_hip.h): the explicit per-launch options of kr_linear_decode_narrow*."""
    _fields_ = [("zero_ptr", c_p), ("zero_bytes", C.c_uint64), ("atomic_out", C.c_int32), ("part_rows", C.c_int32)]


def narrow_opts(zero_ptr: int = 0, zero_bytes: int = 0, atomic_out: bool = False, part_rows: int = 0):
    """A kr_narrow_opts* for one launch (0 = NULL when every field is at its default).  The struct is read during the
    call only, so the temporary may die right after it."""
    if not (zero_bytes or atomic_out or part_rows):
        return None
    return C.byref(NarrowOpts(zero_ptr or None, int(zero_bytes), 1 if atomic_out else 0, int(part_rows)))

class Dec32(C.Structure):
    """kr_dec32 (include/karanta_hip.h): the arguments of kr_linear_decode32."""
    _fields_ = [("xp", c_p), ("w_packed", c_p), ("w_scale", c_p), ("bias", c_p), ("residual", c_p), ("ldr", C.c_int64),
                ("out", c_p), ("out_f32", c_p), ("ldc", C.c_int64), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("waves_ref", C.c_int32), ("ksplit", C.c_int32), ("atomic_out", C.c_int32), ("tiles_per_wg", C.c_int32),
                ("group_split", C.c_int32), ("reserved0", C.c_int32), ("zero_ptr", c_p), ("zero_bytes", C.c_uint64),
                ("cs_table", c_p), ("cs_stride", C.c_int32), ("prompt_len", c_p), ("ctx_len", c_p),
                ("q_out", c_p), ("kcache", c_p), ("vtcache", c_p), ("heads", C.c_int32), ("kv_heads", C.c_int32), ("s_max", C.c_int32)]


ABI_MAJOR = 4              # include/karanta_hip.h KR_ABI_VERSION / 100: the header this binding was written against
DEC_OUT_XP = 0x100         # KR_DEC_OUT_XP
EPI_NONE, EPI_QUICK_GELU, EPI_GELU_ERF, EPI_SILU_MUL, EPI_SILU_MUL8 = 0, 1, 2, 3, 4
DEC_PLAIN, DEC_SILU, DEC_ROPE_KV, DEC_ARGMAX, DEC_SILU8 = 0, 1, 2, 3, 4


class _Lib:
    def __init__(self, path: str):
        if not os.path.exists(path):
            raise KarantaHipError(
                f"{path} not found: build it with `python -m karanta_ocr_amd.build` "
                "(the MI355X engine has no CPU fallback)")
        import torch  # noqa: F401  (loads torch's libamdhip64 / librccl first; see module docstring)

        try:
            self._dll = C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError as e:  # pragma: no cover
            raise KarantaHipError(f"cannot load {path}: {e}") from e
        self.path = path
        for name, argtypes in SIGNATURES.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError as e:
                raise KarantaHipError(f"{path} does not export {name}") from e
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, C.c_int)
            if name in ("kr_version", "kr_last_error"):
                setattr(self, name, fn)
            else:
                setattr(self, name, self._checked(name, fn))
        if self.kr_version() // 100 != ABI_MAJOR:
            raise KarantaHipError(f"{path} reports kr_version() = {self.kr_version()}, this binding is written against ABI major "
                                  f"{ABI_MAJOR} (include/karanta_hip.h): rebuild with `python -m karanta_ocr_amd.build`")
        self.experiments = hasattr(self._dll, "kr_oproj_heads")
        for name, argtypes in EXPERIMENT_SIGNATURES.items():
            if self.experiments:
                fn = getattr(self._dll, name)
                fn.argtypes, fn.restype = argtypes, C.c_int
                setattr(self, name, self._checked(name, fn))
            else:
                setattr(self, name, self._absent(name))

    @staticmethod
    def _absent(name):
        def call(*_):
            raise KarantaHipError(f"{name} is an experiment entry point (include/karanta_hip_experiments.h): this library was built "
                                  "without -DKR_EXPERIMENTS (csrc/tools/build_variant.py; load it through KARANTA_HIP_LIB)")
        call.__name__ = name
        return call

    def _checked(self, name, fn):
        def call(*args):
            rc = fn(*args)
            if rc != 0:
                msg = self._dll.kr_last_error()
                raise KarantaHipError(f"{name} -> {rc}: {msg.decode() if msg else ''}")
            return rc

        call.__name__ = name
        call.raw = fn
        return call


_LIB: Optional[_Lib] = None


def lib() -> _Lib:
    global _LIB
    if _LIB is None:
        _LIB = _Lib(os.environ.get("KARANTA_HIP_LIB", LIB_PATH))
    return _LIB


def ptr(t) -> int:
    """Device/host address of a torch tensor (None -> NULL)."""
    return 0 if t is None else t.data_ptr()
