"""Parity of every HIP kernel against numpy restatements, through the C-ABI (ctypes), on a real
MI355X.  Integer-valued inputs make the MFMA paths bit-exact (any layout slip fails exactly);
random inputs use a stated bf16 tolerance."""
import ctypes as C
import math

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from karanta_ocr_amd import positions as POS  # noqa: E402
from karanta_ocr_amd._lib import (DEC_ARGMAX, DEC_PLAIN, DEC_ROPE_KV, DEC_SILU, DEC_SILU8, EPI_GELU_ERF, EPI_NONE,  # noqa: E402
                                  EPI_QUICK_GELU, EPI_SILU_MUL, EPI_SILU_MUL8, KarantaHipError, lib, narrow_opts, ptr)
from karanta_ocr_amd.weights import bf16_round, pack_w16x64  # noqa: E402
from oracle import qwen2vl_oracle as O  # noqa: E402

DEV = "cuda:0"


@pytest.fixture(scope="module")
def L():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return lib()


def dev_bf16(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV).to(torch.bfloat16).contiguous()


def host(t: torch.Tensor) -> np.ndarray:
    torch.cuda.synchronize()
    return t.float().cpu().numpy()


def rnd(rng, *shape, scale=1.0):
    return bf16_round(rng.standard_normal(shape).astype(np.float32) * np.float32(scale))


def ints(rng, *shape, lo=-1, hi=2):
    return rng.integers(lo, hi, size=shape).astype(np.float32)


def assert_close_bf16(got, ref, rel=2 ** -7, abs_=1e-2, what=""):
    """|got-ref| <= abs_ + rel*|ref| — one bf16 rounding of the result plus accumulation-order noise."""
    err = np.abs(got - ref)
    tol = abs_ + rel * np.abs(ref)
    bad = err > tol
    assert not bad.any(), f"{what}: {bad.sum()} / {bad.size} out of tolerance, max err {err.max():.4g} at " \
                          f"{np.unravel_index(err.argmax(), err.shape)} (ref {ref.flat[err.argmax()]:.4g})"


# ----------------------------------------------------------------------------- self test
def test_mfma_layout_selftest(L):
    L.kr_selftest_mfma(0)


def test_device_info(L):
    name = C.create_string_buffer(64)
    cus, mem = C.c_int(), C.c_size_t()
    L.kr_device_info(0, name, C.byref(cus), C.byref(mem))
    assert b"gfx950" in name.value, name.value
    assert cus.value == 256
    assert mem.value > 200 * 2 ** 30


# ----------------------------------------------------------------------------- elementwise
def test_cast_pad(L):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((37, 1176)).astype(np.float32)
    src = torch.from_numpy(x).to(DEV)
    dst = torch.full((37, 1216), 7.0, dtype=torch.bfloat16, device=DEV)
    L.kr_cast_pad_f32_bf16(ptr(src), ptr(dst), 37, 1176, 1216, 0)
    out = host(dst)
    np.testing.assert_array_equal(out[:, :1176], bf16_round(x))
    assert not out[:, 1176:].any()


@pytest.mark.parametrize("d", [320, 1280, 1536, 3584])
@pytest.mark.parametrize("rows", [1, 5, 130])
def test_layernorm(L, d, rows):
    rng = np.random.default_rng(d + rows)
    x, w, b = rnd(rng, rows, d, scale=2.0) + 0.5, rnd(rng, d) + 1, rnd(rng, d, scale=0.1)
    x = bf16_round(x)
    y = torch.empty(rows, d, dtype=torch.bfloat16, device=DEV)
    xd, wd, bd = dev_bf16(x), dev_bf16(w), dev_bf16(b)  # keep alive: ptr() of a temporary dangles
    L.kr_layernorm(ptr(xd), ptr(wd), ptr(bd), ptr(y), rows, d, 1e-6, 0)
    ref = O.layer_norm(x, bf16_round(w), bf16_round(b), 1e-6)
    assert_close_bf16(host(y), ref, what="layernorm")


@pytest.mark.parametrize("d", [256, 1536, 3584])
def test_rmsnorm_matches_oracle_bit_exact_policy(L, d):
    rng = np.random.default_rng(d)
    rows = 9
    x, w = rnd(rng, rows, d, scale=3.0), bf16_round(rnd(rng, d) * 0.1 + 1)
    big = dev_bf16(np.concatenate([x, np.zeros((rows, 8), np.float32)], 1))  # row stride d+8
    y = torch.empty(rows, d, dtype=torch.bfloat16, device=DEV)
    wd = dev_bf16(w)
    L.kr_rmsnorm(ptr(big), d + 8, ptr(wd), ptr(y), rows, d, 1e-6, 0)
    ref = bf16_round(O.rms_norm(x, w, 1e-6, O._Policy("bf16")))
    got = host(y)
    # identical rounding points; rsqrt / summation order may move a value by one bf16 ulp
    assert_close_bf16(got, ref, rel=2 ** -7, abs_=1e-6, what="rmsnorm")
    assert (got == ref).mean() > 0.98


def test_embed_scatter(L):
    rng = np.random.default_rng(1)
    table, img = rnd(rng, 50, 64), rnd(rng, 6, 64)
    src = np.asarray([3, -1, -2, 49, 0, -6, 7], np.int32)
    out = torch.empty(len(src), 64, dtype=torch.bfloat16, device=DEV)
    sd, td, idv = torch.from_numpy(src).to(DEV), dev_bf16(table), dev_bf16(img)
    L.kr_embed_scatter(ptr(sd), ptr(td), ptr(idv), ptr(out), len(src), 64, 0)
    ref = np.stack([table[i] if i >= 0 else img[-i - 1] for i in src])
    np.testing.assert_array_equal(host(out), ref)


def test_argmax_lowest_index_wins_ties(L):
    rng = np.random.default_rng(2)
    v = 151936
    x = rng.standard_normal((5, v)).astype(np.float32)
    x[0, [100, 90000, 151935]] = 50.0     # tie -> 100
    x[1, 151935] = 60.0                   # last element
    x[2, 0] = 60.0                        # first element
    x[3] = -np.inf
    x[3, 77777] = -1e30
    out = torch.zeros(5, dtype=torch.int32, device=DEV)
    xd = torch.from_numpy(x).to(DEV)
    L.kr_argmax(ptr(xd), v, v, ptr(out), 5, 0)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), x.argmax(1))


def test_argmax_embed_state_machine(L):
    rng = np.random.default_rng(3)
    B, V, d = 3, 512, 64
    table = rnd(rng, V, d)
    dt = dev_bf16(table)
    logits = np.full((B, V), -5.0, np.float32)
    logits[0, 11] = 1; logits[1, 497] = 1; logits[2, 300] = 1     # seq 1 emits EOS 497
    dl = torch.from_numpy(logits).to(DEV)
    tok = torch.zeros(B, dtype=torch.int32, device=DEV)
    hist = torch.full((4, B), -1, dtype=torch.int32, device=DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    ctx = torch.tensor([10, 20, 30], dtype=torch.int32, device=DEV)
    fin = torch.zeros(B, dtype=torch.int32, device=DEV)
    eos = torch.tensor([497, 496], dtype=torch.int32, device=DEV)
    xn = torch.zeros(B, d, dtype=torch.bfloat16, device=DEV)
    for _ in range(2):
        L.kr_argmax_embed(ptr(dl), V, V, ptr(dt), d, ptr(tok), ptr(hist), ptr(step), ptr(ctx), ptr(fin), ptr(eos), 2,
                          496, 0, ptr(xn), B, B, 0)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(hist.cpu().numpy()[:2], [[11, 497, 300], [11, 496, 300]])  # pad after EOS
    np.testing.assert_array_equal(fin.cpu().numpy(), [0, 1, 0])
    np.testing.assert_array_equal(ctx.cpu().numpy(), [12, 22, 32])
    assert int(step.item()) == 2
    np.testing.assert_array_equal(host(xn), table[[11, 496, 300]])


# ----------------------------------------------------------------------------- GEMM
def ref_linear(A, W, bias=None, res=None, epi=EPI_NONE):
    acc = A.astype(np.float64) @ W.astype(np.float64).T
    if epi in (EPI_SILU_MUL, EPI_SILU_MUL8):
        if bias is not None:               # interleaved like the rows: added to both halves before the activation
            acc = acc + bias
        n = W.shape[0]
        grp = 16 if epi == EPI_SILU_MUL else 8
        g = acc.reshape(A.shape[0], n // (2 * grp), 2, grp)
        gate, up = g[:, :, 0].reshape(A.shape[0], -1), g[:, :, 1].reshape(A.shape[0], -1)
        return (gate / (1 + np.exp(-gate)) * up).astype(np.float32)
    if bias is not None:
        acc = acc + bias
    if epi == EPI_QUICK_GELU:
        acc = acc / (1 + np.exp(-1.702 * acc))
    elif epi == EPI_GELU_ERF:
        acc = 0.5 * acc * (1 + np.vectorize(math.erf)(acc / math.sqrt(2)))
    if res is not None:
        acc = acc + res
    return acc.astype(np.float32)


_GEMM_SCRATCH = None


def run_gemm(L, A, W, bias=None, res=None, epi=EPI_NONE, lda_pad=0, packed=False):
    M, K = A.shape
    N = W.shape[0]
    Ad = dev_bf16(np.concatenate([A, np.zeros((M, lda_pad), np.float32)], 1)) if lda_pad else dev_bf16(A)
    Wd = dev_bf16(pack_w16x64(W) if packed else W)
    nc = N // 2 if epi in (EPI_SILU_MUL, EPI_SILU_MUL8) else N
    Cd = torch.full((M, nc), 9.0, dtype=torch.bfloat16, device=DEV)
    bd = dev_bf16(bias) if bias is not None else None
    rd = dev_bf16(res) if res is not None else None
    # with the caller-owned split-K scratch (kr_gemm_bf16_ws), as the engine calls it: long-K tail rounds are cut along K
    global _GEMM_SCRATCH
    if _GEMM_SCRATCH is None:
        _GEMM_SCRATCH = torch.zeros(512 * 65536 // 4, dtype=torch.float32, device=DEV)
    L.kr_gemm_bf16_ws(ptr(Ad), K + lda_pad, ptr(Wd), ptr(bd), ptr(rd), nc if res is not None else 0, ptr(Cd), nc, M, N, K,
                      epi, 1 if packed else 0, ptr(_GEMM_SCRATCH), _GEMM_SCRATCH.numel() * 4, 0)
    return host(Cd)


@pytest.mark.parametrize("M,N,K", [(1, 16, 64), (127, 128, 64), (128, 128, 128), (129, 144, 192), (300, 272, 64),
                                   (257, 1280, 1216)])
def test_gemm_exact_on_integers(L, M, N, K):
    """Entries in {-1,0,1}: every product sum is a small integer, exact in fp32 and (|sum| <= 256) in bf16."""
    rng = np.random.default_rng(M * 1000 + N + K)
    A, W = ints(rng, M, K), ints(rng, N, K)
    # keep |sum| small enough to be bf16-exact: zero out most of K for the big-K case
    if K > 256:
        W[:, 200:] = 0
    got = run_gemm(L, A, W)
    np.testing.assert_array_equal(got, ref_linear(A, W))


@pytest.fixture(params=["256", "512"])
def tile256(request):
    """Force the 256x256 / 8-wave GEMM geometries (chosen automatically only for very large M): 256 = two barriers per
    K-step, 512 = the pipelined kernel (four K = 32 LDS buffers, counted vmcnt, staggered wave rows)."""
    import os
    os.environ["KARANTA_GEMM_TILE"] = request.param
    yield
    os.environ.pop("KARANTA_GEMM_TILE", None)


@pytest.mark.parametrize("M,N,K", [(1, 16, 64), (255, 256, 64), (256, 256, 128), (257, 272, 192), (700, 528, 64),
                                   (513, 1280, 1216), (300, 256, 256), (1024, 512, 320), (2049, 768, 1280)])
def test_gemm_256_tile_exact_on_integers(L, tile256, M, N, K):
    rng = np.random.default_rng(M * 1000 + N + K + 1)
    A, W = ints(rng, M, K), ints(rng, N, K)
    if K > 256:
        W[:, 200:] = 0
    np.testing.assert_array_equal(run_gemm(L, A, W), ref_linear(A, W))
    np.testing.assert_array_equal(run_gemm(L, A, W, packed=True), ref_linear(A, W))


@pytest.mark.parametrize("M,N,K,group", [(700, 2304, 64, 2), (1793, 768, 128, 3), (2049, 2304, 64, 8), (513, 1280, 192, 5)])
def test_gemm_grouped_tile_order_exact_on_integers(L, M, N, K, group):
    """The pipelined kernel walks its tile list in groups of `group` m tiles (L2 reuse per XCD; chosen automatically
    from 8 n tiles up): every tile still computed exactly once, including ragged last groups and the partial M tile."""
    import os
    rng = np.random.default_rng(M + N + K + group)
    A, W = ints(rng, M, K), ints(rng, N, K)
    os.environ["KARANTA_GEMM_TILE"] = "512"
    os.environ["KARANTA_GEMM_GROUP_M"] = str(group)
    try:
        np.testing.assert_array_equal(run_gemm(L, A, W), ref_linear(A, W))
        np.testing.assert_array_equal(run_gemm(L, A, W, packed=True), ref_linear(A, W))
    finally:
        os.environ.pop("KARANTA_GEMM_TILE", None)
        os.environ.pop("KARANTA_GEMM_GROUP_M", None)


@pytest.mark.parametrize("epi", [EPI_NONE, EPI_QUICK_GELU, EPI_GELU_ERF])
def test_gemm_256_tile_epilogues(L, tile256, epi):
    rng = np.random.default_rng(77 + epi)
    M, N, K = 300, 528, 128
    A, W, bias, res = rnd(rng, M, K), rnd(rng, N, K, scale=K ** -0.5), rnd(rng, N, scale=0.1), rnd(rng, M, N)
    assert_close_bf16(run_gemm(L, A, W, bias, res, epi), ref_linear(A, W, bias, res, epi), what=f"256-tile epi {epi}")
    # asymmetric operands: a transposed fragment map cannot hide
    A2 = np.zeros((M, K), np.float32); A2[np.arange(M), np.arange(M) % K] = 1.0
    W2 = (np.arange(N * K).reshape(N, K) % 7 - 3).astype(np.float32)
    np.testing.assert_array_equal(run_gemm(L, A2, W2), ref_linear(A2, W2))


@pytest.mark.parametrize("packed", [False, True])
def test_gemm_256_tile_silu_mul8(L, tile256, packed):
    rng = np.random.default_rng(78)
    M, ff, K = 270, 8 * 70, 128
    A, W = rnd(rng, M, K), rnd(rng, 2 * ff, K, scale=K ** -0.5)
    assert_close_bf16(run_gemm(L, A, W, epi=EPI_SILU_MUL8, packed=packed), ref_linear(A, W, epi=EPI_SILU_MUL8),
                      what="256-tile silu8")


@pytest.mark.parametrize("M,N,K,epi,packed", [(10800, 1536, 64, EPI_NONE, False),        # 258 tiles: 2 in the tail, partial M tile
                                               (11152, 1536, 1536, EPI_GELU_ERF, True),     # prefill o_proj shape: 264 tiles, tail 8
                                               (11152, 1536, 1536, EPI_SILU_MUL8, True),    # the interleaved gate/up epilogue
                                               (9800, 5120, 1536, EPI_QUICK_GELU, False),   # 780 tiles = 3 rounds + 12
                                               (11152, 1536, 4480, EPI_NONE, True),         # long K: the tail is split along K too
                                               (9800, 5120, 5120, EPI_GELU_ERF, False)])    # merger fc1: 12 tail tiles x 5 K ranges
def test_gemm_tail_round_as_quarter_tiles(L, M, N, K, epi, packed):
    """A pipelined-256 launch whose last round of workgroups would be at most half full runs those tiles as 128x128
    quarters in a second launch (launch_gemm_pipe): exact on integers, and the one-launch form's values."""
    import os
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    tiles = -(-M // 256) * (N // 256)
    assert tiles > cus and 0 < tiles % cus <= cus // 2, "shape no longer exercises the tail on this device"
    rng = np.random.default_rng(M + N + K)
    os.environ["KARANTA_GEMM_TILE"] = "512"
    try:
        if epi == EPI_NONE:
            A, W = ints(rng, M, K), ints(rng, N, K)
            np.testing.assert_array_equal(run_gemm(L, A, W), ref_linear(A, W))
        A, W = rnd(rng, M, K), rnd(rng, N, K, scale=K ** -0.5)
        bias = None if epi == EPI_SILU_MUL8 else rnd(rng, N, scale=0.1)
        res = None if epi == EPI_SILU_MUL8 else rnd(rng, M, N)
        got = run_gemm(L, A, W, bias, res, epi, packed=packed)
        os.environ["KARANTA_GEMM_TAIL"] = "0"
        one = run_gemm(L, A, W, bias, res, epi, packed=packed)
        # the same k order and the same epilogue arithmetic; the two instantiations are compiled separately, so an
        # element sitting on a bf16 rounding boundary may land on either side: (almost) all bits equal, none far off
        # (with K >= 4096 the tail's quarter tiles are also cut along K: partials in one launch, their sum in split order
        # + epilogue in the next — an fp32 sum in another association, so a few per mille of the elements sit on the
        # other side of a bf16 boundary)
        assert (got != one).mean() < (2e-2 if K >= 4096 else 1e-5), f"{(got != one).sum()} elements differ from the one-launch form"
        assert_close_bf16(got, one, what="tail split vs one launch")
        os.environ.pop("KARANTA_GEMM_TAIL", None)
        for _ in range(3):      # reproducible: nothing depends on which workgroup finishes first
            np.testing.assert_array_equal(got, run_gemm(L, A, W, bias, res, epi, packed=packed))
        os.environ["KARANTA_GEMM_TAIL_KSPLIT"] = "1"       # the unsplit tail
        assert (run_gemm(L, A, W, bias, res, epi, packed=packed) != one).mean() < 1e-5
        rows = np.r_[0:300, M - 600:M]          # the head and the tail tiles' rows against the host reference
        assert_close_bf16(got[rows], ref_linear(A[rows], W, bias, None if res is None else res[rows], epi), what="tail split")
    finally:
        os.environ.pop("KARANTA_GEMM_TILE", None)
        os.environ.pop("KARANTA_GEMM_TAIL", None)
        os.environ.pop("KARANTA_GEMM_TAIL_KSPLIT", None)


@pytest.mark.parametrize("M,N,K,epi,packed", [(9800, 2560, 192, EPI_NONE, False),          # 390 tiles: 2 rounds on 256 CUs, the second partial
                                               (9800, 5120, 1536, EPI_QUICK_GELU, False),   # 780 tiles = 3 rounds + 12 (tail launch)
                                               (11152, 3072, 1536, EPI_GELU_ERF, True),     # 528 tiles, packed W, bias + residual
                                               (11152, 4096, 1536, EPI_SILU_MUL8, True),    # the interleaved gate/up epilogue (64-byte rows)
                                               (20000, 1280, 1280, EPI_NONE, False)])       # ragged last m tile, 5 n tiles (m-major list)
def test_gemm_persistent_tile_loop_gives_the_one_tile_per_workgroup_values(L, M, N, K, epi, packed):
    """Experiment build, KARANTA_GEMM_PERSIST=1: one workgroup per CU walks the tile list and requests the next tile's first K-tiles
    before the current tile's epilogue (gemm_pipe_kernel<.., PERSIST>): exact on integers, the same k order and epilogue arithmetic as the
    one-tile-per-workgroup launch (compiled separately: an element on a bf16 rounding boundary may land on either side),
    reproducible, and right against the host reference on the first and last tiles' rows."""
    import os
    if "persist" not in os.path.basename(os.environ.get("KARANTA_HIP_LIB", "")):
        pytest.skip("experiment build only: tools/build_variant.py persist kr_gemm.hip -DKR_GEMM_PERSIST_EXPERIMENT, then "
                    "KARANTA_HIP_LIB=karanta_ocr_amd/csrc/_build/variants/libkaranta_hip.persist.so (measured, not adopted: DESIGN §5-r3)")
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert -(-M // 256) * (N // 256) > cus, "shape no longer needs more than one round on this device"
    rng = np.random.default_rng(M + N + K + 7)
    os.environ["KARANTA_GEMM_TILE"] = "512"
    try:
        os.environ["KARANTA_GEMM_PERSIST"] = "1"
        if epi == EPI_NONE:
            A, W = ints(rng, M, K), ints(rng, N, K)
            np.testing.assert_array_equal(run_gemm(L, A, W), ref_linear(A, W))
        A, W = rnd(rng, M, K), rnd(rng, N, K, scale=K ** -0.5)
        bias = None if epi == EPI_SILU_MUL8 else rnd(rng, N, scale=0.1)
        res = None if epi == EPI_SILU_MUL8 else rnd(rng, M, N)
        got = run_gemm(L, A, W, bias, res, epi, packed=packed)
        for _ in range(2):
            np.testing.assert_array_equal(got, run_gemm(L, A, W, bias, res, epi, packed=packed))
        os.environ["KARANTA_GEMM_PERSIST"] = "0"
        one = run_gemm(L, A, W, bias, res, epi, packed=packed)
        assert (got != one).mean() < 1e-5, f"{(got != one).sum()} elements differ from the one-tile-per-workgroup form"
        assert_close_bf16(got, one, what="persistent tile loop vs one tile per workgroup")
        rows = np.r_[0:300, M - 600:M]
        assert_close_bf16(got[rows], ref_linear(A[rows], W, bias, None if res is None else res[rows], epi), what="persistent tile loop")
    finally:
        os.environ.pop("KARANTA_GEMM_TILE", None)
        os.environ.pop("KARANTA_GEMM_PERSIST", None)


def test_gemm_asymmetric_operands_catch_transposes(L):
    M, N, K = 128, 128, 64
    A = np.zeros((M, K), np.float32); A[np.arange(64), np.arange(64)] = 1  # rows 0..63 = identity on K
    W = (np.arange(N)[:, None] % 7 + 2 * (np.arange(K)[None, :] % 5)).astype(np.float32)
    got = run_gemm(L, A, W)
    np.testing.assert_array_equal(got, ref_linear(A, W))


@pytest.mark.parametrize("epi", [EPI_NONE, EPI_QUICK_GELU, EPI_GELU_ERF])
def test_gemm_bias_residual_epilogues(L, epi):
    rng = np.random.default_rng(10 + epi)
    M, N, K = 200, 320, 320
    A, W = rnd(rng, M, K), rnd(rng, N, K, scale=K ** -0.5)
    bias, res = rnd(rng, N, scale=0.1), rnd(rng, M, N)
    got = run_gemm(L, A, W, bias=bias, res=res, epi=epi, lda_pad=64)
    assert_close_bf16(got, ref_linear(A, W, bias, res, epi), what=f"gemm epi {epi}")


def test_gemm_silu_mul_interleaved(L):
    rng = np.random.default_rng(20)
    M, ff, K = 150, 512, 256
    A, Wp = rnd(rng, M, K), rnd(rng, 2 * ff, K, scale=K ** -0.5)
    got = run_gemm(L, A, Wp, epi=EPI_SILU_MUL)
    assert got.shape == (M, ff)
    assert_close_bf16(got, ref_linear(A, Wp, epi=EPI_SILU_MUL), what="gemm silu_mul")


@pytest.mark.parametrize("M,N,K", [(129, 144, 192), (300, 272, 64)])
def test_gemm_packed_weights_exact(L, M, N, K):
    rng = np.random.default_rng(M + N + K + 1)
    A, W = ints(rng, M, K), ints(rng, N, K)
    np.testing.assert_array_equal(run_gemm(L, A, W, packed=True), ref_linear(A, W))


def test_gemm_packed_silu(L):
    rng = np.random.default_rng(22)
    M, ff, K = 150, 512, 256
    A, Wp = rnd(rng, M, K), rnd(rng, 2 * ff, K, scale=K ** -0.5)
    assert_close_bf16(run_gemm(L, A, Wp, epi=EPI_SILU_MUL, packed=True), ref_linear(A, Wp, epi=EPI_SILU_MUL), what="packed silu")


@pytest.mark.parametrize("packed", [False, True])
def test_gemm_silu_mul8(L, packed):
    rng = np.random.default_rng(23)
    M, ff, K = 150, 520, 256          # 2*ff = 1040 = 65 tiles: ragged last N tile
    A, Wp = rnd(rng, M, K), rnd(rng, 2 * ff, K, scale=K ** -0.5)
    got = run_gemm(L, A, Wp, epi=EPI_SILU_MUL8, packed=packed)
    assert got.shape == (M, ff)
    assert_close_bf16(got, ref_linear(A, Wp, epi=EPI_SILU_MUL8), what="gemm silu8")


@pytest.mark.parametrize("packed", [False, True])
def test_gemm_silu_mul8_with_interleaved_bias(L, packed):
    """The biased SwiGLU of the Qwen2.5-VL vision MLP: bias rows interleaved like the weight rows, added before the
    activation on both halves; zero padding rows (weight and bias) give zero features."""
    rng = np.random.default_rng(24)
    M, ff, K = 150, 520, 256
    A, Wp, bias = rnd(rng, M, K), rnd(rng, 2 * ff, K, scale=K ** -0.5), rnd(rng, 2 * ff, scale=0.5)
    Wp[-16:] = 0; bias[-16:] = 0                      # last 8 features: padding
    got = run_gemm(L, A, Wp, bias=bias, epi=EPI_SILU_MUL8, packed=packed)
    assert_close_bf16(got, ref_linear(A, Wp, bias, epi=EPI_SILU_MUL8), what="gemm silu8 + bias")
    assert not got[:, -8:].any()


def test_gemm_in_place_residual(L):
    rng = np.random.default_rng(21)
    M, N, K = 130, 256, 128
    A, W, X = rnd(rng, M, K), rnd(rng, N, K, scale=K ** -0.5), rnd(rng, M, N)
    Xd = dev_bf16(X)
    Ad, Wd = dev_bf16(A), dev_bf16(W)
    L.kr_gemm_bf16(ptr(Ad), K, ptr(Wd), 0, ptr(Xd), N, ptr(Xd), N, M, N, K, EPI_NONE, 0, 0)
    assert_close_bf16(host(Xd), ref_linear(A, W, res=X), what="in-place residual")


def test_gemm_rejects_bad_shapes(L):
    with pytest.raises(KarantaHipError):
        L.kr_gemm_bf16(256, 100, 256, 0, 0, 0, 256, 128, 4, 128, 100, 0, 0, 0)


# ----------------------------------------------------------------------------- GEMV
def run_gemv(L, x, W, bias=None, res=None, epi=EPI_NONE, norm_w=None, f32=False, eps=1e-6):
    M, K = x.shape
    N = W.shape[0]
    nc = N // 2 if epi == EPI_SILU_MUL else N
    xd, Wd = dev_bf16(x), dev_bf16(W)
    out = torch.full((M, nc), 9.0, dtype=torch.float32 if f32 else torch.bfloat16, device=DEV)
    bd = dev_bf16(bias) if bias is not None else None
    rd = dev_bf16(res) if res is not None else None
    nd = dev_bf16(norm_w) if norm_w is not None else None
    L.kr_gemv_bf16(ptr(xd), K, ptr(Wd), ptr(bd), ptr(rd), nc if res is not None else 0, 0 if f32 else ptr(out),
                   ptr(out) if f32 else 0, nc, M, N, K, epi, ptr(nd), eps, 0)
    return host(out)


@pytest.mark.parametrize("M", [1, 3, 8, 16])
@pytest.mark.parametrize("N,K", [(16, 64), (48, 128), (2048, 256), (512, 1536), (96, 8960)])
def test_gemv_exact_on_integers(L, M, N, K):
    rng = np.random.default_rng(M + N + K)
    x, W = ints(rng, M, K), ints(rng, N, K)
    if K > 256:
        W[:, 256:] = 0
        W[:, :256] = np.roll(W[:, :256], 7, axis=1)
        x = np.roll(x, 3, axis=1)
    np.testing.assert_array_equal(run_gemv(L, x, W), ref_linear(x, W))


def test_gemv_k_order_sensitivity(L):
    """x = one-hot at each k in turn picks out column k of W: any k-permutation mismatch between the
    weight and activation fragments shows up as a wrong column."""
    N, K = 32, 256
    W = (np.arange(N)[:, None] * 3 + np.arange(K)[None, :] % 11).astype(np.float32) % 13
    for k0 in (0, 5, 8, 17, 63, 64, 100, 255):
        x = np.zeros((2, K), np.float32); x[0, k0] = 1; x[1, (k0 + 1) % K] = 2
        got = run_gemv(L, x, W)
        np.testing.assert_array_equal(got, ref_linear(x, W))


@pytest.mark.parametrize("M", [1, 8])
def test_gemv_bias_residual_f32(L, M):
    rng = np.random.default_rng(30 + M)
    N, K = 512, 1536
    x, W = rnd(rng, M, K), rnd(rng, N, K, scale=K ** -0.5)
    bias, res = rnd(rng, N, scale=0.1), rnd(rng, M, N)
    assert_close_bf16(run_gemv(L, x, W, bias=bias, res=res), ref_linear(x, W, bias, res), what="gemv bias+res")
    got32 = run_gemv(L, x, W, bias=bias, f32=True)
    np.testing.assert_allclose(got32, ref_linear(x, W, bias), atol=2e-3, rtol=1e-4)


def test_gemv_wide_n_two_tiles_per_block(L):
    rng = np.random.default_rng(31)
    M, N, K = 8, 16 * 2 * 512 + 32, 256      # "wide" path, with a ragged last block
    x, W = rnd(rng, M, K), rnd(rng, N, K, scale=K ** -0.5)
    got32 = run_gemv(L, x, W, f32=True)
    np.testing.assert_allclose(got32, ref_linear(x, W), atol=2e-3, rtol=1e-4)


@pytest.mark.parametrize("M,K", [(8, 1536), (5, 256), (16, 3584)])
def test_gemv_fused_rmsnorm(L, M, K):
    rng = np.random.default_rng(32 + M)
    N = 256
    x, W, nw = rnd(rng, M, K, scale=2.0), rnd(rng, N, K, scale=K ** -0.5), bf16_round(1 + 0.1 * rnd(rng, K))
    xn = bf16_round(O.rms_norm(x, nw, 1e-6, O._Policy("bf16")))
    assert_close_bf16(run_gemv(L, x, W, norm_w=nw), ref_linear(xn, W), what="gemv fused rmsnorm")


def test_gemv_silu_mul(L):
    rng = np.random.default_rng(33)
    M, ff, K = 8, 1024, 1536
    x, Wp = rnd(rng, M, K), rnd(rng, 2 * ff, K, scale=K ** -0.5)
    got = run_gemv(L, x, Wp, epi=EPI_SILU_MUL)
    assert got.shape == (M, ff)
    assert_close_bf16(got, ref_linear(x, Wp, epi=EPI_SILU_MUL), what="gemv silu")
    # big K (x not staged in LDS) + SILU
    K2 = 8960
    x2, W2 = rnd(rng, 2, K2), rnd(rng, 64, K2, scale=K2 ** -0.5)
    assert_close_bf16(run_gemv(L, x2, W2, epi=EPI_SILU_MUL), ref_linear(x2, W2, epi=EPI_SILU_MUL), what="gemv silu bigK")


def test_gemv_in_place_residual(L):
    rng = np.random.default_rng(34)
    M, N, K = 8, 256, 512
    a, W, X = rnd(rng, M, K), rnd(rng, N, K, scale=K ** -0.5), rnd(rng, M, N)
    Xd = dev_bf16(X)
    ad, Wd = dev_bf16(a), dev_bf16(W)
    L.kr_gemv_bf16(ptr(ad), K, ptr(Wd), 0, ptr(Xd), N, ptr(Xd), 0, N, M, N, K, EPI_NONE, 0, 0.0, 0)
    assert_close_bf16(host(Xd), ref_linear(a, W, res=X), what="gemv in-place")


# ----------------------------------------------------------------------------- rotary
@pytest.mark.parametrize("hd", [80, 128])
def test_rope_inplace(L, hd):
    rng = np.random.default_rng(40 + hd)
    n, H = 33, 3
    x = rnd(rng, n, H, hd)
    ang = rng.uniform(0, 6.28, size=(n, hd // 2)).astype(np.float32)
    cos = np.concatenate([np.cos(ang), np.cos(ang)], -1).astype(np.float32)
    sin = np.concatenate([np.sin(ang), np.sin(ang)], -1).astype(np.float32)
    xd = dev_bf16(x.reshape(n, H * hd))
    fn = L.kr_rope2d_vision if hd == 80 else L.kr_mrope
    cd, sd = torch.from_numpy(cos).to(DEV), torch.from_numpy(sin).to(DEV)
    fn(ptr(xd), ptr(cd), ptr(sd), n, H, hd, H * hd, 0)
    ref = x * cos[:, None] + O.rotate_half(x) * sin[:, None]
    assert_close_bf16(host(xd).reshape(n, H, hd), ref, abs_=1e-3, what="rope")


# ----------------------------------------------------------------------------- prep + varlen attention
def np_attention(q, k, v, scale, causal, q_pos0=0):
    """q [H,nq,hd], k/v [KVH,nk,hd] -> [nq, H*hd]; fp64 reference with bf16-rounded P like the kernel/HF."""
    H, nq, hd = q.shape
    KVH = k.shape[0]
    g = H // KVH
    out = np.zeros((nq, H, hd), np.float64)
    for h in range(H):
        s = (q[h].astype(np.float64) @ k[h // g].astype(np.float64).T) * scale
        if causal:
            mask = np.arange(k.shape[1])[None, :] <= (np.arange(nq)[:, None] + q_pos0)
            s = np.where(mask, s, -np.inf)
        p = np.exp(s - s.max(-1, keepdims=True))
        p = p / p.sum(-1, keepdims=True)
        out[:, h] = p @ v[h // g].astype(np.float64)
    return out.reshape(nq, H * hd).astype(np.float32)


def run_prep_attn(L, lens, H, KVH, hd, causal, seed, as_cache=False, s_max=256, q_block=None):
    """Fused qkv rows -> kr_qkv_prep -> kr_attn_varlen, against numpy, for ragged segments."""
    rng = np.random.default_rng(seed)
    n = sum(lens)
    qd, kd = H * hd, KVH * hd
    qkv = rnd(rng, n, qd + 2 * kd)
    ang = rng.uniform(0, 6.28, size=(n, hd // 2)).astype(np.float32)
    cos = np.concatenate([np.cos(ang)] * 2, -1).astype(np.float32)
    sin = np.concatenate([np.sin(ang)] * 2, -1).astype(np.float32)
    if as_cache:
        plan = POS.prefill_attn_plan(lens, list(range(len(lens))), KVH, s_max)
        if q_block is not None and q_block != plan.q_block:
            plan = POS.make_attn_plan(lens, [i * KVH * s_max for i in range(len(lens))], [i * KVH * (s_max // 64) for i in range(len(lens))],
                                      True, q_block=q_block)
        B = len(lens)
        k_out = torch.zeros(B, KVH, s_max, hd, dtype=torch.bfloat16, device=DEV)
        vt_out = torch.zeros(B, KVH, s_max // 64, hd, 64, dtype=torch.bfloat16, device=DEV)
        k_hs, vt_hs = s_max * hd, (s_max // 64) * hd * 64
    else:
        assert KVH == H
        plan = POS.make_attn_plan(lens, np.concatenate([[0], np.cumsum(lens)[:-1]]),
                                  np.concatenate([[0], np.cumsum([(x + 63) // 64 for x in lens])[:-1]]), causal, q_block=q_block)
        k_out = torch.zeros(KVH, n, hd, dtype=torch.bfloat16, device=DEV)
        vt_out = torch.full((KVH, plan.n_vt_blocks, hd, 64), 3.0, dtype=torch.bfloat16, device=DEV)
        k_hs, vt_hs = n * hd, plan.n_vt_blocks * hd * 64
    q_out = torch.zeros(H, n, hd, dtype=torch.bfloat16, device=DEV)
    o = torch.zeros(n, qd, dtype=torch.bfloat16, device=DEV)
    t_ = lambda a: torch.from_numpy(a).to(DEV)
    args = [t_(plan.blk_tok0), t_(plan.blk_ntok), t_(plan.blk_k_row0), t_(plan.blk_vt_blk), t_(plan.qblk), t_(plan.qblk_len)]
    qkv_d, cos_d, sin_d = dev_bf16(qkv), t_(cos), t_(sin)
    L.kr_qkv_prep(ptr(qkv_d), qd + 2 * kd, 0, qd, qd + kd, ptr(cos_d), ptr(sin_d), ptr(args[0]), ptr(args[1]), ptr(args[2]),
                  ptr(args[3]), len(plan.blk_tok0), ptr(q_out), n * hd, ptr(k_out), k_hs, ptr(vt_out), vt_hs, H, KVH, hd, 0)
    scale = hd ** -0.5
    if plan.q_block == 128:      # the entry point without a block size = 128-query work lists
        L.kr_attn_varlen(ptr(q_out), ptr(k_out), ptr(vt_out), ptr(o), ptr(args[4]), ptr(args[5]), plan.qblk.shape[0], n, H,
                         KVH, hd, k_hs, vt_hs, scale, 1 if causal else 0, 0)
    else:
        L.kr_attn_varlen_q(ptr(q_out), ptr(k_out), ptr(vt_out), ptr(o), ptr(args[4]), ptr(args[5]), plan.qblk.shape[0], n, H,
                           KVH, hd, k_hs, vt_hs, scale, 1 if causal else 0, plan.q_block, 0)
    # ---- reference
    q = qkv[:, :qd].reshape(n, H, hd)
    k = qkv[:, qd:qd + kd].reshape(n, KVH, hd)
    v = qkv[:, qd + kd:].reshape(n, KVH, hd)
    qr = bf16_round(q * cos[:, None] + O.rotate_half(q) * sin[:, None])
    kr = bf16_round(k * cos[:, None] + O.rotate_half(k) * sin[:, None])
    got_q = host(q_out)
    assert_close_bf16(got_q, qr.transpose(1, 0, 2), abs_=1e-3, what="prep q")
    ref = np.zeros((n, qd), np.float32)
    off = 0
    for s_i, ln in enumerate(lens):
        sl = slice(off, off + ln)
        if as_cache:
            gk = host(k_out)[s_i, :, :ln]
            gvt = host(vt_out)[s_i]  # [KVH, blocks, hd, 64]
            gv = POS.vt_rows(gvt)[:, :ln]
            assert not POS.vt_rows(host(vt_out)[s_i])[:, ln:((ln + 63) // 64) * 64].any()
        else:
            gk = host(k_out)[:, sl]
            b0 = int(plan.qblk[[i for i in range(len(plan.qblk)) if plan.qblk[i, 0] == off][0], 3])
            nb = (ln + 63) // 64
            gv = POS.vt_rows(host(vt_out)[:, b0:b0 + nb])
            assert not gv[:, ln:].any(), "V^T padding must be zero"
            gv = gv[:, :ln]
        assert_close_bf16(gk, kr[sl].transpose(1, 0, 2), abs_=1e-3, what="prep k")
        np.testing.assert_array_equal(gv, v[sl].transpose(1, 0, 2))
        ref[sl] = np_attention(qr[sl].transpose(1, 0, 2), kr[sl].transpose(1, 0, 2), v[sl].transpose(1, 0, 2), scale, causal)
        off += ln
    assert_close_bf16(host(o), ref, rel=2 ** -6, abs_=2e-2, what="attention out")


@pytest.mark.parametrize("q_block", [128, 256, "4x64"])
@pytest.mark.parametrize("lens", [[24], [64], [130, 5, 200], [129], [1, 63, 65], [257, 300], [4900]])
def test_vit_attention_hd80(L, lens, q_block, monkeypatch):
    """"4x64": the 256-query workgroup as 4 waves x 64 queries (attn_varlen_q64_kernel, one wave per SIMD)."""
    monkeypatch.setenv("KARANTA_ATTN_Q64_NOW", "1" if q_block == "4x64" else "0")
    run_prep_attn(L, lens, H=4, KVH=4, hd=80, causal=False, seed=sum(lens), q_block=256 if q_block == "4x64" else q_block)


@pytest.mark.parametrize("lens", [[4900], [4920, 1408]])
def test_vit_attention_hd80_page_sized_segments(L, lens):
    """The segment lengths of real pages: 70x70 = 4900 patches (1024x1024 scans, 77 KV tiles per query block — the
    lazy-rescale path runs dozens of times), 82x60 = 4920 (tests/sample.jpg) next to a small second image."""
    run_prep_attn(L, lens, H=4, KVH=4, hd=80, causal=False, seed=sum(lens))


def test_prefill_attention_hd128_page_sized_prompt(L):
    """Causal GQA prefill over a page-sized prompt (1394 tokens = the bench's P) next to a short one, 12 q / 2 kv heads."""
    run_prep_attn(L, [1394, 77], H=12, KVH=2, hd=128, causal=True, seed=1394, as_cache=True, s_max=1408)


def test_prefill_attention_hd128_config5_prompt(L):
    """BASELINE.json config 5's prompt: a 1700x2200 scan at max_pixels 12 845 056 is 4819 image tokens, 4988 with the
    chat template around it — 78 causal KV tiles per query block, the 7B model's GQA group of 7 (14 q / 2 kv heads here),
    cache rows of the length that config decodes into (4988 + 128 -> 5184)."""
    run_prep_attn(L, [4988], H=14, KVH=2, hd=128, causal=True, seed=4988, as_cache=True, s_max=5184)


@pytest.mark.parametrize("q_block", [128, 256])
@pytest.mark.parametrize("lens", [[36], [130, 5, 200], [257]])
@pytest.mark.parametrize("H,KVH", [(2, 1), (6, 2), (3, 3)])
def test_prefill_attention_hd128_causal_gqa(L, lens, H, KVH, q_block):
    run_prep_attn(L, lens, H=H, KVH=KVH, hd=128, causal=True, seed=sum(lens) + H, as_cache=True, s_max=320, q_block=q_block)


def test_attention_online_softmax_rescale_branch(L):
    """Force the running max to jump in a late KV tile (cdna_hip_programming.md §5.4 rule 26):
    one key far down the sequence matches one query strongly."""
    rng = np.random.default_rng(77)
    H, hd, n = 1, 80, 300
    q, k, v = rnd(rng, H, n, hd, scale=0.3), rnd(rng, H, n, hd, scale=0.3), rnd(rng, H, n, hd)
    # (the model's scale, hd^-0.5: the spikes below are then 12 |q|^2 hd^-0.5 ~ 10 nats = 14 in the kernel's log2 units, well
    # past the 2^8 rescale threshold, at score magnitudes a trained head can reach.  The hd 80 kernel rounds Q * scale to
    # bf16 once — an error proportional to |score| — so unit scale with 40-nat scores would test a regime it never sees.)
    scale = hd ** -0.5
    k[0, 250] = bf16_round(q[0, 17] * 12)  # spike in tile 3 for query 17
    k[0, 3] = bf16_round(q[0, 200] * 12)   # and an early spike for a query of the second q-block
    qd, kd = dev_bf16(q), dev_bf16(k)
    vpad = np.concatenate([v, np.zeros((H, 320 - n, hd), np.float32)], 1)
    vt = POS.vt_blocks(vpad)                      # [H, 5 blocks, hd, 64]-shaped storage of [2][hd][32] halves
    plan = POS.make_attn_plan([n], [0], [0], False)
    o = torch.zeros(n, hd, dtype=torch.bfloat16, device=DEV)
    t_ = lambda a: torch.from_numpy(a).to(DEV)
    qb, ql = t_(plan.qblk), t_(plan.qblk_len)
    vtd = dev_bf16(vt)
    L.kr_attn_varlen(ptr(qd), ptr(kd), ptr(vtd), ptr(o), ptr(qb), ptr(ql), plan.qblk.shape[0], n, H, H, hd,
                     n * hd, 5 * hd * 64, scale, 0, 0)
    ref = np_attention(q, k, v, scale, False)
    s17 = float(q[0, 17] @ k[0, 250]) * scale * 1.4427
    others = float(np.max(np.delete(q[0, 17] @ k[0].T, 250))) * scale * 1.4427
    assert s17 - max(others, 0.0) > 8.0, "the spike no longer crosses the kernel's rescale threshold"
    assert_close_bf16(host(o), ref, rel=2 ** -6, abs_=2e-2, what="rescale branch")


# ----------------------------------------------------------------------------- decode path
@pytest.mark.parametrize("H,KVH", [(2, 1), (12, 2), (28, 4), (3, 1)])
@pytest.mark.parametrize("ctxs", [[0, 5, 63], [64, 100, 700], [1, 1279, 2047], [2431, 1393, 2559], [4988, 5116, 5183]])
def test_decode_prep_and_attention(L, H, KVH, ctxs):
    rng = np.random.default_rng(H * 100 + sum(ctxs))
    # contexts up to the bench's last step (P = 1394, T_out = 1024 -> 2417 cached tokens) and past it; the last case is
    # BASELINE config 5's decode range (a 4988-token prompt + 128 tokens) up to the last row of a 5184-row cache
    hd, s_max, B, n_split = 128, max(2048, (max(ctxs) + 64) // 64 * 64), len(ctxs), 4
    kc = np.zeros((B, KVH, s_max, hd), np.float32)
    vc = np.zeros((B, KVH, s_max, hd), np.float32)
    for b, c in enumerate(ctxs):
        kc[b, :, :c] = rnd(rng, KVH, c, hd)
        vc[b, :, :c] = rnd(rng, KVH, c, hd)
    vt = POS.vt_blocks(vc)
    kc_d, vt_d = dev_bf16(kc), dev_bf16(vt)
    qkv = rnd(rng, B, (H + 2 * KVH) * hd)
    delta = np.asarray([-3, 0, -1190][:B], np.int32)
    inv = POS.rope_inv_freq(hd, 1e6)
    ctx_d = torch.tensor(ctxs, dtype=torch.int32, device=DEV)
    q_d = torch.zeros(B, H, hd, dtype=torch.bfloat16, device=DEV)
    qkv_d, inv_d, delta_d = dev_bf16(qkv), torch.from_numpy(inv).to(DEV), torch.from_numpy(delta).to(DEV)
    L.kr_decode_qkv_prep(ptr(qkv_d), ptr(inv_d), ptr(ctx_d),
                         ptr(delta_d), ptr(q_d), ptr(kc_d), ptr(vt_d), B, H, KVH, hd, 0, s_max, 0)
    ws = torch.zeros(B * H * n_split * 4 * (hd + 2), dtype=torch.float32, device=DEV)
    o_d = torch.zeros(B, H * hd, dtype=torch.bfloat16, device=DEV)
    scale = hd ** -0.5
    L.kr_attn_decode_gqa(ptr(q_d), ptr(kc_d), ptr(vt_d), ptr(ctx_d), ptr(o_d), ptr(ws), B, H, KVH, hd, 0, s_max, n_split,
                         scale, 0)
    # ---- reference: rope with bf16-rounded cos/sin at position ctx+delta
    q = qkv[:, :H * hd].reshape(B, H, hd)
    k = qkv[:, H * hd:(H + KVH) * hd].reshape(B, KVH, hd)
    v = qkv[:, (H + KVH) * hd:].reshape(B, KVH, hd)
    got_k, got_vt, got_q, got_o = host(kc_d), host(vt_d), host(q_d), host(o_d)
    for b, c in enumerate(ctxs):
        ang = np.float32(c + delta[b]) * inv
        cos = bf16_round(np.concatenate([np.cos(ang)] * 2).astype(np.float32))
        sin = bf16_round(np.concatenate([np.sin(ang)] * 2).astype(np.float32))
        qr = bf16_round(q[b] * cos + O.rotate_half(q[b]) * sin)
        kr = bf16_round(k[b] * cos + O.rotate_half(k[b]) * sin)
        assert_close_bf16(got_q[b], qr, abs_=2e-2, what="decode q rope")   # cos/sin may round differently by 1 bf16 ulp
        assert_close_bf16(got_k[b, :, c], kr, abs_=2e-2, what="decode k append")
        gv = POS.vt_rows(got_vt[b])
        np.testing.assert_array_equal(gv[:, c], v[b])
        np.testing.assert_array_equal(gv[:, :c], vc[b, :, :c])           # earlier columns untouched
        kk = np.concatenate([kc[b, :, :c], got_k[b, :, c:c + 1]], 1)
        vv = np.concatenate([vc[b, :, :c], v[b][:, None]], 1)
        ref = np_attention(got_q[b][:, None], kk, vv, scale, False)
        assert_close_bf16(got_o[b:b + 1], ref, rel=2 ** -6, abs_=2e-2, what=f"decode attention b={b}")


# ----------------------------------------------------------------------------- fused decode step kernels
def dec_call(L, mode, x, ldx, W, M, N, K, out=0, out_f32=0, ldc=0, bias=0, norm_w=0, res=0, ldr=0, waves=4, ksplit=1, ws=0,
             cnt=0, attn=0, attn_split=1, cs=0, cs_stride=0, plen=0, ctx=0, q_out=0, kc=0, vc=0, heads=0, kv_heads=0,
             s_max=64, av=0, ai=0, max_blocks=0):
    L.kr_linear_decode(mode, x, ldx, W, bias, norm_w, 1e-6, res, ldr, out, out_f32, ldc, M, N, K, waves, max_blocks, ksplit, ws, cnt,
                       attn, attn_split, cs, cs_stride, plen, ctx, q_out, kc, vc, heads, kv_heads, s_max, av, ai, 0)


def run_dec(L, mode, x, W, ksplit=1, bias=None, res=None, norm_w=None, f32=False, waves=4, max_blocks=0):
    M, K = x.shape
    N = W.shape[0]
    nc = N // 2 if mode in (DEC_SILU, DEC_SILU8) else N
    xd, Wd = dev_bf16(x), dev_bf16(pack_w16x64(W))
    out = torch.full((M, nc), 9.0, dtype=torch.float32 if f32 else torch.bfloat16, device=DEV)
    bd = dev_bf16(bias) if bias is not None else None
    rd = dev_bf16(res) if res is not None else None
    nd = dev_bf16(norm_w) if norm_w is not None else None
    groups = N // 16
    ws = torch.zeros(groups * max(ksplit, 1) * 2 * 256, dtype=torch.float32, device=DEV)
    cnt = torch.zeros(groups, dtype=torch.int32, device=DEV)
    dec_call(L, mode, ptr(xd), K, ptr(Wd), M, N, K, out=0 if f32 else ptr(out), out_f32=ptr(out) if f32 else 0, ldc=nc,
             bias=ptr(bd), norm_w=ptr(nd), res=ptr(rd), ldr=nc if res is not None else 0, waves=waves, ksplit=ksplit,
             ws=ptr(ws), cnt=ptr(cnt), max_blocks=max_blocks)
    res_ = host(out)
    assert not cnt.cpu().numpy().any(), "arrival counters must be left at zero"
    return res_


@pytest.mark.parametrize("M", [1, 8, 16])
@pytest.mark.parametrize("N,K,ksplit,waves", [(16, 64, 1, 4), (48, 256, 1, 8), (48, 256, 4, 4), (1536, 1536, 3, 4),
                                              (1536, 1536, 1, 8), (96, 8960, 4, 4), (96, 8960, 1, 16), (96, 8960, 7, 8),
                                              (16 * 2 * 1024 + 16, 128, 1, 4)])
def test_linear_decode_plain_exact_on_integers(L, M, N, K, ksplit, waves):
    rng = np.random.default_rng(M + N + K + ksplit)
    x, W = ints(rng, M, K), ints(rng, N, K)
    if K > 256:
        keep = rng.choice(K, 200, replace=False)
        mask = np.zeros(K, bool); mask[keep] = True
        W[:, ~mask] = 0
    np.testing.assert_array_equal(run_dec(L, DEC_PLAIN, x, W, ksplit=ksplit, waves=waves), ref_linear(x, W))


@pytest.mark.parametrize("waves", [4, 8, 16])
def test_linear_decode_k_order(L, waves):
    N, K = 32, 2048
    W = ((np.arange(N)[:, None] * 3 + np.arange(K)[None, :]) % 13).astype(np.float32)
    for k0 in (0, 7, 8, 15, 16, 63, 64, 130, 511, 1999, 2047):
        x = np.zeros((2, K), np.float32); x[0, k0] = 1; x[1, (k0 + 1) % K] = 2
        np.testing.assert_array_equal(run_dec(L, DEC_PLAIN, x, W, waves=waves), ref_linear(x, W))


def test_linear_decode_wide_k_reads_x_from_global(L):
    """M*K too large for the LDS stage (xmode 0): down_proj of the 7B model at batch 16."""
    rng = np.random.default_rng(49)
    M, N, K = 16, 64, 18944
    x, W = rnd(rng, M, K), rnd(rng, N, K, scale=K ** -0.5)
    res = rnd(rng, M, N)
    assert_close_bf16(run_dec(L, DEC_PLAIN, x, W, res=res, waves=16), ref_linear(x, W, res=res), what="dec wide K")


@pytest.mark.parametrize("ksplit,waves", [(1, 4), (3, 4), (1, 8)])
def test_linear_decode_bias_residual_norm(L, ksplit, waves):
    rng = np.random.default_rng(50 + ksplit)
    M, N, K = 8, 512, 1536
    x, W = rnd(rng, M, K, scale=2.0), rnd(rng, N, K, scale=K ** -0.5)
    bias, res, nw = rnd(rng, N, scale=0.1), rnd(rng, M, N), bf16_round(1 + 0.1 * rnd(rng, K))
    xn = bf16_round(O.rms_norm(x, nw, 1e-6, O._Policy("bf16")))
    assert_close_bf16(run_dec(L, DEC_PLAIN, x, W, ksplit=ksplit, bias=bias, res=res, norm_w=nw, waves=waves),
                      ref_linear(xn, W, bias, res), what="dec plain norm+bias+res")
    got32 = run_dec(L, DEC_PLAIN, x, W, ksplit=ksplit, f32=True, waves=waves)
    np.testing.assert_allclose(got32, ref_linear(x, W), atol=3e-3, rtol=1e-4)


@pytest.mark.parametrize("M,K", [(8, 1536), (16, 3584), (3, 256)])
def test_linear_decode_silu_norm(L, M, K):
    rng = np.random.default_rng(60 + M)
    ff = 1024
    x, Wp, nw = rnd(rng, M, K, scale=2.0), rnd(rng, 2 * ff, K, scale=K ** -0.5), bf16_round(1 + 0.1 * rnd(rng, K))
    xn = bf16_round(O.rms_norm(x, nw, 1e-6, O._Policy("bf16")))
    got = run_dec(L, DEC_SILU, x, Wp, norm_w=nw)
    assert got.shape == (M, ff)
    assert_close_bf16(got, ref_linear(xn, Wp, epi=EPI_SILU_MUL), what="dec silu")


@pytest.mark.parametrize("M,K,waves", [(8, 1536, 4), (16, 3584, 8), (3, 256, 4)])
def test_linear_decode_silu8_norm(L, M, K, waves):
    rng = np.random.default_rng(62 + M)
    ff = 1016                          # 127 tiles of 8 features
    x, Wp, nw = rnd(rng, M, K, scale=2.0), rnd(rng, 2 * ff, K, scale=K ** -0.5), bf16_round(1 + 0.1 * rnd(rng, K))
    xn = bf16_round(O.rms_norm(x, nw, 1e-6, O._Policy("bf16")))
    got = run_dec(L, DEC_SILU8, x, Wp, norm_w=nw, waves=waves)
    assert got.shape == (M, ff)
    assert_close_bf16(got, ref_linear(xn, Wp, epi=EPI_SILU_MUL8), what="dec silu8")


@pytest.mark.parametrize("max_blocks", [1, 3, 7, 64])
def test_linear_decode_persistent_workgroups(L, max_blocks):
    """Persistent workgroups walking several tile groups (double-buffered reduction, next-group prefetch)."""
    rng = np.random.default_rng(90 + max_blocks)
    M, N, K = 8, 16 * 37, 512
    x, W = ints(rng, M, K), ints(rng, N, K)
    W[:, 200:] = 0
    np.testing.assert_array_equal(run_dec(L, DEC_PLAIN, x, W, max_blocks=max_blocks, waves=8), ref_linear(x, W))
    # fused norm + SiLU (8-row interleave) + persistent
    ff = 8 * 45
    x2, Wp, nw = rnd(rng, M, K, scale=2.0), rnd(rng, 2 * ff, K, scale=K ** -0.5), bf16_round(1 + 0.1 * rnd(rng, K))
    xn = bf16_round(O.rms_norm(x2, nw, 1e-6, O._Policy("bf16")))
    assert_close_bf16(run_dec(L, DEC_SILU8, x2, Wp, norm_w=nw, max_blocks=max_blocks), ref_linear(xn, Wp, epi=EPI_SILU_MUL8),
                      what="persistent silu8")
    # residual in place + bias, persistent
    bias, res = rnd(rng, N, scale=0.1), rnd(rng, M, N)
    assert_close_bf16(run_dec(L, DEC_PLAIN, x2, rnd(np.random.default_rng(5), N, K, scale=K ** -0.5), bias=bias, res=res,
                              max_blocks=max_blocks),
                      ref_linear(x2, rnd(np.random.default_rng(5), N, K, scale=K ** -0.5), bias, res), what="persistent plain")


def test_linear_decode_in_place_residual_splitk(L):
    rng = np.random.default_rng(61)
    M, N, K = 8, 256, 8960
    a, W, X = rnd(rng, M, K), rnd(rng, N, K, scale=K ** -0.5), rnd(rng, M, N)
    ad, Wd, Xd = dev_bf16(a), dev_bf16(pack_w16x64(W)), dev_bf16(X)
    ws = torch.zeros(16 * 4 * 256, dtype=torch.float32, device=DEV); cnt = torch.zeros(16, dtype=torch.int32, device=DEV)
    for ksplit, waves in ((4, 4), (4, 4), (1, 16)):  # the repeated call reuses the (reset) counters
        Xd = dev_bf16(X)
        dec_call(L, DEC_PLAIN, ptr(ad), K, ptr(Wd), M, N, K, out=ptr(Xd), ldc=N, res=ptr(Xd), ldr=N, waves=waves,
                 ksplit=ksplit, ws=ptr(ws), cnt=ptr(cnt))
        assert_close_bf16(host(Xd), ref_linear(a, W, res=X), what="dec in-place")


@pytest.mark.parametrize("H,KVH", [(2, 1), (12, 2), (3, 1)])
@pytest.mark.parametrize("ksplit,waves", [(1, 4), (2, 4), (1, 8)])
def test_linear_decode_rope_kv(L, H, KVH, ksplit, waves):
    """Fused RMSNorm + qkv projection + bias + M-RoPE + q / K-cache / V^T-cache writes."""
    rng = np.random.default_rng(70 + H + ksplit)
    hd, B, s_max, K, T = 128, 5, 256, 1024, 7
    N = (H + 2 * KVH) * hd
    plen = np.asarray([0, 3, 60, 64, 198], np.int32)
    step = np.asarray([0, 2, 3, 0, 6], np.int32)       # index of the decode position of each row
    ctxs = plen + step
    x, W = rnd(rng, B, K, scale=2.0), rnd(rng, N, K, scale=K ** -0.5)
    bias, nw = rnd(rng, N, scale=0.1), bf16_round(1 + 0.1 * rnd(rng, K))
    ang = rng.uniform(0, 6.28, size=(B, T, 64)).astype(np.float32)
    cs = np.concatenate([bf16_round(np.cos(ang)), bf16_round(np.sin(ang))], -1).astype(np.float32)  # [B,T,128]
    kc = rnd(rng, B, KVH, s_max, hd); vt = rnd(rng, B, KVH, s_max // 64, hd, 64)
    kc_d, vt_d = dev_bf16(kc), dev_bf16(vt)
    q_d = torch.zeros(B, H, hd, dtype=torch.bfloat16, device=DEV)
    xd, Wd, bd, nd = dev_bf16(x), dev_bf16(pack_w16x64(W)), dev_bf16(bias), dev_bf16(nw)
    cs_d, ctx_d, pl_d = torch.from_numpy(cs).to(DEV), torch.from_numpy(ctxs).to(DEV), torch.from_numpy(plen).to(DEV)
    ws = torch.zeros((N // 32) * ksplit * 2 * 256, dtype=torch.float32, device=DEV)
    cnt = torch.zeros(N // 32, dtype=torch.int32, device=DEV)
    dec_call(L, DEC_ROPE_KV, ptr(xd), K, ptr(Wd), B, N, K, bias=ptr(bd), norm_w=ptr(nd), waves=waves, ksplit=ksplit,
             ws=ptr(ws), cnt=ptr(cnt), cs=ptr(cs_d), cs_stride=T, plen=ptr(pl_d), ctx=ptr(ctx_d), q_out=ptr(q_d),
             kc=ptr(kc_d), vc=ptr(vt_d), heads=H, kv_heads=KVH, s_max=s_max)
    xn = bf16_round(O.rms_norm(x, nw, 1e-6, O._Policy("bf16")))
    qkv = bf16_round(ref_linear(xn, W, bias))
    q = qkv[:, :H * hd].reshape(B, H, hd); k = qkv[:, H * hd:(H + KVH) * hd].reshape(B, KVH, hd)
    v = qkv[:, (H + KVH) * hd:].reshape(B, KVH, hd)
    csb = cs[np.arange(B), step]
    cos, sin = np.concatenate([csb[:, :64]] * 2, -1), np.concatenate([csb[:, 64:]] * 2, -1)
    qr = q * cos[:, None] + O.rotate_half(q) * sin[:, None]
    kr = k * cos[:, None] + O.rotate_half(k) * sin[:, None]
    assert_close_bf16(host(q_d), qr, abs_=3e-2, what="fused q")
    got_k, got_vt = host(kc_d), host(vt_d)
    gv = POS.vt_rows(got_vt)
    ref_v_all = POS.vt_rows(vt).copy()
    ref_k_all = kc.copy()
    for b, c in enumerate(ctxs):
        assert_close_bf16(got_k[b, :, c], kr[b], abs_=3e-2, what="fused k append")
        assert_close_bf16(gv[b, :, c], v[b], abs_=3e-2, what="fused v append")
        ref_k_all[b, :, c] = got_k[b, :, c]
        ref_v_all[b, :, c] = gv[b, :, c]
    np.testing.assert_array_equal(got_k, ref_k_all)   # nothing else in the cache was touched
    np.testing.assert_array_equal(gv, ref_v_all)
    assert not cnt.cpu().numpy().any()


def test_linear_decode_argmax_and_sample(L):
    """lm_head partial argmax + kr_sample_greedy: ties -> lowest index; logits optional; bookkeeping."""
    rng = np.random.default_rng(80)
    B, V, d = 3, 16 * 2 * 40 + 16, 256     # odd tile count: last workgroup has one tile
    x, W = ints(rng, B, d), ints(rng, V, d)
    W[:, 200:] = 0
    nw = np.ones(d, np.float32)
    logits_ref = ref_linear(bf16_round(O.rms_norm(x, nw, 1e-6, O._Policy("bf16"))), W)
    xd, Wd, nd = dev_bf16(x), dev_bf16(pack_w16x64(W)), dev_bf16(nw)
    n_part = (V // 16 + 1) // 2
    av = torch.zeros(B, n_part, dtype=torch.float32, device=DEV); ai = torch.zeros(B, n_part, dtype=torch.int32, device=DEV)
    lg = torch.zeros(B, V, dtype=torch.float32, device=DEV)
    dec_call(L, DEC_ARGMAX, ptr(xd), d, ptr(Wd), B, V, d, out_f32=ptr(lg), ldc=V, norm_w=ptr(nd), av=ptr(av), ai=ptr(ai),
             max_blocks=6)
    got = host(lg)
    np.testing.assert_allclose(got, logits_ref, atol=2e-2, rtol=1e-2)
    table = rnd(rng, V, d); td = dev_bf16(table)
    tok = torch.zeros(B, dtype=torch.int32, device=DEV); hist = torch.full((6, 4), -1, dtype=torch.int32, device=DEV)
    plen = torch.tensor([10, 19, 27], dtype=torch.int32, device=DEV)
    ctx = torch.tensor([10, 20, 30], dtype=torch.int32, device=DEV)      # -> generated-token index 1, 2, 4
    fin = torch.zeros(B, dtype=torch.int32, device=DEV); eos = torch.tensor([-5], dtype=torch.int32, device=DEV)
    xn = torch.zeros(B, d, dtype=torch.bfloat16, device=DEV)
    L.kr_sample_greedy(ptr(av), ptr(ai), n_part, ptr(td), d, ptr(tok), ptr(hist), 4, ptr(plen), ptr(ctx), ptr(fin), ptr(eos),
                       1, 0, 0, ptr(xn), B, 0)
    torch.cuda.synchronize()
    want = got.argmax(1)           # numpy argmax = lowest index among ties (integer logits tie often)
    np.testing.assert_array_equal(tok.cpu().numpy(), want)
    h = hist.cpu().numpy()
    assert h[1, 0] == want[0] and h[2, 1] == want[1] and h[4, 2] == want[2] and (h == -1).sum() == 24 - 3
    assert ctx.cpu().tolist() == [11, 21, 31]
    np.testing.assert_array_equal(host(xn), table[want])


def run_wide(L, mode, x, W, blocks, waves, bias=None, res=None, norm_w=None, f32=False):
    M, K = x.shape
    N = W.shape[0]
    nc = N // 2 if mode == DEC_SILU8 else N
    xd, Wd = dev_bf16(x), dev_bf16(pack_w16x64(W))
    out = torch.full((M, nc), 9.0, dtype=torch.float32 if f32 else torch.bfloat16, device=DEV)
    bd = dev_bf16(bias) if bias is not None else None
    rd = dev_bf16(res) if res is not None else None
    nd = dev_bf16(norm_w) if norm_w is not None else None
    L.kr_linear_decode_wide(mode, ptr(xd), K, ptr(Wd), ptr(bd), ptr(nd), 1e-6, ptr(rd), nc if res is not None else 0,
                            0 if f32 else ptr(out), ptr(out) if f32 else 0, nc, M, N, K, blocks, waves, 0, 0, 0)
    return host(out)


@pytest.mark.parametrize("M", [1, 8, 16])
@pytest.mark.parametrize("N,K,blocks,waves", [(16, 512, 1, 1), (16 * 37, 512, 3, 5), (16 * 37, 1536, 256, 4),
                                              (16 * 300, 1024, 7, 8), (16 * 41, 3584, 5, 2), (16 * 45, 2048, 4, 3)])
def test_linear_wide_plain_exact_on_integers(L, M, N, K, blocks, waves):
    """One wave per tile, register ring running across tile boundaries: exact on small integers; the zeroed K
    tail makes a wrong chunk order / wrong next-tile prefetch visible."""
    rng = np.random.default_rng(M + N + K + blocks)
    x, W = ints(rng, M, K), ints(rng, N, K)
    W[:, K // 2 + 40:] = 0
    W[::3, 100:300] = 1
    np.testing.assert_array_equal(run_wide(L, DEC_PLAIN, x, W, blocks, waves), ref_linear(x, W))
    np.testing.assert_array_equal(run_wide(L, DEC_PLAIN, x, W, blocks, waves, f32=True), ref_linear(x, W))


@pytest.mark.parametrize("M,K,blocks,waves", [(8, 1536, 256, 5), (16, 3584, 9, 8), (3, 512, 2, 3), (8, 2048, 11, 6)])
def test_linear_wide_silu8_norm_bias_residual(L, M, K, blocks, waves):
    rng = np.random.default_rng(162 + M)
    ff = 1016
    x, Wp, nw = rnd(rng, M, K, scale=2.0), rnd(rng, 2 * ff, K, scale=K ** -0.5), bf16_round(1 + 0.1 * rnd(rng, K))
    xn = bf16_round(O.rms_norm(x, nw, 1e-6, O._Policy("bf16")))
    got = run_wide(L, DEC_SILU8, x, Wp, blocks, waves, norm_w=nw)
    assert got.shape == (M, ff)
    assert_close_bf16(got, ref_linear(xn, Wp, epi=EPI_SILU_MUL8), what="wide silu8")
    N = 16 * 23
    W2, bias, res = rnd(rng, N, K, scale=K ** -0.5), rnd(rng, N, scale=0.1), rnd(rng, M, N)
    assert_close_bf16(run_wide(L, DEC_PLAIN, x, W2, blocks, waves, bias=bias, res=res), ref_linear(x, W2, bias, res),
                      what="wide plain")


@pytest.mark.parametrize("M", [17, 24, 32])
@pytest.mark.parametrize("N,K,blocks,waves", [(16 * 37, 512, 3, 5), (16 * 37, 1536, 256, 4), (16 * 45, 2048, 4, 3), (16 * 300, 1024, 7, 8)])
def test_linear_wide_two_column_tiles(L, M, N, K, blocks, waves):
    """M > 16: two 16-row column tiles of the batch per weight fragment (exact on integers), and the fused norm +
    SiLU*mul and per-wave argmax epilogues on both tiles."""
    rng = np.random.default_rng(M + N + K + blocks)
    x, W = ints(rng, M, K), ints(rng, N, K)
    W[:, K // 2 + 40:] = 0
    W[::3, 100:300] = 1
    np.testing.assert_array_equal(run_wide(L, DEC_PLAIN, x, W, blocks, waves), ref_linear(x, W))
    ff = N // 2 - (N // 2) % 8
    x2, Wp, nw = rnd(rng, M, K, scale=2.0), rnd(rng, 2 * ff, K, scale=K ** -0.5), bf16_round(1 + 0.1 * rnd(rng, K))
    xn = bf16_round(O.rms_norm(x2, nw, 1e-6, O._Policy("bf16")))
    assert_close_bf16(run_wide(L, DEC_SILU8, x2, Wp, blocks, waves, norm_w=nw), ref_linear(xn, Wp, epi=EPI_SILU_MUL8),
                      what="wide silu8, 2 column tiles")
    # argmax partials of every row
    xd, Wd = dev_bf16(x), dev_bf16(pack_w16x64(W))
    n_part = blocks * waves
    av = torch.zeros(M, n_part, device=DEV); ai = torch.zeros(M, n_part, dtype=torch.int32, device=DEV)
    L.kr_linear_decode_wide(DEC_ARGMAX, ptr(xd), K, ptr(Wd), 0, 0, 1e-6, 0, 0, 0, 0, 0, M, N, K, blocks, waves, ptr(av), ptr(ai), 0)
    torch.cuda.synchronize()
    ref = ref_linear(x, W)
    a_, i_ = av.cpu().numpy(), ai.cpu().numpy()
    for b in range(M):
        assert i_[b, int(np.lexsort((i_[b], -a_[b]))[0])] == int(ref[b].argmax()) and a_[b].max() == ref[b].max()


def test_linear_wide_at_the_bench_shapes(L):
    """The two wide launches of the headline workload at their real sizes (Qwen2-VL-2B, batch 8): gate/up
    (N = 17920, 224 workgroups x 5 waves, one tile per wave) and the lm_head (N = 151936, 256 x 8, 4-5 tiles per wave):
    exact on integers over every output / the argmax of every row."""
    rng = np.random.default_rng(2025)
    M, K = 8, 1536
    x = ints(rng, M, K)
    W = ints(rng, 17920, K)
    W[:, K // 2 + 40:] = 0
    W[::3, 100:300] = 1
    np.testing.assert_array_equal(run_wide(L, DEC_PLAIN, x, W, 224, 5), ref_linear(x, W))
    V = 151936
    Wv = ints(rng, V, K)
    Wv[:, 300:] = 0
    ref = ref_linear(x, Wv)
    xd, Wd = dev_bf16(x), dev_bf16(pack_w16x64(Wv))
    blocks, waves = 256, 8
    n_part = blocks * waves
    av = torch.zeros(M, n_part, device=DEV); ai = torch.zeros(M, n_part, dtype=torch.int32, device=DEV)
    lg = torch.zeros(M, V, dtype=torch.float32, device=DEV)
    L.kr_linear_decode_wide(DEC_ARGMAX, ptr(xd), K, ptr(Wd), 0, 0, 1e-6, 0, 0, 0, ptr(lg), V, M, V, K, blocks, waves, ptr(av), ptr(ai), 0)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(lg.cpu().numpy(), ref)
    a_, i_ = av.cpu().numpy(), ai.cpu().numpy()
    for b in range(M):
        best = int(np.lexsort((i_[b], -a_[b]))[0])                   # highest value, ties to the lowest token id
        assert i_[b, best] == int(np.flatnonzero(ref[b] == ref[b].max())[0]) and a_[b, best] == ref[b].max()


@pytest.mark.parametrize("M", [17, 21, 32])
@pytest.mark.parametrize("N,blocks", [(16 * 37, 3), (16 * 300, 11), (16 * 90, 256), (16 * 1200, 32)])
def test_linear_wide_k_halves_at_the_7b_width(L, M, N, blocks):
    """17..32 rows at K = 3584 (32 x rows do not fit the LDS): dec_wide_kh_kernel stages K in two halves.  Fused RMSNorm +
    SiLU*mul against the numpy reference, and — exact on integers — the logits and the per-wave argmax partials of every
    row (1..5 tiles per wave, waves without tiles, the ring running from one tile's half into the next)."""
    rng = np.random.default_rng(M + N + blocks)
    K, waves = 3584, 8
    ff = N // 2 - (N // 2) % 8
    x2, Wp, nw = rnd(rng, M, K, scale=2.0), rnd(rng, 2 * ff, K, scale=K ** -0.5), bf16_round(1 + 0.1 * rnd(rng, K))
    xn = bf16_round(O.rms_norm(x2, nw, 1e-6, O._Policy("bf16")))
    assert_close_bf16(run_wide(L, DEC_SILU8, x2, Wp, blocks, waves, norm_w=nw), ref_linear(xn, Wp, epi=EPI_SILU_MUL8),
                      what="wide silu8, K halves")
    x, W = ints(rng, M, K), ints(rng, N, K)
    W[:, K // 2 + 40:] = 0                      # a wrong half / chunk order shows
    W[::3, 100:300] = 1
    W[1::5, 1800:1900] = -1
    xd, Wd = dev_bf16(x), dev_bf16(pack_w16x64(W))
    n_part = blocks * waves
    av = torch.zeros(M, n_part, device=DEV); ai = torch.zeros(M, n_part, dtype=torch.int32, device=DEV)
    lg = torch.zeros(M, N, dtype=torch.float32, device=DEV)
    L.kr_linear_decode_wide(DEC_ARGMAX, ptr(xd), K, ptr(Wd), 0, 0, 1e-6, 0, 0, 0, ptr(lg), N, M, N, K, blocks, waves, ptr(av), ptr(ai), 0)
    torch.cuda.synchronize()
    ref = ref_linear(x, W)
    np.testing.assert_array_equal(lg.cpu().numpy(), ref)
    a_, i_ = av.cpu().numpy(), ai.cpu().numpy()
    for b in range(M):
        best = int(np.lexsort((i_[b], -a_[b]))[0])
        assert i_[b, best] == int(np.flatnonzero(ref[b] == ref[b].max())[0]) and a_[b, best] == ref[b].max()
    if N // 16 < n_part:                        # waves without a tile leave (-inf, INT_MAX)
        assert np.isneginf(a_[:, N // 16:]).all() and (i_[:, N // 16:] == 0x7fffffff).all()


def test_linear_wide_rejects_33_rows_and_unsupported_17_row_shapes(L):
    x = torch.zeros(33, 3584, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(KarantaHipError):
        L.kr_linear_decode_wide(DEC_PLAIN, ptr(x), 512, ptr(x), 0, 0, 1e-6, 0, 0, ptr(x), 0, 16, 33, 16, 512, 1, 4, 0, 0, 0)
    with pytest.raises(KarantaHipError):     # 17 rows at K = 3584: SILU8 / ARGMAX only (K halves), 8 waves, <= 5 tiles per wave
        L.kr_linear_decode_wide(DEC_PLAIN, ptr(x), 3584, ptr(x), 0, 0, 1e-6, 0, 0, ptr(x), 0, 16, 17, 16, 3584, 1, 8, 0, 0, 0)
    with pytest.raises(KarantaHipError):
        L.kr_linear_decode_wide(DEC_SILU8, ptr(x), 3584, ptr(x), 0, 0, 1e-6, 0, 0, ptr(x), 0, 16, 17, 32, 3584, 1, 4, 0, 0, 0)
    with pytest.raises(KarantaHipError):
        L.kr_linear_decode_wide(DEC_SILU8, ptr(x), 3584, ptr(x), 0, 0, 1e-6, 0, 0, ptr(x), 0, 16 * 50, 17, 32 * 50, 3584, 1, 8, 0, 0, 0)
    with pytest.raises(KarantaHipError):     # 32 rows at K = 3072: they do not fit the LDS and there is no K-halves build
        L.kr_linear_decode_wide(DEC_SILU8, ptr(x), 3072, ptr(x), 0, 0, 1e-6, 0, 0, ptr(x), 0, 16, 32, 32, 3072, 1, 8, 0, 0, 0)


def test_linear_wide_argmax_and_sample(L):
    rng = np.random.default_rng(81)
    B, V, d = 3, 16 * 83, 512
    x, W = ints(rng, B, d), ints(rng, V, d)
    W[:, 200:] = 0
    nw = np.ones(d, np.float32)
    logits_ref = ref_linear(bf16_round(O.rms_norm(x, nw, 1e-6, O._Policy("bf16"))), W)
    xd, Wd, nd = dev_bf16(x), dev_bf16(pack_w16x64(W)), dev_bf16(nw)
    blocks, waves = 6, 4
    n_part = blocks * waves
    av = torch.zeros(B, n_part, dtype=torch.float32, device=DEV); ai = torch.zeros(B, n_part, dtype=torch.int32, device=DEV)
    lg = torch.zeros(B, V, dtype=torch.float32, device=DEV)
    L.kr_linear_decode_wide(DEC_ARGMAX, ptr(xd), d, ptr(Wd), 0, ptr(nd), 1e-6, 0, 0, 0, ptr(lg), V, B, V, d, blocks, waves,
                            ptr(av), ptr(ai), 0)
    got = host(lg)
    np.testing.assert_allclose(got, logits_ref, atol=2e-2, rtol=1e-2)
    avh, aih = av.cpu().numpy(), ai.cpu().numpy()
    for b in range(B):
        for p_ in range(n_part):       # partial p = the tiles p, p + n_part, ...; ties -> lowest index
            cols = np.concatenate([np.arange(t * 16, t * 16 + 16) for t in range(p_, V // 16, n_part)])
            j = int(np.argmax(got[b, cols]))
            assert avh[b, p_] == got[b, cols[j]] and aih[b, p_] == cols[j]
        assert aih[b, int(np.lexsort((aih[b], -avh[b]))[0])] == int(got[b].argmax())
    # without the logits buffer; more waves than tiles: empty partials are (-inf, INT_MAX)
    blocks2, waves2 = 256, 8
    av2 = torch.zeros(B, blocks2 * waves2, device=DEV); ai2 = torch.zeros(B, blocks2 * waves2, dtype=torch.int32, device=DEV)
    L.kr_linear_decode_wide(DEC_ARGMAX, ptr(xd), d, ptr(Wd), 0, ptr(nd), 1e-6, 0, 0, 0, 0, 0, B, V, d, blocks2, waves2,
                            ptr(av2), ptr(ai2), 0)
    torch.cuda.synchronize()
    a2, i2 = av2.cpu().numpy(), ai2.cpu().numpy()
    assert np.isneginf(a2[:, V // 16:]).all() and (i2[:, V // 16:] == 0x7fffffff).all()
    for b in range(B):
        assert i2[b, int(np.lexsort((i2[b], -a2[b]))[0])] == int(got[b].argmax())
    tok = torch.zeros(B, dtype=torch.int32, device=DEV); hist = torch.full((4, 4), -1, dtype=torch.int32, device=DEV)
    plen = torch.zeros(B, dtype=torch.int32, device=DEV); ctx = torch.zeros(B, dtype=torch.int32, device=DEV)
    fin = torch.zeros(B, dtype=torch.int32, device=DEV); eos = torch.tensor([-5], dtype=torch.int32, device=DEV)
    table = rnd(rng, V, d); td = dev_bf16(table); xn = torch.zeros(B, d, dtype=torch.bfloat16, device=DEV)
    L.kr_sample_greedy(ptr(av2), ptr(ai2), blocks2 * waves2, ptr(td), d, ptr(tok), ptr(hist), 4, ptr(plen), ptr(ctx), ptr(fin),
                       ptr(eos), 1, 0, 0, ptr(xn), B, 0)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(tok.cpu().numpy(), got.argmax(1))


def test_linear_wide_rejects_bad_shapes(L):
    x = torch.zeros(8, 576, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(KarantaHipError):
        L.kr_linear_decode_wide(DEC_PLAIN, ptr(x), 576, ptr(x), 0, 0, 1e-6, 0, 0, ptr(x), 0, 16, 8, 16, 576, 1, 4, 0, 0, 0)
    with pytest.raises(KarantaHipError):
        L.kr_linear_decode_wide(DEC_PLAIN, ptr(x), 512, ptr(x), 0, 0, 1e-6, 0, 0, ptr(x), 0, 16, 8, 16, 512, 1, 9, 0, 0, 0)


def narrow_call(L, mode, x, W, M, N, K, out=0, out_f32=0, ldc=0, bias=0, norm_w=0, res=0, ldr=0, waves=8, ksplit=1,
                part_in=0, x_out=0, cs=0, cs_stride=0, plen=0, ctx=0, q_out=0, kc=0, vc=0, heads=0, kv_heads=0, s_max=64,
                n_part=2, opts=None):
    L.kr_linear_decode_narrow(mode, x, K, part_in, n_part if part_in else 0, x_out, K, W, bias, norm_w, 1e-6, res, ldr, out,
                              out_f32, ldc, M, N, K, waves, ksplit, cs, cs_stride, plen, ctx, q_out, kc, vc, heads,
                              kv_heads, s_max, opts, 0)


@pytest.mark.parametrize("M", [1, 8, 16, 23, 32])
@pytest.mark.parametrize("N,K,waves", [(16, 64, 8), (48, 256, 8), (1536, 1536, 8), (96, 8960, 16), (96, 8960, 8),
                                       (16 * 5, 64 * 19, 16), (32, 3584, 8)])
def test_linear_narrow_plain_exact_on_integers(L, M, N, K, waves):
    if M > 16 and waves == 16:
        with pytest.raises(KarantaHipError):       # two column tiles: 8-wave workgroups only
            narrow_call(L, DEC_PLAIN, ptr(torch.zeros(M, K, dtype=torch.bfloat16, device=DEV)),
                        ptr(torch.zeros(N, K, dtype=torch.bfloat16, device=DEV)), M, N, K,
                        out=ptr(torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)), ldc=N, waves=waves)
        return
    """K split over the waves of one workgroup (uneven chunk counts, waves without any chunk), x fragments straight
    from global memory; the zeroed K tail makes a wrong chunk order visible."""
    rng = np.random.default_rng(M + N + K + waves)
    x, W = ints(rng, M, K), ints(rng, N, K)
    W[:, K // 2 + 8:] = 0
    W[::3, 10:40] = 1
    xd, Wd = dev_bf16(x), dev_bf16(pack_w16x64(W))
    out = torch.full((M, N), 9.0, dtype=torch.bfloat16, device=DEV)
    narrow_call(L, DEC_PLAIN, ptr(xd), ptr(Wd), M, N, K, out=ptr(out), ldc=N, waves=waves)
    np.testing.assert_array_equal(host(out), ref_linear(x, W))
    res, bias = rnd(rng, M, N), rnd(rng, N, scale=0.1)
    rd, bd = dev_bf16(res), dev_bf16(bias)
    narrow_call(L, DEC_PLAIN, ptr(xd), ptr(Wd), M, N, K, out=ptr(rd), ldc=N, res=ptr(rd), ldr=N, bias=ptr(bd), waves=waves)
    assert_close_bf16(host(rd), ref_linear(x, W, bias, res), what="narrow plain in-place residual")


@pytest.mark.parametrize("M,N,K,ksplit,waves", [(8, 1536, 8960, 2, 16), (16, 96, 8960, 3, 8), (3, 48, 256, 2, 8),
                                                (8, 64, 64 * 9, 8, 8), (32, 1536, 8960, 2, 8), (19, 96, 8960, 3, 8),
                                                # >= 192 tiles: two tiles per workgroup on one x-fragment ring (7B down_proj)
                                                (8, 3584, 18944, 2, 8), (13, 3584, 18944, 2, 16), (16, 3072, 1216, 3, 8),
                                                (27, 3584, 18944, 2, 8), (32, 3072, 640, 2, 8)])   # ... and two batch column tiles
def test_linear_narrow_deferred_splitk_slabs(L, M, N, K, ksplit, waves):
    """ksplit > 1 writes f32 slabs [ksplit][M][N]; their sum is the product (exact on integers)."""
    rng = np.random.default_rng(M + N + ksplit)
    x, W = ints(rng, M, K), ints(rng, N, K)
    xd, Wd = dev_bf16(x), dev_bf16(pack_w16x64(W))
    slabs = torch.full((ksplit, M, N), 7.0, dtype=torch.float32, device=DEV)
    narrow_call(L, DEC_PLAIN, ptr(xd), ptr(Wd), M, N, K, out_f32=ptr(slabs), ldc=N, waves=waves, ksplit=ksplit)
    got = slabs.cpu().numpy()
    np.testing.assert_array_equal(got.sum(0), ref_linear(x, W))
    cpb = -(-(K // 64) // ksplit) * 64
    for k in range(ksplit):      # slab k holds exactly its K range
        np.testing.assert_array_equal(got[k], ref_linear(x[:, k * cpb:(k + 1) * cpb], W[:, k * cpb:(k + 1) * cpb])
                                      if k * cpb < K else np.zeros((M, N), np.float32))
    with pytest.raises(KarantaHipError):   # slabs only: no residual / bf16 output in the split form
        narrow_call(L, DEC_PLAIN, ptr(xd), ptr(Wd), M, N, K, out=ptr(xd), ldc=N, waves=waves, ksplit=ksplit)


@pytest.mark.parametrize("M,N,K,waves", [(8, 1536, 8960, 16), (32, 1536, 8960, 8), (5, 3584, 18944, 16), (16, 96, 256, 8),
                                         (32, 3584, 18944, 8), (8, 3584, 18944, 8)])
def test_linear_narrow_one_slab_atomic_split(L, M, N, K, waves):
    """kr_narrow_opts.atomic_out: the two K ranges of a ksplit-2 launch ADD into ONE zeroed f32 slab (float atomics; two
    addends onto zero: the same bits in either order = slab0 + slab1 of the two-slab form), and a launch can carry the
    zeroing of another range (zero_ptr / zero_bytes).  Exact on integers, identical run to run, equal to the sum of the
    two-slab launch on real data.  The options are arguments of ONE launch: the next launch without them writes two slabs."""
    rng = np.random.default_rng(M + N + K)
    x, W = ints(rng, M, K), ints(rng, N, K)
    xd, Wd = dev_bf16(x), dev_bf16(pack_w16x64(W))
    acc = torch.zeros(M, N, dtype=torch.float32, device=DEV)
    other = torch.full((M * N + 8,), 3.0, dtype=torch.float32, device=DEV)     # the range this launch is asked to zero
    narrow_call(L, DEC_PLAIN, ptr(xd), ptr(Wd), M, N, K, out_f32=ptr(acc), ldc=N, waves=waves, ksplit=2,
                opts=narrow_opts(ptr(other), M * N * 4, True))
    np.testing.assert_array_equal(acc.cpu().numpy(), ref_linear(x, W))
    oc = other.cpu().numpy()
    assert not oc[:M * N].any() and (oc[M * N:] == 3.0).all(), "zeroing job: exactly the requested range"
    # no hidden state: the next launch (no options) writes two slabs again and zeroes nothing
    slabs = torch.full((2, M, N), 7.0, dtype=torch.float32, device=DEV)
    narrow_call(L, DEC_PLAIN, ptr(xd), ptr(Wd), M, N, K, out_f32=ptr(slabs), ldc=N, waves=waves, ksplit=2)
    np.testing.assert_array_equal(slabs.cpu().numpy().sum(0), ref_linear(x, W))
    # real data: a + b does not depend on which range lands first
    xr, Wr = rnd(rng, M, K), rnd(rng, N, K, scale=K ** -0.5)
    xd, Wd = dev_bf16(xr), dev_bf16(pack_w16x64(Wr))
    narrow_call(L, DEC_PLAIN, ptr(xd), ptr(Wd), M, N, K, out_f32=ptr(slabs), ldc=N, waves=waves, ksplit=2)
    two = slabs.cpu().numpy()
    for _ in range(3):
        acc.zero_()
        narrow_call(L, DEC_PLAIN, ptr(xd), ptr(Wd), M, N, K, out_f32=ptr(acc), ldc=N, waves=waves, ksplit=2,
                    opts=narrow_opts(atomic_out=True))
        np.testing.assert_array_equal(acc.cpu().numpy(), two[0] + two[1])
    with pytest.raises(KarantaHipError):       # three addends would make the sum order-dependent: refused
        narrow_call(L, DEC_PLAIN, ptr(xd), ptr(Wd), M, N, K, out_f32=ptr(acc), ldc=N, waves=8, ksplit=3, opts=narrow_opts(atomic_out=True))
    with pytest.raises(KarantaHipError):       # a failed launch leaves nothing armed: the next plain launch is plain
        narrow_call(L, DEC_PLAIN, ptr(xd), ptr(Wd), M, N, K, out_f32=ptr(acc), ldc=N, waves=5, ksplit=2,
                    opts=narrow_opts(ptr(other), M * N * 4, True))
    other.fill_(3.0)
    narrow_call(L, DEC_PLAIN, ptr(xd), ptr(Wd), M, N, K, out_f32=ptr(slabs), ldc=N, waves=waves, ksplit=2)
    assert (other.cpu().numpy() == 3.0).all()
    np.testing.assert_array_equal(slabs.cpu().numpy(), two)


@pytest.mark.parametrize("M,K", [(8, 1536), (16, 3584), (32, 1536), (27, 2048)])
def test_linear_narrow_norm_with_one_partial_slab(L, M, K):
    """n_part_in = 1: x_new = bf16(x + slab) -> x_out, RMSNorm(x_new) @ W^T + bias (the consumer of the one-slab split)."""
    rng = np.random.default_rng(900 + M + K)
    N = 16 * 9
    x, W, nw = rnd(rng, M, K, scale=2.0), rnd(rng, N, K, scale=K ** -0.5), bf16_round(1 + 0.1 * rnd(rng, K))
    bias = rnd(rng, N, scale=0.1)
    p = (rng.standard_normal((1, M, K)) * 0.5).astype(np.float32)
    xd, Wd, nd, bd = dev_bf16(x), dev_bf16(pack_w16x64(W)), dev_bf16(nw), dev_bf16(bias)
    pd = torch.from_numpy(p).to(DEV)
    xo = torch.full((M, K), 5.0, dtype=torch.bfloat16, device=DEV)
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    narrow_call(L, DEC_PLAIN, ptr(xd), ptr(Wd), M, N, K, out=ptr(out), ldc=N, norm_w=ptr(nd), bias=ptr(bd), part_in=ptr(pd),
                x_out=ptr(xo), n_part=1)
    x_new = bf16_round(x + p[0])
    np.testing.assert_array_equal(host(xo), x_new)          # one f32 add, one rounding: exact
    xn = bf16_round(O.rms_norm(x_new, nw, 1e-6, O._Policy("bf16")))
    assert_close_bf16(host(out), ref_linear(xn, W, bias), what="narrow norm + one partial slab")


@pytest.mark.parametrize("M,K,parts", [(8, 1536, True), (8, 1536, False), (16, 3584, True), (3, 256, False), (11, 1536, True),
                                       (8, 2048, True), (5, 2048, False), (32, 1536, True), (27, 2048, True), (20, 512, False)])
def test_linear_narrow_norm_with_deferred_partials(L, M, K, parts):
    """x_new = bf16(x + slab0 + slab1) -> x_out (written once, by workgroup 0), RMSNorm(x_new) @ W^T + bias."""
    rng = np.random.default_rng(300 + M + K)
    N = 16 * 9
    x, W, nw = rnd(rng, M, K, scale=2.0), rnd(rng, N, K, scale=K ** -0.5), bf16_round(1 + 0.1 * rnd(rng, K))
    bias = rnd(rng, N, scale=0.1)
    p = (rng.standard_normal((2, M, K)) * 0.5).astype(np.float32)
    xd, Wd, nd, bd = dev_bf16(x), dev_bf16(pack_w16x64(W)), dev_bf16(nw), dev_bf16(bias)
    pd = torch.from_numpy(p).to(DEV)
    xo = torch.full((M, K), 5.0, dtype=torch.bfloat16, device=DEV)
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    narrow_call(L, DEC_PLAIN, ptr(xd), ptr(Wd), M, N, K, out=ptr(out), ldc=N, norm_w=ptr(nd), bias=ptr(bd),
                part_in=ptr(pd) if parts else 0, x_out=ptr(xo) if parts else 0)
    x_new = bf16_round(x + p[0] + p[1]) if parts else x
    if parts:
        got_x = host(xo)
        assert np.abs(got_x - x_new).max() <= np.abs(x_new).max() * 2 ** -7     # fp32 sum order may move one bf16 ulp
        x_new = got_x
    xn = bf16_round(O.rms_norm(x_new, nw, 1e-6, O._Policy("bf16")))
    assert_close_bf16(host(out), ref_linear(xn, W, bias), what="narrow norm + partials")


def test_linear_narrow_partials_reject_unsupported_k(L):
    x = torch.zeros(8, 256, dtype=torch.bfloat16, device=DEV); p = torch.zeros(2, 8, 256, device=DEV)
    with pytest.raises(KarantaHipError):
        narrow_call(L, DEC_PLAIN, ptr(x), ptr(x), 8, 16, 256, out=ptr(x), ldc=16, norm_w=ptr(x), part_in=ptr(p), x_out=ptr(p))
    with pytest.raises(KarantaHipError):   # x_out must not alias x
        L.kr_linear_decode_narrow(DEC_PLAIN, ptr(x), 256, ptr(p), 2, ptr(x), 256, ptr(x), 0, ptr(x), 1e-6, 0, 0, ptr(x), 0, 16,
                                  8, 16, 256, 8, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 64, None, 0)


@pytest.mark.parametrize("H,KVH,K,parts", [(2, 1, 1024, False), (12, 2, 1536, True), (3, 1, 1536, False), (28, 4, 3584, True),
                                           (16, 2, 2048, True)])
@pytest.mark.parametrize("B", [5, 21])
def test_linear_narrow_rope_kv(L, H, KVH, K, parts, B):
    """Fused (partial sums +) RMSNorm + qkv projection + bias + M-RoPE + q / K-cache / V^T-cache writes; B = 21: two batch
    column tiles."""
    if B > 16 and K == 3584:
        pytest.skip("two column tiles need K <= 2048")
    rng = np.random.default_rng(170 + H)
    hd, s_max, T = 128, 256, 7
    N = (H + 2 * KVH) * hd
    plen = np.asarray([0, 3, 60, 64, 198] + list(rng.integers(0, 190, size=B - 5)), np.int32)
    step = np.asarray([0, 2, 3, 0, 6] + list(rng.integers(0, T, size=B - 5)), np.int32)
    ctxs = plen + step
    x, W = rnd(rng, B, K, scale=2.0), rnd(rng, N, K, scale=K ** -0.5)
    bias, nw = rnd(rng, N, scale=0.1), bf16_round(1 + 0.1 * rnd(rng, K))
    p = (rng.standard_normal((2, B, K)) * 0.5).astype(np.float32)
    ang = rng.uniform(0, 6.28, size=(B, T, 64)).astype(np.float32)
    cs = np.concatenate([bf16_round(np.cos(ang)), bf16_round(np.sin(ang))], -1).astype(np.float32)
    kc = rnd(rng, B, KVH, s_max, hd); vt = rnd(rng, B, KVH, s_max // 64, hd, 64)
    kc_d, vt_d = dev_bf16(kc), dev_bf16(vt)
    q_d = torch.zeros(B, H, hd, dtype=torch.bfloat16, device=DEV)
    xd, Wd, bd, nd = dev_bf16(x), dev_bf16(pack_w16x64(W)), dev_bf16(bias), dev_bf16(nw)
    pd = torch.from_numpy(p).to(DEV); xo = torch.zeros(B, K, dtype=torch.bfloat16, device=DEV)
    cs_d, ctx_d, pl_d = torch.from_numpy(cs).to(DEV), torch.from_numpy(ctxs).to(DEV), torch.from_numpy(plen).to(DEV)
    narrow_call(L, DEC_ROPE_KV, ptr(xd), ptr(Wd), B, N, K, bias=ptr(bd), norm_w=ptr(nd), part_in=ptr(pd) if parts else 0,
                x_out=ptr(xo) if parts else 0, cs=ptr(cs_d), cs_stride=T, plen=ptr(pl_d), ctx=ptr(ctx_d), q_out=ptr(q_d),
                kc=ptr(kc_d), vc=ptr(vt_d), heads=H, kv_heads=KVH, s_max=s_max)
    x_new = host(xo) if parts else x
    if parts:
        ref_x = bf16_round(x + p[0] + p[1])
        assert np.abs(x_new - ref_x).max() <= np.abs(ref_x).max() * 2 ** -7
    xn = bf16_round(O.rms_norm(x_new, nw, 1e-6, O._Policy("bf16")))
    qkv = bf16_round(ref_linear(xn, W, bias))
    q = qkv[:, :H * hd].reshape(B, H, hd); k = qkv[:, H * hd:(H + KVH) * hd].reshape(B, KVH, hd)
    v = qkv[:, (H + KVH) * hd:].reshape(B, KVH, hd)
    csb = cs[np.arange(B), step]
    cos, sin = np.concatenate([csb[:, :64]] * 2, -1), np.concatenate([csb[:, 64:]] * 2, -1)
    qr = q * cos[:, None] + O.rotate_half(q) * sin[:, None]
    kr = k * cos[:, None] + O.rotate_half(k) * sin[:, None]
    assert_close_bf16(host(q_d), qr, abs_=3e-2, what="narrow fused q")
    got_k, got_vt = host(kc_d), host(vt_d)
    gv = POS.vt_rows(got_vt)
    ref_v_all = POS.vt_rows(vt).copy()
    ref_k_all = kc.copy()
    for b, c in enumerate(ctxs):
        assert_close_bf16(got_k[b, :, c], kr[b], abs_=3e-2, what="narrow fused k append")
        assert_close_bf16(gv[b, :, c], v[b], abs_=3e-2, what="narrow fused v append")
        ref_k_all[b, :, c] = got_k[b, :, c]
        ref_v_all[b, :, c] = gv[b, :, c]
    np.testing.assert_array_equal(got_k, ref_k_all)
    np.testing.assert_array_equal(gv, ref_v_all)


@pytest.mark.parametrize("H,KVH,K,n_part", [(12, 2, 1536, 1), (28, 4, 3584, 1), (12, 2, 1536, 2), (28, 4, 3584, 0), (16, 2, 2048, 1),
                                            (2, 1, 256, 0)])
@pytest.mark.parametrize("B", [8, 21, 32])
def test_resnorm_then_direct_qkv_gives_the_fused_launch_bits(L, H, KVH, K, n_part, B):
    """Decode batches above 16 rows: kr_decode_resnorm (residual sum + RMSNorm once for the batch) followed by ONE
    kr_linear_decode_narrow(ROPE_KV, norm_w = NULL) launch whose x fragments come straight from L2 — against the fused
    launch (norm prologue, per 16-row range: what batches up to 16 rows run), BIT FOR BIT: x_new, q, the K-cache rows and
    the V^T-cache columns.  That equality is what keeps a page's tokens independent of the batch size it decodes in.
    At the 7B width (K = 3584) 32 rows do not fit the fused kernel's LDS: the fused reference runs per 16-row range."""
    rng = np.random.default_rng(270 + H + B + n_part)
    hd, s_max, T = 128, 256, 7
    N = (H + 2 * KVH) * hd
    plen = np.asarray([0, 3, 60, 64, 198] + list(rng.integers(0, 190, size=B - 5)), np.int32)
    step = np.asarray([0, 2, 3, 0, 6] + list(rng.integers(0, T, size=B - 5)), np.int32)
    ctxs = plen + step
    x, W = rnd(rng, B, K, scale=2.0), rnd(rng, N, K, scale=K ** -0.5)
    bias, nw = rnd(rng, N, scale=0.1), bf16_round(1 + 0.1 * rnd(rng, K))
    p = (rng.standard_normal((max(n_part, 1), B, K)) * 0.5).astype(np.float32)
    ang = rng.uniform(0, 6.28, size=(B, T, 64)).astype(np.float32)
    cs = np.concatenate([bf16_round(np.cos(ang)), bf16_round(np.sin(ang))], -1).astype(np.float32)
    kc = rnd(rng, B, KVH, s_max, hd); vt = rnd(rng, B, KVH, s_max // 64, hd, 64)
    xd, Wd, bd, nd = dev_bf16(x), dev_bf16(pack_w16x64(W)), dev_bf16(bias), dev_bf16(nw)
    pd = torch.from_numpy(p).to(DEV)
    cs_d, ctx_d, pl_d = torch.from_numpy(cs).to(DEV), torch.from_numpy(ctxs).to(DEV), torch.from_numpy(plen).to(DEV)
    out = {}
    for form in ("fused", "resnorm"):
        kc_d, vt_d = dev_bf16(kc), dev_bf16(vt)
        q_d = torch.zeros(B, H, hd, dtype=torch.bfloat16, device=DEV)
        xo = torch.full((B, K), 5.0, dtype=torch.bfloat16, device=DEV)
        if form == "fused":
            ranges = [(0, B)] if (B <= 16 or K <= 2048) else [(0, 16), (16, B - 16)]
            for r0, m in ranges:
                L.kr_linear_decode_narrow(DEC_ROPE_KV, ptr(xd[r0:]), K, ptr(pd[:, r0:]) if n_part else 0, n_part, ptr(xo[r0:]) if n_part else 0, K,
                                          ptr(Wd), ptr(bd), ptr(nd), 1e-6, 0, 0, 0, 0, 0, m, N, K, 8, 1, ptr(cs_d[r0:]), T, ptr(pl_d[r0:]),
                                          ptr(ctx_d[r0:]), ptr(q_d[r0:]), ptr(kc_d[r0:]), ptr(vt_d[r0:]), H, KVH, s_max,
                                          narrow_opts(part_rows=B) if (n_part and len(ranges) > 1) else None, 0)
        else:
            h = torch.full((B, K + 8), 9.0, dtype=torch.bfloat16, device=DEV)
            L.kr_decode_resnorm(ptr(xd), K, ptr(pd) if n_part else 0, n_part, B, ptr(xo) if n_part else 0, K, ptr(nd), 1e-6, ptr(h), K + 8, B, K, 0)
            assert (host(h)[:, K:] == 9.0).all()
            L.kr_linear_decode_narrow(DEC_ROPE_KV, ptr(h), K + 8, 0, 0, 0, 0, ptr(Wd), ptr(bd), 0, 1e-6, 0, 0, 0, 0, 0, B, N, K, 8, 1, ptr(cs_d), T,
                                      ptr(pl_d), ptr(ctx_d), ptr(q_d), ptr(kc_d), ptr(vt_d), H, KVH, s_max, None, 0)
        out[form] = (host(xo), host(q_d), host(kc_d), host(vt_d))
    for a, b, what in zip(out["fused"], out["resnorm"], ("x_new", "q", "K cache", "V^T cache")):
        np.testing.assert_array_equal(a, b, err_msg=what)
    if n_part:      # and the values are right: the residual sum against numpy
        ref_x = bf16_round(x + p[:n_part].sum(0, dtype=np.float32))
        assert np.abs(out["resnorm"][0] - ref_x).max() <= np.abs(ref_x).max() * 2 ** -7
    with pytest.raises(KarantaHipError):       # partial sums need a separate x_out
        L.kr_decode_resnorm(ptr(xd), K, ptr(pd), 1, B, ptr(xd), K, ptr(nd), 1e-6, ptr(xd), K, B, K, 0)


@pytest.mark.parametrize("h,w,rh,rw", [(100, 160, 56, 84), (60, 90, 140, 112), (308, 200, 308, 140), (56, 84, 56, 84),
                                       (17, 400, 28, 420), (1024, 1024, 980, 980)])
def test_image_front_end_bit_exact_vs_pil_path(L, h, w, rh, rw):
    """kr_image_resize_bicubic_u8 == PIL BICUBIC (uint8, bit for bit); kr_image_normalize_patchify == the host
    processor path's fp32 pixel_values (bit for bit: one rounding per operation, same patch order)."""
    from PIL import Image
    from karanta_ocr_amd import image_processing as IP
    rng = np.random.default_rng(h * 7 + w)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    src = torch.from_numpy(img).to(DEV)
    dst = torch.zeros(rh, rw, 3, dtype=torch.uint8, device=DEV)
    tmp = torch.zeros(h, rw, 3, dtype=torch.uint8, device=DEV)
    t_ = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    hb, hk = (t_(a) for a in IP.resample_tables(w, rw)) if rw != w else (None, None)
    vb, vk = (t_(a) for a in IP.resample_tables(h, rh)) if rh != h else (None, None)
    L.kr_image_resize_bicubic_u8(ptr(src), h, w, ptr(dst), rh, rw, ptr(tmp), ptr(hb), ptr(hk), hk.shape[1] if hk is not None else 0,
                                 ptr(vb), ptr(vk), vk.shape[1] if vk is not None else 0, 0)
    torch.cuda.synchronize()
    ref = np.asarray(Image.fromarray(img, "RGB").resize((rw, rh), resample=Image.BICUBIC)) if (rh, rw) != (h, w) else img
    np.testing.assert_array_equal(dst.cpu().numpy(), ref)
    # normalise + patchify of the resized page vs the host processor path on the original page
    gh, gw = rh // 14, rw // 14
    out = torch.full((gh * gw, 1176), 7.0, dtype=torch.float32, device=DEV)
    mean = (C.c_float * 3)(*[float(x) for x in IP.CLIP_MEAN]); std = (C.c_float * 3)(*[float(x) for x in IP.CLIP_STD])
    L.kr_image_normalize_patchify(ptr(dst), rh, rw, mean, std, 14, 2, 2, ptr(out), 0)
    torch.cuda.synchronize()
    pv, grid = IP.image_to_patches(ref, min_pixels=1, max_pixels=10 ** 9)    # ref is already rh x rw: no second resize
    assert grid == (1, gh, gw)
    np.testing.assert_array_equal(out.cpu().numpy(), pv)


def test_image_front_end_rejects_bad_geometry(L):
    x = torch.zeros(56, 56, 3, dtype=torch.uint8, device=DEV); o = torch.zeros(16, 1176, device=DEV)
    mean = (C.c_float * 3)(0, 0, 0); std = (C.c_float * 3)(1, 1, 1)
    with pytest.raises(KarantaHipError):
        L.kr_image_normalize_patchify(ptr(x), 50, 56, mean, std, 14, 2, 2, ptr(o), 0)
    with pytest.raises(KarantaHipError):     # both axes change but no tmp / tables
        L.kr_image_resize_bicubic_u8(ptr(x), 56, 56, ptr(x), 28, 28, 0, 0, 0, 0, 0, 0, 0, 0)


# ----------------------------------------------------------------------------- fp8 (e4m3fn) weights
def test_fp8_hardware_conversion_is_ocp_e4m3fn(L):
    """Every fp8 code through the conversion instruction the decode kernels use == the host's e4m3fn table
    (bias 7, max 448, NaN at 0x7F / 0xFF): pins the number format the quantiser targets."""
    from karanta_ocr_amd import weights as W
    codes = torch.arange(256, dtype=torch.uint8, device=DEV)
    out = torch.zeros(256, dtype=torch.bfloat16, device=DEV)
    L.kr_fp8_to_bf16(ptr(codes), ptr(out), 256, 0)
    got = out.float().cpu().numpy()
    ok = ~np.isnan(W.E4M3)
    np.testing.assert_array_equal(got[ok], W.E4M3[ok])
    assert np.isnan(got[~ok]).all()


def fp8_ref(x, Wf, bias=None, res=None, epi=EPI_NONE):
    """Reference of an fp8 linear: weights replaced by scale * e4m3(codes), then the bf16 reference."""
    from karanta_ocr_amd import weights as W
    q, sc = W.quantize_fp8_rows(Wf)
    return q, sc, ref_linear(x, (W.fp8_e4m3_to_f32(q) * sc[:, None]).astype(np.float32), bias, res, epi)


@pytest.mark.parametrize("M", [1, 8, 16])
@pytest.mark.parametrize("N,K,blocks,waves", [(16 * 37, 512, 3, 5), (16 * 37, 1536, 256, 4), (16 * 41, 3584, 5, 2), (16 * 45, 2048, 4, 3)])
def test_linear_wide_fp8_exact_on_small_integers(L, M, N, K, blocks, waves):
    """Integer weights in [-8, 8] are exact in e4m3; rows scaled by powers of two keep everything exact: a wrong k
    permutation between the fp8 fragment and x, a wrong chunk stride or a missing scale shows as an integer error."""
    from karanta_ocr_amd import weights as W
    rng = np.random.default_rng(M + N + K)
    x = ints(rng, M, K)
    Wi = rng.integers(-8, 9, size=(N, K)).astype(np.float32)
    Wi[:, K // 2 + 40:] = 0
    Wi[::3, 100:300] = 1
    Wi[:, 0] = 448                                    # every row's max is 448 -> the row scale is exactly its power of two
    scale_pow = (2.0 ** rng.integers(-2, 3, size=N)).astype(np.float32)
    q, sc = W.quantize_fp8_rows(Wi * scale_pow[:, None])
    np.testing.assert_array_equal(W.fp8_e4m3_to_f32(q) * sc[:, None], Wi * scale_pow[:, None])    # exact by construction
    xd, qd, sd = dev_bf16(x), torch.from_numpy(W.pack_w16x64_fp8(q)).to(DEV), torch.from_numpy(sc).to(DEV)
    out = torch.zeros(M, N, dtype=torch.float32, device=DEV)
    L.kr_linear_decode_wide_fp8(DEC_PLAIN, ptr(xd), K, ptr(qd), ptr(sd), 0, 0, 1e-6, 0, 0, 0, ptr(out), N, M, N, K, blocks, waves,
                                0, 0, 0)
    np.testing.assert_array_equal(out.cpu().numpy(), ref_linear(x, Wi * scale_pow[:, None]))


@pytest.mark.parametrize("M,K,blocks,waves", [(8, 1536, 256, 5), (16, 3584, 9, 8), (3, 512, 2, 3)])
def test_linear_wide_fp8_silu8_norm(L, M, K, blocks, waves):
    from karanta_ocr_amd import weights as W
    rng = np.random.default_rng(262 + M)
    ff = 1016
    x, Wp, nw = rnd(rng, M, K, scale=2.0), rnd(rng, 2 * ff, K, scale=K ** -0.5), bf16_round(1 + 0.1 * rnd(rng, K))
    xn = bf16_round(O.rms_norm(x, nw, 1e-6, O._Policy("bf16")))
    q, sc, ref = fp8_ref(xn, Wp, epi=EPI_SILU_MUL8)
    xd, qd, sd, nd = dev_bf16(x), torch.from_numpy(W.pack_w16x64_fp8(q)).to(DEV), torch.from_numpy(sc).to(DEV), dev_bf16(nw)
    out = torch.zeros(M, ff, dtype=torch.bfloat16, device=DEV)
    L.kr_linear_decode_wide_fp8(DEC_SILU8, ptr(xd), K, ptr(qd), ptr(sd), 0, ptr(nd), 1e-6, 0, 0, ptr(out), 0, ff, M, 2 * ff, K, blocks,
                                waves, 0, 0, 0)
    assert_close_bf16(host(out), ref, what="wide fp8 silu8")


@pytest.mark.parametrize("M,N,K,waves,ksplit", [(8, 1536, 8960, 16, 2), (16, 96, 8960, 8, 1), (3, 48, 256, 8, 1), (8, 1536, 1536, 8, 1)])
def test_linear_narrow_fp8_plain_and_slabs(L, M, N, K, waves, ksplit):
    from karanta_ocr_amd import weights as W
    rng = np.random.default_rng(M + N + K + 5)
    x, Wf = rnd(rng, M, K), rnd(rng, N, K, scale=K ** -0.5)
    q, sc, ref = fp8_ref(x, Wf)
    xd, qd, sd = dev_bf16(x), torch.from_numpy(W.pack_w16x64_fp8(q)).to(DEV), torch.from_numpy(sc).to(DEV)
    if ksplit > 1:
        slabs = torch.zeros(ksplit, M, N, dtype=torch.float32, device=DEV)
        L.kr_linear_decode_narrow_fp8(DEC_PLAIN, ptr(xd), K, 0, 0, 0, 0, ptr(qd), ptr(sd), 0, 0, 1e-6, 0, 0, 0, ptr(slabs), N, M, N, K,
                                      waves, ksplit, 0, 0, 0, 0, 0, 0, 0, 0, 0, 64, None, 0)
        got = slabs.sum(0).cpu().numpy()
    else:
        out = torch.zeros(M, N, dtype=torch.float32, device=DEV)
        L.kr_linear_decode_narrow_fp8(DEC_PLAIN, ptr(xd), K, 0, 0, 0, 0, ptr(qd), ptr(sd), 0, 0, 1e-6, 0, 0, 0, ptr(out), N, M, N, K,
                                      waves, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 64, None, 0)
        got = out.cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("M,N,K", [(1, 16, 64), (255, 256, 64), (257, 272, 192), (700, 528, 320), (2049, 768, 1280), (300, 1536, 8960)])
def test_gemm_fp8_exact_on_small_integers(L, M, N, K):
    """kr_gemm_fp8 (weight-only fp8 prefill GEMM): integer weights exact in e4m3, power-of-two row scales, integer
    activations -> exact result; catches a wrong half-block / k-group / row-block address, a wrong conversion order, a
    missing scale, a miscounted vmcnt (stale LDS tile)."""
    from karanta_ocr_amd import weights as W
    rng = np.random.default_rng(M + N + K + 9)
    A = ints(rng, M, K)
    Wi = rng.integers(-8, 9, size=(N, K)).astype(np.float32)
    if K > 256:
        Wi[:, 200:] = 0
        Wi[::5, K - 40:] = rng.integers(-1, 2, size=(len(Wi[::5]), 40))
    Wi[:, 0] = 448
    A[:, 0] = 0                                                        # (keeps the sums small: column 0 only fixes the scales)
    scale_pow = (2.0 ** rng.integers(-2, 3, size=N)).astype(np.float32)
    q, sc = W.quantize_fp8_rows(Wi * scale_pow[:, None])
    np.testing.assert_array_equal(W.fp8_e4m3_to_f32(q) * sc[:, None], Wi * scale_pow[:, None])
    Ad, qd, sd = dev_bf16(A), torch.from_numpy(W.pack_w16x64_fp8(q)).to(DEV), torch.from_numpy(sc).to(DEV)
    ldc = (N + 7) // 8 * 8 + 8
    C_ = torch.full((M, ldc), 7.0, dtype=torch.bfloat16, device=DEV)
    L.kr_gemm_fp8(ptr(Ad), K, ptr(qd), ptr(sd), 0, 0, 0, ptr(C_), ldc, M, N, K, EPI_NONE, 0)
    got = host(C_)
    np.testing.assert_array_equal(got[:, :N], bf16_round(ref_linear(A, Wi * scale_pow[:, None])))   # exact sums, one rounding
    assert (got[:, N:] == 7.0).all(), "columns beyond N untouched"


@pytest.mark.parametrize("epi", [EPI_NONE, EPI_SILU_MUL8])
def test_gemm_fp8_matches_dequantised_reference(L, epi):
    rng = np.random.default_rng(91 + epi)
    from karanta_ocr_amd import weights as W
    M, N, K = 1394, 2048 if epi == EPI_NONE else 1024, 1536
    A, Wf = rnd(rng, M, K), rnd(rng, N, K, scale=K ** -0.5)
    bias = rnd(rng, N, scale=0.1)
    res = rnd(rng, M, N) if epi == EPI_NONE else None
    q, sc, ref = fp8_ref(A, Wf, bias, res, epi)
    Ad, qd, sd, bd = dev_bf16(A), torch.from_numpy(W.pack_w16x64_fp8(q)).to(DEV), torch.from_numpy(sc).to(DEV), dev_bf16(bias)
    nc = N // 2 if epi == EPI_SILU_MUL8 else N
    C_ = torch.zeros(M, nc, dtype=torch.bfloat16, device=DEV)
    rd = dev_bf16(res) if res is not None else None
    L.kr_gemm_fp8(ptr(Ad), K, ptr(qd), ptr(sd), ptr(bd), ptr(rd) if rd is not None else 0, N if rd is not None else 0, ptr(C_), nc, M, N, K,
                  epi, 0)
    assert_close_bf16(host(C_), ref, what=f"gemm fp8 epi {epi}")
    with pytest.raises(KarantaHipError):
        L.kr_gemm_fp8(ptr(Ad), K, ptr(qd), ptr(sd), 0, 0, 0, ptr(C_), nc, M, N, K, EPI_QUICK_GELU, 0)
    with pytest.raises(KarantaHipError):
        L.kr_gemm_fp8(ptr(Ad), K, ptr(qd), ptr(sd), 0, 0, 0, ptr(C_[:, 4:]), nc, M, N, K, epi, 0)    # C not 16-byte aligned


@pytest.mark.parametrize("rows,K", [(1, 64), (37, 1536), (130, 3584), (9, 8960), (5, 18944), (3, 24576)])
def test_quantize_rows_fp8_is_bit_identical_to_the_host_quantiser(L, rows, K):
    """kr_quantize_rows_fp8 (dynamic per-token e4m3 codes of the W8A8 prefill): scales and codes equal
    weights.quantize_fp8_rows bit for bit — rows of very different magnitude, an all-zero row, values on rounding ties
    and in the subnormal range of the scaled row, row strides wider than K, bytes beyond K untouched."""
    from karanta_ocr_amd import weights as W
    rng = np.random.default_rng(rows + K)
    x = bf16_round(rng.standard_normal((rows, K)).astype(np.float32) * rng.uniform(1e-3, 40, (rows, 1)).astype(np.float32))
    x[0, :: 7] = bf16_round(x[0, 0] * 2.0 ** -rng.integers(0, 14, size=len(x[0, ::7])).astype(np.float32))   # down to the subnormals
    if rows > 2:
        x[1] = 0
        x[2, :40] = bf16_round(np.linspace(-1, 1, 40).astype(np.float32) * np.abs(x[2]).max())                 # grid points and ties
    ldx, ldq = K + 8, (K + 15) // 16 * 16 + 16
    xd = dev_bf16(np.concatenate([x, np.full((rows, 8), 3.0, np.float32)], 1))
    qd = torch.full((rows, ldq), 0xAB, dtype=torch.uint8, device=DEV)
    sd = torch.zeros(rows, dtype=torch.float32, device=DEV)
    L.kr_quantize_rows_fp8(ptr(xd), ldx, ptr(qd), ldq, ptr(sd), rows, K, 0)
    q_ref, s_ref = W.quantize_fp8_rows(x)
    got_q, got_s = qd.cpu().numpy(), sd.cpu().numpy()
    np.testing.assert_array_equal(got_s, s_ref)
    np.testing.assert_array_equal(got_q[:, :K], q_ref)
    assert (got_q[:, K:] == 0xAB).all(), "bytes beyond K untouched"
    with pytest.raises(KarantaHipError):
        L.kr_quantize_rows_fp8(ptr(xd), ldx, ptr(qd), ldq, ptr(sd), rows, K + 4, 0)


@pytest.fixture(params=[(1, 0), (1, 1), (0, 0)], ids=["mx-scaled-32x32x64", "mx-scaled-two-k-tiles-per-barrier-pair", "fp8-16x16x32"])
def fp8_mx(request):
    """The instruction forms of kr_gemm_fp8a: KARANTA_FP8_MX = 1 (default: v_mfma_scale_f32_32x32x64_f8f6f4, block scales 2^0,
    twice the bf16 rate; KARANTA_FP8_MX2 = 1: two K-tiles per barrier pair where K % 128 == 0) and 0 (v_mfma_f32_16x16x32_fp8_fp8
    in the bf16 kernel's pipeline)."""
    import os
    os.environ["KARANTA_FP8_MX"], os.environ["KARANTA_FP8_MX2"] = str(request.param[0]), str(request.param[1])
    yield request.param
    os.environ.pop("KARANTA_FP8_MX", None)
    os.environ.pop("KARANTA_FP8_MX2", None)


@pytest.mark.parametrize("M,N,K", [(1, 32, 64), (255, 256, 64), (257, 288, 192), (700, 544, 320), (2049, 768, 1280), (300, 1536, 8960),
                                   (130, 16, 128), (513, 320, 128), (300, 256, 3584)])
def test_gemm_fp8a_exact_on_small_integers(L, fp8_mx, M, N, K):
    """kr_gemm_fp8a (both operands e4m3 codes through v_mfma_f32_16x16x32_fp8_fp8): integer activations and weights that are
    exact in e4m3, power-of-two scales on both sides -> exact result.  Catches a k-order mismatch between the A and W
    fragments, a wrong LDS swizzle of the 32-byte A rows, a wrong row / half address of the staging, a missing scale on
    either side, a miscounted vmcnt (stale LDS tile)."""
    from karanta_ocr_amd import weights as W
    rng = np.random.default_rng(M + N + K + 19)
    A = rng.integers(-4, 5, size=(M, K)).astype(np.float32)
    Wi = rng.integers(-8, 9, size=(N, K)).astype(np.float32)
    if K > 256:
        Wi[:, 200:] = 0
        Wi[::5, K - 40:] = rng.integers(-1, 2, size=(len(Wi[::5]), 40))
        A[::3, 100:180] = 1
    Wi[:, 0] = 448
    A[:, 0] = 0
    A[:, 1] = 448                                                       # every row's max is 448: its scale is exactly its power of two
    Wi[:, 1] = 0
    a_pow = (2.0 ** rng.integers(-3, 2, size=M)).astype(np.float32)
    w_pow = (2.0 ** rng.integers(-2, 3, size=N)).astype(np.float32)
    Af, Wf = A * a_pow[:, None], Wi * w_pow[:, None]
    qa, sa = W.quantize_fp8_rows(Af)
    qw, sw = W.quantize_fp8_rows(Wf)
    np.testing.assert_array_equal(W.fp8_e4m3_to_f32(qa) * sa[:, None], Af)
    np.testing.assert_array_equal(W.fp8_e4m3_to_f32(qw) * sw[:, None], Wf)
    lda = (K + 15) // 16 * 16 + 32
    qa_pad = np.full((M, lda), 0x7E, np.uint8)                           # (a read beyond K would add 448s)
    qa_pad[:, :K] = qa
    ad, sad = torch.from_numpy(qa_pad).to(DEV), torch.from_numpy(sa).to(DEV)
    qd, sd = torch.from_numpy(W.pack_w16x64_fp8(qw)).to(DEV), torch.from_numpy(sw).to(DEV)
    ldc = (N + 7) // 8 * 8 + 8
    C_ = torch.full((M, ldc), 7.0, dtype=torch.bfloat16, device=DEV)
    L.kr_gemm_fp8a(ptr(ad), lda, ptr(sad), ptr(qd), ptr(sd), 0, 0, 0, ptr(C_), ldc, M, N, K, EPI_NONE, 0)
    got = host(C_)
    np.testing.assert_array_equal(got[:, :N], bf16_round(ref_linear(Af, Wf)))          # exact sums, one rounding
    assert (got[:, N:] == 7.0).all(), "columns beyond N untouched"


@pytest.mark.parametrize("epi", [EPI_NONE, EPI_SILU_MUL8])
def test_quantize_then_gemm_fp8a_matches_the_fake_quantised_reference(L, fp8_mx, epi):
    """The W8A8 prefill pair as the engine runs it: kr_quantize_rows_fp8 on bf16 activations, kr_gemm_fp8a on the codes —
    against the oracle's statement of it (fake_quant_rows_fp8 of A, dequantised W, f32 product), bias / residual /
    SiLU*mul epilogues, a ragged M with a partial last tile."""
    rng = np.random.default_rng(191 + epi)
    from karanta_ocr_amd import weights as W
    M, N, K = 1394, 2048 if epi == EPI_NONE else 1024, 1536
    A, Wf = rnd(rng, M, K) * rng.uniform(0.2, 5, (M, 1)).astype(np.float32), rnd(rng, N, K, scale=K ** -0.5)
    A = bf16_round(A)
    bias = rnd(rng, N, scale=0.1)
    res = rnd(rng, M, N) if epi == EPI_NONE else None
    q, sc, ref = fp8_ref(O.fake_quant_rows_fp8(A), Wf, bias, res, epi)
    Ad, qd, sd, bd = dev_bf16(A), torch.from_numpy(W.pack_w16x64_fp8(q)).to(DEV), torch.from_numpy(sc).to(DEV), dev_bf16(bias)
    a8 = torch.zeros(M, K, dtype=torch.uint8, device=DEV)
    a_s = torch.zeros(M, dtype=torch.float32, device=DEV)
    nc = N // 2 if epi == EPI_SILU_MUL8 else N
    C_ = torch.zeros(M, nc, dtype=torch.bfloat16, device=DEV)
    rd = dev_bf16(res) if res is not None else None
    L.kr_quantize_rows_fp8(ptr(Ad), K, ptr(a8), K, ptr(a_s), M, K, 0)
    L.kr_gemm_fp8a(ptr(a8), K, ptr(a_s), ptr(qd), ptr(sd), ptr(bd), ptr(rd) if rd is not None else 0, N if rd is not None else 0,
                   ptr(C_), nc, M, N, K, epi, 0)
    assert_close_bf16(host(C_), ref, what=f"quantise + gemm fp8a epi {epi}")   # (ref_linear takes A as given: f64 product)
    with pytest.raises(KarantaHipError):
        L.kr_gemm_fp8a(ptr(a8), K, ptr(a_s), ptr(qd), ptr(sd), 0, 0, 0, ptr(C_), nc, M, N, K, EPI_QUICK_GELU, 0)
    with pytest.raises(KarantaHipError):
        L.kr_gemm_fp8a(ptr(a8), K + 8, ptr(a_s), ptr(qd), ptr(sd), 0, 0, 0, ptr(C_), nc, M, N, K, epi, 0)    # lda not a multiple of 16


def test_gumbel_argmax_matches_oracle_noise(L):
    """kr_gumbel_argmax partials against the oracle's sample_scores: same counter-based noise (integer hash bit-exact,
    the two logs within float rounding), T = 0 rows untouched, ties to the lowest index, sampled frequencies follow
    softmax(logits / T)."""
    rng = np.random.default_rng(17)
    B, V, n_part = 4, 5003, 7
    logits = rng.standard_normal((B, V)).astype(np.float32) * 2
    logits[1, 100] = logits[1, 4000] = 50.0                      # T = 0 row with an exact tie
    temps = np.asarray([0.7, 0.0, 0.1, 1.5], np.float32)
    seeds = np.asarray([1, 2, 0xFFFFFFFF, 12345], np.uint32)
    plen = np.asarray([10, 5, 7, 0], np.int32); ctx = np.asarray([12, 5, 30, 2], np.int32)   # n = ctx + 1 - plen
    ld = torch.from_numpy(logits).to(DEV)
    av = torch.zeros(B, n_part, device=DEV); ai = torch.zeros(B, n_part, dtype=torch.int32, device=DEV)
    td, sd = torch.from_numpy(temps).to(DEV), torch.from_numpy(seeds.view(np.int32)).to(DEV)
    pd, cd = torch.from_numpy(plen).to(DEV), torch.from_numpy(ctx).to(DEV)
    L.kr_gumbel_argmax(ptr(ld), V, V, ptr(td), ptr(sd), ptr(cd), ptr(pd), ptr(av), ptr(ai), n_part, B, 0)
    torch.cuda.synchronize()
    avh, aih = av.cpu().numpy(), ai.cpu().numpy()
    per = ((-(-V // n_part)) + 3) & ~3
    for b in range(B):
        sc = O.sample_scores(logits[b], float(temps[b]), int(seeds[b]), int(ctx[b] + 1 - plen[b]))
        for p_ in range(n_part):
            seg = sc[p_ * per:min(V, (p_ + 1) * per)]
            j = int(np.argmax(seg))
            top2 = np.sort(seg)[-2:]
            if top2[1] - top2[0] > 1e-4:                         # the device's logf and numpy's differ in the last bits
                assert aih[b, p_] == p_ * per + j
            np.testing.assert_allclose(avh[b, p_], seg[j], rtol=2e-6, atol=2e-5)
    assert aih[1].min() >= 0 and 100 in aih[1] and avh[1].max() == 50.0 and aih[1][np.argmax(avh[1])] == 100
    # distribution: 3 candidates with probabilities softmax([2, 1, 0] / T), 4000 draws through n
    lg = np.full((1, 64), -1e4, np.float32); lg[0, [3, 17, 40]] = [2.0, 1.0, 0.0]
    T = 0.8
    want = np.exp(np.asarray([2.0, 1.0, 0.0]) / T); want /= want.sum()
    ldd = torch.from_numpy(lg).to(DEV); t1 = torch.tensor([T], device=DEV); s1 = torch.tensor([99], dtype=torch.int32, device=DEV)
    p1 = torch.zeros(1, dtype=torch.int32, device=DEV)
    counts = {3: 0, 17: 0, 40: 0}
    a1 = torch.zeros(1, 1, device=DEV); i1 = torch.zeros(1, 1, dtype=torch.int32, device=DEV)
    draws = 4000
    cs = torch.arange(draws, dtype=torch.int32, device=DEV)
    picks = []
    for n in range(draws):
        L.kr_gumbel_argmax(ptr(ldd), 64, 64, ptr(t1), ptr(s1), ptr(cs[n:]), ptr(p1), ptr(a1), ptr(i1), 1, 1, 0)
        picks.append(i1.clone())
    torch.cuda.synchronize()
    for t_ in torch.cat(picks).cpu().numpy().ravel():
        counts[int(t_)] += 1
    got = np.asarray([counts[3], counts[17], counts[40]]) / draws
    assert np.abs(got - want).max() < 0.03, (got, want)


def _unmix32(h: int) -> int:
    """Inverse of the sampler's 32-bit finaliser (a bijection): the x with mix32(x) == h."""
    M = 0xFFFFFFFF
    h ^= h >> 16
    h = (h * pow(0x846ca68b, -1, 1 << 32)) & M
    h ^= (h >> 15) ^ (h >> 30)
    h = (h * pow(0x7feb352d, -1, 1 << 32)) & M
    h ^= h >> 16
    return h


def test_gumbel_noise_is_finite_at_the_extreme_hash_codes(L):
    """with u = ((h >> 8) + 0.5) * 2^-24 the top code rounds to u == 1 and the noise is +inf, so that
    token wins whatever its logit.  Force the hash of one token to 0xFFFFFFFF (and of another to 0) by inverting the
    finaliser: both draws must be finite, equal to the oracle's, and the -50 logit must not win."""
    V, tok_hi, tok_lo = 64, 5, 9
    assert int(O._mix32(np.uint64(_unmix32(0xFFFFFFFF)))) == 0xFFFFFFFF
    old_u = (np.float32(0xFFFFFF) + np.float32(0.5)) * np.float32(2.0 ** -24)
    assert old_u == np.float32(1.0)                       # the defect this test pins
    for h_forced, tok in ((0xFFFFFFFF, tok_hi), (0, tok_lo)):
        base = (_unmix32(h_forced) - tok) & 0xFFFFFFFF       # mix32(base + tok) == h_forced
        seed = _unmix32(base)                                # n = 0: base = mix32(seed)
        g = O.gumbel_noise(seed, 0, V)
        assert np.isfinite(g).all() and (g[tok] == g.max() if h_forced else g[tok] == g.min())
        logits = np.zeros((1, V), np.float32)
        logits[0, tok] = -50.0
        ld = torch.from_numpy(logits).to(DEV)
        t1 = torch.tensor([1.0], device=DEV)
        s1 = torch.from_numpy(np.asarray([seed], np.uint32).view(np.int32)).to(DEV)
        c1 = torch.tensor([-1], dtype=torch.int32, device=DEV); p1 = torch.zeros(1, dtype=torch.int32, device=DEV)
        av = torch.zeros(1, V, device=DEV); ai = torch.zeros(1, V, dtype=torch.int32, device=DEV)
        # one partial per token (n_part = V/4 groups of 4): read back every group's best noisy score
        L.kr_gumbel_argmax(ptr(ld), V, V, ptr(t1), ptr(s1), ptr(c1), ptr(p1), ptr(av), ptr(ai), V // 4, 1, 0)
        torch.cuda.synchronize()
        vals = av.cpu().numpy()[0, :V // 4]
        assert np.isfinite(vals).all(), vals
        sc = O.sample_scores(logits[0], 1.0, seed, 0)
        np.testing.assert_allclose(vals, sc.reshape(V // 4, 4).max(1), rtol=2e-6, atol=2e-5)
        assert int(ai.cpu().numpy()[0, :V // 4][np.argmax(vals)]) != tok


def test_sample_greedy_freeze_finished(L):
    """Flag bit 1: a finished sequence neither advances ctx_len nor writes history (slot scheduler)."""
    B, d, n_part = 3, 64, 4
    rng = np.random.default_rng(91)
    av = torch.tensor([[1, 5, 2, 0], [3, 1, 9, 2], [4, 4, 4, 8]], dtype=torch.float32, device=DEV)
    ai = torch.tensor([[10, 11, 12, 13], [20, 21, 22, 23], [30, 31, 32, 33]], dtype=torch.int32, device=DEV)
    table = rnd(rng, 64, d); td = dev_bf16(table)
    tok = torch.zeros(B, dtype=torch.int32, device=DEV); hist = torch.full((6, B), -1, dtype=torch.int32, device=DEV)
    plen = torch.tensor([4, 4, 4], dtype=torch.int32, device=DEV)
    ctx = torch.tensor([5, 6, 7], dtype=torch.int32, device=DEV)
    fin = torch.tensor([0, 1, 0], dtype=torch.int32, device=DEV)
    eos = torch.tensor([33], dtype=torch.int32, device=DEV)
    xn = torch.zeros(B, d, dtype=torch.bfloat16, device=DEV)
    for _ in range(2):
        L.kr_sample_greedy(ptr(av), ptr(ai), n_part, ptr(td), d, ptr(tok), ptr(hist), B, ptr(plen), ptr(ctx), ptr(fin), ptr(eos),
                           1, 63, 2, ptr(xn), B, 0)
    torch.cuda.synchronize()
    assert tok.cpu().tolist() == [11, 63, 63]                 # row 1 frozen all along, row 2 hit EOS in call 1
    assert ctx.cpu().tolist() == [7, 6, 8]                    # +2, frozen, +1 then frozen
    assert fin.cpu().tolist() == [0, 1, 1]
    h = hist.cpu().numpy()
    assert h[2, 0] == 11 and h[3, 0] == 11 and h[4, 2] == 33 and (h[:, 1] == -1).all() and (h[5] == -1).all()
    np.testing.assert_array_equal(host(xn), table[[11, 63, 63]])


def test_sample_greedy_eos_and_pad(L):
    B, d, n_part = 3, 64, 5
    rng = np.random.default_rng(81)
    table = rnd(rng, 512, d); td = dev_bf16(table)
    av = torch.full((B, n_part), -1.0, device=DEV); ai = torch.zeros(B, n_part, dtype=torch.int32, device=DEV)
    av[0, 2] = 3; ai[0, 2] = 11; av[1, 4] = 2; ai[1, 4] = 497; av[2, 0] = 1; ai[2, 0] = 300
    av[2, 3] = 1; ai[2, 3] = 299   # tie on value -> lower index 299
    tok = torch.zeros(B, dtype=torch.int32, device=DEV); hist = torch.full((4, B), -1, dtype=torch.int32, device=DEV)
    plen = torch.tensor([10, 20, 30], dtype=torch.int32, device=DEV); ctx = torch.tensor([9, 19, 29], dtype=torch.int32, device=DEV)
    fin = torch.zeros(B, dtype=torch.int32, device=DEV); eos = torch.tensor([497, 496], dtype=torch.int32, device=DEV)
    xn = torch.zeros(B, d, dtype=torch.bfloat16, device=DEV)
    for _ in range(2):
        L.kr_sample_greedy(ptr(av), ptr(ai), n_part, ptr(td), d, ptr(tok), ptr(hist), B, ptr(plen), ptr(ctx), ptr(fin),
                           ptr(eos), 2, 496, 0, ptr(xn), B, 0)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(hist.cpu().numpy()[:2], [[11, 497, 299], [11, 496, 299]])   # pad after EOS
    np.testing.assert_array_equal(fin.cpu().numpy(), [0, 1, 0])
    np.testing.assert_array_equal(ctx.cpu().numpy(), [11, 21, 31])
    np.testing.assert_array_equal(host(xn), table[[11, 496, 299]])


@pytest.mark.parametrize("H,KVH", [(2, 1), (12, 2), (28, 4)])
@pytest.mark.parametrize("n_split", [1, 4, 8])
def test_attn_partials_merged_by_o_proj_prologue(L, H, KVH, n_split):
    """Engine path: attention leaves split partials (out == NULL); the o_proj decode linear merges
    them in its prologue, multiplies by W_o and adds the residual."""
    rng = np.random.default_rng(H * 10 + n_split + 1)
    hd, s_max, d = 128, 1024, 256
    ctxs = [0, 70, 130, 1023]
    B = len(ctxs)
    kc = np.zeros((B, KVH, s_max, hd), np.float32); vc = np.zeros((B, KVH, s_max, hd), np.float32)
    for b, c in enumerate(ctxs):
        kc[b, :, :c + 1] = rnd(rng, KVH, c + 1, hd)
        vc[b, :, :c + 1] = rnd(rng, KVH, c + 1, hd)
    vt = POS.vt_blocks(vc)
    q = rnd(rng, B, H, hd)
    Wo, X = rnd(rng, d, H * hd, scale=(H * hd) ** -0.5), rnd(rng, B, d)
    kc_d, vt_d, q_d, Wd, Xd = dev_bf16(kc), dev_bf16(vt), dev_bf16(q), dev_bf16(pack_w16x64(Wo)), dev_bf16(X)
    ctx_d = torch.tensor(ctxs, dtype=torch.int32, device=DEV)
    ws = torch.zeros(B * H * n_split * (hd + 4), dtype=torch.float32, device=DEV)
    L.kr_attn_decode_fused(ptr(q_d), ptr(kc_d), ptr(vt_d), ptr(ctx_d), 0, ptr(ws), 0, B, H, KVH, hd, s_max, n_split,
                           hd ** -0.5, 0)
    dec_call(L, DEC_PLAIN, 0, 0, ptr(Wd), B, d, H * hd, out=ptr(Xd), ldc=d, res=ptr(Xd), ldr=d, waves=8, attn=ptr(ws),
             attn_split=n_split)
    attn = np.concatenate([np_attention(q[b][:, None], kc[b, :, :c + 1], vc[b, :, :c + 1], hd ** -0.5, False)
                           for b, c in enumerate(ctxs)])
    assert_close_bf16(host(Xd), ref_linear(bf16_round(attn), Wo, res=X), rel=2 ** -6, abs_=3e-2, what="o_proj with merge")


@pytest.mark.parametrize("H,KVH", [(12, 2), (28, 4), (3, 1)])
@pytest.mark.parametrize("n_split", [8, 16, 32])
@pytest.mark.parametrize("config5", [False, True])
def test_attn_decode_partials_then_merge_launch(L, H, KVH, n_split, config5):
    """The engine's deterministic path: split-KV partials (8 splits x 8 waves, 16 x 4, 32 x 2 workgroup shapes) and the
    merge launch, at the contexts of the bench's decode loop; config5: at the contexts BASELINE config 5 decodes at (a
    4988-token prompt, then 128 tokens: ctx 4988 .. 5116) up to the last row of its 5184-row cache."""
    rng = np.random.default_rng(H * 10 + n_split + 7)
    hd, s_max = 128, 5184 if config5 else 2560
    ctxs = [63, 4987, 4988, 5052, 5116, 5183] if config5 else [0, 31, 32, 1393, 1906, 2431, 2559]
    B = len(ctxs)
    kc = np.zeros((B, KVH, s_max, hd), np.float32); vc = np.zeros((B, KVH, s_max, hd), np.float32)
    for b, c in enumerate(ctxs):
        kc[b, :, :c + 1] = rnd(rng, KVH, c + 1, hd)
        vc[b, :, :c + 1] = rnd(rng, KVH, c + 1, hd)
    vt = POS.vt_blocks(vc)
    q = rnd(rng, B, H, hd)
    kc_d, vt_d, q_d = dev_bf16(kc), dev_bf16(vt), dev_bf16(q)
    ctx_d = torch.tensor(ctxs, dtype=torch.int32, device=DEV)
    ws = torch.full((B * H * n_split * (hd + 4),), 9.0, dtype=torch.float32, device=DEV)
    o_d = torch.zeros(B, H * hd, dtype=torch.bfloat16, device=DEV)
    L.kr_attn_decode_fused(ptr(q_d), ptr(kc_d), ptr(vt_d), ptr(ctx_d), 0, ptr(ws), 0, B, H, KVH, hd, s_max, n_split, hd ** -0.5, 0)
    L.kr_attn_decode_merge(ptr(ws), ptr(o_d), B, H, hd, n_split, 0)
    got = host(o_d)
    for b, c in enumerate(ctxs):
        ref = np_attention(q[b][:, None], kc[b, :, :c + 1], vc[b, :, :c + 1], hd ** -0.5, False)
        assert_close_bf16(got[b:b + 1], ref, rel=2 ** -6, abs_=2e-2, what=f"partials + merge b={b} n_split={n_split}")
    # the serving form: slots whose finished flag is set are skipped (their partials stay as they were), live slots get the same bits
    fin = torch.tensor([b % 3 == 1 for b in range(B)], dtype=torch.int32, device=DEV)
    ws2 = torch.full_like(ws, 5.0)
    L.kr_attn_decode_slots(ptr(q_d), ptr(kc_d), ptr(vt_d), ptr(ctx_d), ptr(fin), ptr(ws2), B, H, KVH, hd, s_max, n_split, hd ** -0.5, 0)
    a, b2 = ws.cpu().numpy().reshape(B, -1), ws2.cpu().numpy().reshape(B, -1)
    for b in range(B):
        if b % 3 == 1:
            assert (b2[b] == 5.0).all(), f"slot {b} is finished: untouched"
        else:
            np.testing.assert_array_equal(b2[b], a[b])


@pytest.mark.parametrize("H,KVH", [(2, 1), (12, 2), (28, 4)])
@pytest.mark.parametrize("n_split", [1, 4, 8, 16])
@pytest.mark.parametrize("long_ctx", [False, True])
def test_attn_decode_fused(L, H, KVH, n_split, long_ctx):
    rng = np.random.default_rng(H * 10 + n_split)
    hd, s_max = 128, 2560 if long_ctx else 2048
    # long_ctx: the contexts of the bench's decode loop (1394 .. 2417) and the last cache row
    ctxs = [1393, 1394, 2000, 2431, 2559] if long_ctx else [0, 64, 100, 1279, 2047]
    B = len(ctxs)
    kc = np.zeros((B, KVH, s_max, hd), np.float32); vc = np.zeros((B, KVH, s_max, hd), np.float32)
    for b, c in enumerate(ctxs):
        kc[b, :, :c + 1] = rnd(rng, KVH, c + 1, hd)
        vc[b, :, :c + 1] = rnd(rng, KVH, c + 1, hd)
    vt = POS.vt_blocks(vc)
    q = rnd(rng, B, H, hd)
    kc_d, vt_d, q_d = dev_bf16(kc), dev_bf16(vt), dev_bf16(q)
    ctx_d = torch.tensor(ctxs, dtype=torch.int32, device=DEV)
    ws = torch.zeros(B * H * n_split * (hd + 4), dtype=torch.float32, device=DEV)
    cnt = torch.zeros(B * KVH, dtype=torch.int32, device=DEV)
    o_d = torch.zeros(B, H * hd, dtype=torch.bfloat16, device=DEV)
    if n_split > 1 and not L.experiments:
        # the in-launch merge is an experiment (measured slower than the merge launch): the shipped library refuses it
        with pytest.raises(KarantaHipError, match="KR_EXPERIMENTS"):
            L.kr_attn_decode_fused(ptr(q_d), ptr(kc_d), ptr(vt_d), ptr(ctx_d), ptr(o_d), ptr(ws), ptr(cnt), B, H, KVH, hd, s_max,
                                   n_split, hd ** -0.5, 0)
        return
    for _ in range(2):
        L.kr_attn_decode_fused(ptr(q_d), ptr(kc_d), ptr(vt_d), ptr(ctx_d), ptr(o_d), ptr(ws), ptr(cnt), B, H, KVH, hd, s_max,
                               n_split, hd ** -0.5, 0)
    got = host(o_d)
    assert not cnt.cpu().numpy().any()
    for b, c in enumerate(ctxs):
        ref = np_attention(q[b][:, None], kc[b, :, :c + 1], vc[b, :, :c + 1], hd ** -0.5, False)
        assert_close_bf16(got[b:b + 1], ref, rel=2 ** -6, abs_=2e-2, what=f"fused decode attention b={b}")


def test_graph_capture_and_replay(L):
    """A captured kr_* launch replays with device-side state (the decode step pattern)."""
    stream = torch.cuda.Stream()
    s = stream.cuda_stream
    B, V, d = 2, 512, 64
    rng = np.random.default_rng(5)
    table = dev_bf16(rnd(rng, V, d))
    logits = torch.full((B, V), -1.0, device=DEV); logits[0, 7] = 3; logits[1, 9] = 3
    tok = torch.zeros(B, dtype=torch.int32, device=DEV); hist = torch.zeros(8, B, dtype=torch.int32, device=DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV); ctx = torch.zeros(B, dtype=torch.int32, device=DEV)
    fin = torch.zeros(B, dtype=torch.int32, device=DEV); eos = torch.tensor([1], dtype=torch.int32, device=DEV)
    xn = torch.zeros(B, d, dtype=torch.bfloat16, device=DEV)
    torch.cuda.synchronize()
    L.kr_graph_begin_capture(s)
    L.kr_argmax_embed(ptr(logits), V, V, ptr(table), d, ptr(tok), ptr(hist), ptr(step), ptr(ctx), ptr(fin), ptr(eos), 1, 0, 1,
                      ptr(xn), B, B, s)
    g = C.c_void_p()
    L.kr_graph_end_capture(s, C.byref(g))
    for _ in range(5):
        L.kr_graph_launch(g.value, s)
    L.kr_stream_synchronize(s)
    assert int(step.item()) == 5 and ctx.cpu().tolist() == [5, 5]
    np.testing.assert_array_equal(hist.cpu().numpy()[:5], [[7, 9]] * 5)
    L.kr_graph_destroy(g.value)


def test_gemm_with_a_long_k_tail_inside_a_stream_capture(L):
    """The split-K scratch of a GEMM tail is the CALLER's (kr_gemm_bf16_ws; ADVICE r2: a scratch the library allocated on
    first use made the accumulation order depend on whether the stream's first GEMM ran inside a capture).  The library
    allocates nothing: with the scratch a captured launch gives the eager launch's bits EXACTLY, on a fresh stream too;
    without it the tail runs unsplit (another summation order, same tolerance), captured or not."""
    import os
    stream = torch.cuda.Stream()          # a fresh stream
    s = stream.cuda_stream
    M, N, K = 11152, 1536, 4480           # 264 tiles = 1 round + 8; K >= 4096: with a scratch the tail is split along K
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert 0 < (-(-M // 256) * (N // 256)) % cus <= cus // 2
    rng = np.random.default_rng(11)
    A, W = rnd(rng, M, K), rnd(rng, N, K, scale=K ** -0.5)
    Ad, Wd = dev_bf16(A), dev_bf16(W)
    scratch = torch.zeros(512 * 65536 // 4, dtype=torch.float32, device=DEV)

    def gemm(ws, stream_):
        Cd = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=DEV)
        torch.cuda.synchronize()
        if ws:
            L.kr_gemm_bf16_ws(ptr(Ad), K, ptr(Wd), 0, 0, 0, ptr(Cd), N, M, N, K, EPI_NONE, 0, ptr(scratch), scratch.numel() * 4, stream_)
        else:
            L.kr_gemm_bf16(ptr(Ad), K, ptr(Wd), 0, 0, 0, ptr(Cd), N, M, N, K, EPI_NONE, 0, stream_)
        return Cd

    os.environ["KARANTA_GEMM_TILE"] = "512"
    try:
        out = {}
        for ws in (True, False):
            Cd = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=DEV)
            torch.cuda.synchronize()
            L.kr_graph_begin_capture(s)
            if ws:
                L.kr_gemm_bf16_ws(ptr(Ad), K, ptr(Wd), 0, 0, 0, ptr(Cd), N, M, N, K, EPI_NONE, 0, ptr(scratch), scratch.numel() * 4, s)
            else:
                L.kr_gemm_bf16(ptr(Ad), K, ptr(Wd), 0, 0, 0, ptr(Cd), N, M, N, K, EPI_NONE, 0, s)
            g = C.c_void_p()
            L.kr_graph_end_capture(s, C.byref(g))
            L.kr_graph_launch(g.value, s)
            L.kr_stream_synchronize(s)
            out["cap", ws] = host(Cd)
            L.kr_graph_destroy(g.value)
            Ce = gemm(ws, 0)
            torch.cuda.synchronize()
            out["eager", ws] = host(Ce)
    finally:
        os.environ.pop("KARANTA_GEMM_TILE", None)
    np.testing.assert_array_equal(out["cap", True], out["eager", True])      # same call, same bits: captured or eager
    np.testing.assert_array_equal(out["cap", False], out["eager", False])
    assert not np.array_equal(out["eager", True], out["eager", False]), "the scratch should change the tail's K order"
    rows = np.r_[0:200, M - 400:M]
    for k in (True, False):
        assert_close_bf16(out["eager", k][rows], ref_linear(A[rows], W), what=f"GEMM tail (scratch={k})")
    with pytest.raises(Exception):      # a scratch smaller than KR_GEMM_SCRATCH_BYTES is refused
        L.kr_gemm_bf16_ws(ptr(Ad), K, ptr(Wd), 0, 0, 0, ptr(Ad), N, M, N, K, EPI_NONE, 0, ptr(scratch), 1 << 20, 0)


def test_events_time_a_kernel(L):
    stream = torch.cuda.Stream()
    s = stream.cuda_stream
    e0, e1 = C.c_void_p(), C.c_void_p()
    L.kr_event_create(C.byref(e0)); L.kr_event_create(C.byref(e1))
    x = torch.zeros(4096, 1536, dtype=torch.bfloat16, device=DEV)
    w = torch.ones(1536, dtype=torch.bfloat16, device=DEV)
    y = torch.empty_like(x)
    torch.cuda.synchronize()
    L.kr_event_record(e0, s)
    L.kr_rmsnorm(ptr(x), 1536, ptr(w), ptr(y), 4096, 1536, 1e-6, s)
    L.kr_event_record(e1, s)
    L.kr_event_synchronize(e1)
    ms = C.c_float()
    L.kr_event_elapsed_ms(e0, e1, C.byref(ms))
    assert 0 < ms.value < 50
    L.kr_event_destroy(e0); L.kr_event_destroy(e1)


def test_rccl_single_rank_broadcast(L):
    """kr_comm_* + kr_bcast_weights with a 1-rank communicator (the only multi-GPU primitive on the path)."""
    uid = (C.c_uint8 * 128)()
    L.kr_comm_unique_id(uid)
    comm = C.c_void_p()
    L.kr_comm_init(C.byref(comm), 1, 0, uid)
    buf = torch.arange(1 << 20, dtype=torch.int32, device=DEV)
    L.kr_bcast_weights(comm, ptr(buf), buf.numel() * 4, 0, 0)
    torch.cuda.synchronize()
    assert int(buf[12345].item()) == 12345
    L.kr_comm_destroy(comm)


def test_rccl_single_rank_broadcast_of_an_arena_above_2_gib_on_the_engine_stream(L):
    """First-contact insurance for the 8-GPU run: the 7B arena is 16.6 GB, so kr_bcast_weights walks
    1 GiB pieces with 64-bit offsets — here a 2.5 GiB buffer (three pieces, the last one partial) on a NON-default stream,
    behind a fill that is still executing on that stream (the broadcast must be stream-ordered, as the engine's
    load -> broadcast -> first launch sequence needs), with a 1-rank communicator; bytes around every piece boundary and the
    tail are checked."""
    n = (5 << 29) + 4096 + 13                      # 2.5 GiB + a ragged tail
    free, _ = torch.cuda.mem_get_info()
    if free < n + (1 << 30):
        pytest.skip("not enough free HBM for a 2.5 GiB buffer")
    stream = torch.cuda.Stream()
    uid = (C.c_uint8 * 128)()
    L.kr_comm_unique_id(uid)
    comm = C.c_void_p()
    L.kr_comm_init(C.byref(comm), 1, 0, uid)
    cnt = C.c_int(0)
    L.kr_comm_count(comm, C.byref(cnt))
    assert cnt.value == 1
    buf = torch.empty(n, dtype=torch.uint8, device=DEV)
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        buf.fill_(7)
        buf[::4093] = 201                         # (4093 is prime: a pattern that does not line up with any piece size)
        L.kr_bcast_weights(comm, ptr(buf), n, 0, stream.cuda_stream)
        probe = []
        for edge in (0, 1 << 30, 2 << 30, n):
            lo, hi = max(0, edge - 4096), min(n, edge + 4096)
            probe.append((lo, buf[lo:hi].clone()))
        total = int((buf == 201).sum().item())
    stream.synchronize()
    assert total == -(-n // 4093)
    for lo, got in probe:
        idx = np.arange(lo, lo + got.numel())
        want = np.where(idx % 4093 == 0, 201, 7).astype(np.uint8)
        np.testing.assert_array_equal(got.cpu().numpy(), want)
    L.kr_comm_destroy(comm)
    del buf
    torch.cuda.empty_cache()


# ----------------------------------------------------------------------------- guided decoding + log-probabilities
def _random_vocab(rng, V, eos):
    """Token byte strings: all 256 single bytes first (as byte-level BPE has), then random 2..6 byte pieces over a small
    alphabet so that many of them survive a pattern; a few specials (b"")."""
    alphabet = b"abcdef0123456789-;: \n{}\",ntrue"
    voc = [bytes([i]) for i in range(256)]
    while len(voc) < V:
        voc.append(bytes(rng.choice(list(alphabet)) for _ in range(int(rng.integers(2, 7)))))
    for i in rng.integers(256, V, 20):
        voc[int(i)] = b""
    for e in eos:
        voc[e] = b""
    return voc


@pytest.mark.parametrize("pattern", [r"[a-f]{3}-[0-9]{2}(?:;[a-z]+)?", r"\{\"n\": (?:true|null|-?[0-9]+)\}",
                                     r"---\nab: (?:[a-z]{2}|null)\n(?:---|---\n[\s\S]+)"])
def test_guide_masks_and_advance_match_oracle(L, pattern):
    """kr_guide_build_masks against the oracle's token-by-token rule for EVERY state of the DFA (bit-exact), and
    kr_guide_advance against guide_walk."""
    from karanta_ocr_amd import guided as G
    rng = np.random.default_rng(3)
    V, eos = 3001, [2999, 3000]
    voc = _random_vocab(rng, V, eos)
    g = G.compile_regex(pattern)
    S, mw = g.n_states, 2 * ((V + 63) // 64)
    off, flat = G.pack_vocab(voc)
    d_off, d_flat = torch.from_numpy(off).to(DEV), torch.from_numpy(flat).to(DEV)
    d_trans = torch.from_numpy(g.trans.view(np.int16)).to(DEV)
    d_acc = torch.from_numpy(g.accept.astype(np.uint8)).to(DEV)
    d_eos = torch.tensor(eos, dtype=torch.int32, device=DEV)
    masks = torch.full((S, mw), -1, dtype=torch.int32, device=DEV)
    L.kr_guide_build_masks(ptr(d_trans), ptr(d_acc), S, ptr(d_off), ptr(d_flat), V, ptr(d_eos), len(eos), ptr(masks), mw, 0)
    torch.cuda.synchronize()
    bits = np.unpackbits(masks.cpu().numpy().view(np.uint8), axis=1, bitorder="little")
    assert not bits[:, V:].any(), "bits beyond the vocabulary must be clear"
    for s in range(S):
        want = O.guide_token_mask(g.trans, g.accept, s, voc, eos)
        np.testing.assert_array_equal(bits[s, :V].astype(bool), want, err_msg=f"state {s}")
    assert bits[1:, :V].any(axis=1).all(), "every live state allows at least one token"
    # advance: rows 0..5 guided from various states, row 6 unconstrained, row 7 finished
    B = 8
    states = rng.integers(1, S, B).astype(np.int32)
    toks = np.zeros(B, np.int32)
    for b in range(B):
        ok = np.flatnonzero(O.guide_token_mask(g.trans, g.accept, int(states[b]), voc, []))
        toks[b] = int(rng.choice(ok)) if ok.size else 0
    fin = np.zeros(B, np.int32); fin[7] = 1
    gt = np.full(B, d_trans.data_ptr(), np.int64); gt[6] = 0
    d_st = torch.from_numpy(states.copy()).to(DEV)
    d_tok, d_fin, d_gt = torch.from_numpy(toks).to(DEV), torch.from_numpy(fin).to(DEV), torch.from_numpy(gt).to(DEV)
    L.kr_guide_advance(ptr(d_tok), ptr(d_fin), ptr(d_gt), ptr(d_st), ptr(d_off), ptr(d_flat), V, B, 0)
    torch.cuda.synchronize()
    got = d_st.cpu().numpy()
    for b in range(B):
        want = states[b] if b >= 6 else O.guide_walk(g.trans, int(states[b]), voc[int(toks[b])])
        assert got[b] == want, (b, got[b], want)       # (a final state without continuation walks byte 0 to the dead state)
    with pytest.raises(KarantaHipError):
        L.kr_guide_build_masks(ptr(d_trans), ptr(d_acc), S, ptr(d_off), ptr(d_flat), V, ptr(d_eos), len(eos), ptr(masks), mw - 1, 0)


def test_gumbel_argmax_guided_masks_rows(L):
    """Masked rows pick the best ALLOWED token (greedy and sampled rows), unmasked rows are those of kr_gumbel_argmax,
    an all-clear mask row yields the fallback token."""
    rng = np.random.default_rng(23)
    B, V, n_part = 5, 5003, 7
    mw = 2 * ((V + 63) // 64)
    logits = rng.standard_normal((B, V)).astype(np.float32) * 3
    temps = np.asarray([0.0, 0.9, 0.0, 0.0, 0.7], np.float32)
    seeds = np.asarray([1, 2, 3, 4, 5], np.uint32)
    plen = np.asarray([4, 4, 4, 4, 4], np.int32); ctx = np.asarray([6, 9, 4, 5, 11], np.int32)
    S = 3
    allow = rng.random((S, V)) < 0.02
    allow[2] = False                                              # a row that allows nothing
    words = np.packbits(np.pad(allow, ((0, 0), (0, mw * 32 - V))), axis=1, bitorder="little").view(np.int32)
    d_masks = torch.from_numpy(words.copy()).to(DEV)
    gm = np.asarray([d_masks.data_ptr(), d_masks.data_ptr(), 0, d_masks.data_ptr(), 0], np.int64)
    gs = np.asarray([0, 1, 0, 2, 1], np.int32)
    t = lambda a: torch.from_numpy(a).to(DEV)
    ld, td, sd, pd, cd = t(logits), t(temps), t(seeds.view(np.int32)), t(plen), t(ctx)
    av = torch.zeros(B, n_part, device=DEV); ai = torch.zeros(B, n_part, dtype=torch.int32, device=DEV)
    d_gm, d_gs = t(gm), t(gs)
    L.kr_gumbel_argmax_guided(ptr(ld), V, V, ptr(td), ptr(sd), ptr(cd), ptr(pd), ptr(av), ptr(ai), n_part, B, ptr(d_gm), ptr(d_gs),
                              mw, 77, 0)
    av0 = torch.zeros(B, n_part, device=DEV); ai0 = torch.zeros(B, n_part, dtype=torch.int32, device=DEV)
    L.kr_gumbel_argmax(ptr(ld), V, V, ptr(td), ptr(sd), ptr(cd), ptr(pd), ptr(av0), ptr(ai0), n_part, B, 0)
    torch.cuda.synchronize()
    avh, aih = av.cpu().numpy(), ai.cpu().numpy()
    pick = lambda b: int(aih[b][np.lexsort((aih[b], -avh[b]))[0]])     # best value, ties to the lowest id (the sampler's rule)
    for b, st in ((0, 0), (1, 1)):
        sc = O.sample_scores(logits[b], float(temps[b]), int(seeds[b]), int(ctx[b] + 1 - plen[b]))
        sc = np.where(allow[st], sc, -np.inf)
        top2 = np.sort(sc)[-2:]
        assert allow[st][pick(b)]
        if top2[1] - top2[0] > 1e-4:
            assert pick(b) == int(np.argmax(sc))
    for b in (2, 4):
        np.testing.assert_array_equal(aih[b], ai0.cpu().numpy()[b])
        np.testing.assert_array_equal(avh[b], av0.cpu().numpy()[b])
    assert pick(3) == 77 and np.isneginf(avh[3]).all()
    with pytest.raises(KarantaHipError):
        L.kr_gumbel_argmax_guided(ptr(ld), V, V, ptr(td), ptr(sd), ptr(cd), ptr(pd), ptr(av), ptr(ai), n_part, B, ptr(d_gm), ptr(d_gs),
                                  mw, V, 0)


@pytest.mark.parametrize("V,n_part,k", [(151936, 64, 5), (5003, 2, 20), (512, 1, 0), (300, 1, 20)])
def test_logprobs_topk_matches_oracle(L, V, n_part, k):
    """log-softmax of the chosen token and the k most probable tokens (ids exact, ties to the lowest id; values within
    fp32 rounding of the float64 oracle), written at the history index ctx - prompt; finished rows record nothing."""
    rng = np.random.default_rng(V + k)
    B, HB, H, KS = 4, 6, 5, 20
    logits = (rng.standard_normal((B, V)) * 4).astype(np.float32)
    logits[1, 7] = logits[1, 3] = logits[1].max() + 1.0            # an exact tie at the top
    logits[2] = np.round(logits[2])                                 # many ties further down
    toks = rng.integers(0, V, B).astype(np.int32)
    plen = np.asarray([3, 5, 2, 9], np.int32); ctx = np.asarray([3, 8, 6, 10], np.int32)   # history rows 0, 3, 4, (1)
    fin = np.asarray([0, 0, 0, 1], np.int32)
    t = lambda a: torch.from_numpy(a).to(DEV)
    ld = t(logits)
    pv = torch.zeros(B, n_part, max(k, 1), device=DEV); pi = torch.zeros(B, n_part, max(k, 1), dtype=torch.int32, device=DEV)
    ms = torch.zeros(B, n_part, 2, device=DEV)
    out = torch.full((H, HB, 1 + KS), 7.0, device=DEV); oi = torch.full((H, HB, KS), -3, dtype=torch.int32, device=DEV)
    slot0 = 1                                                       # rows go to history columns 1..4 of 6
    d_tok, d_ctx, d_plen, d_fin = t(toks), t(ctx), t(plen), t(fin)
    L.kr_logprobs_topk(ptr(ld), V, V, k, n_part, ptr(pv), ptr(pi), ptr(ms), ptr(d_tok), ptr(d_ctx), ptr(d_plen), ptr(d_fin),
                       ptr(out[:, slot0:]), ptr(oi[:, slot0:]), H, HB, KS, B, 0)
    torch.cuda.synchronize()
    oh, ih = out.cpu().numpy(), oi.cpu().numpy()
    touched = np.zeros((H, HB), bool)
    for b in range(3):
        h = int(ctx[b] - plen[b])
        touched[h, slot0 + b] = True
        lp = O.log_softmax(logits[b])
        np.testing.assert_allclose(oh[h, slot0 + b, 0], lp[toks[b]], rtol=1e-5, atol=5e-5)
        ids, vals = O.top_logprobs(logits[b], k)
        np.testing.assert_array_equal(ih[h, slot0 + b, :k], ids)
        np.testing.assert_allclose(oh[h, slot0 + b, 1:1 + k], vals, rtol=1e-5, atol=5e-5)
        assert (oh[h, slot0 + b, 1 + k:] == 7.0).all() and (ih[h, slot0 + b, k:] == -3).all()
    assert (oh[~touched] == 7.0).all() and (ih[~touched] == -3).all(), "finished rows and other history rows stay untouched"
    with pytest.raises(KarantaHipError):
        L.kr_logprobs_topk(ptr(ld), V, V, 21, n_part, ptr(pv), ptr(pi), ptr(ms), ptr(d_tok), ptr(d_ctx), ptr(d_plen), ptr(d_fin),
                           ptr(out), ptr(oi), H, HB, KS, B, 0)
    if V > 4096 * 2:
        with pytest.raises(KarantaHipError):
            L.kr_logprobs_topk(ptr(ld), V, V, k, 2, ptr(pv), ptr(pi), ptr(ms), ptr(d_tok), ptr(d_ctx), ptr(d_plen), ptr(d_fin),
                               ptr(out), ptr(oi), H, HB, KS, B, 0)


# ----------------------------------------------------------------------------- fast-residual mode kernels
@pytest.mark.parametrize("H,d,n_split,M", [(12, 1536, 8, 8), (28, 3584, 8, 4), (2, 256, 4, 3), (12, 1536, 8, 21), (16, 2048, 16, 16),
                                            (3, 384, 1, 5)])
def test_oproj_heads_merges_and_accumulates(L, H, d, n_split, M):
    """kr_oproj_heads = attn_merge_kernel + o_proj + residual add, with K split by head and float atomics: against the
    merge formula in numpy (merged head rounded to bf16, as the separate merge launch leaves it) and an f64 matmul."""
    if not L.experiments:
        pytest.skip("experiment entry point: -DKR_EXPERIMENTS builds only (include/karanta_hip_experiments.h)")
    rng = np.random.default_rng(H * 1000 + d + n_split + M)
    hd = 128
    ws = np.zeros((M, H, n_split, hd + 4), np.float32)
    ws[..., :hd] = rng.standard_normal((M, H, n_split, hd)).astype(np.float32) * 3
    ws[..., hd] = rng.uniform(-20, 5, (M, H, n_split)).astype(np.float32)          # running max (log2 domain)
    ws[..., hd + 1] = rng.uniform(0.5, 40, (M, H, n_split)).astype(np.float32)     # running sum
    if n_split > 1:
        ws[0, 0, 1, hd + 1] = 0.0; ws[0, 0, 1, :hd] = 0.0; ws[0, 0, 1, hd] = -1e30   # a split without keys
    Wo = rnd(rng, d, H * hd, scale=(H * hd) ** -0.5)
    x0 = rng.standard_normal((M, d)).astype(np.float32)
    mm = ws[..., hd].max(-1, keepdims=True)
    sc = np.exp2(ws[..., hd] - mm)
    merged = (ws[..., :hd] * sc[..., None]).sum(2) / (ws[..., hd + 1] * sc).sum(-1)[..., None]
    merged = bf16_round(merged.astype(np.float32)).reshape(M, H * hd)
    ref = x0.astype(np.float64) + merged.astype(np.float64) @ Wo.astype(np.float64).T
    wsd, Wd = torch.from_numpy(ws).to(DEV), dev_bf16(pack_w16x64(Wo))
    pad = 24
    acc = torch.zeros(M, d + pad, dtype=torch.float32, device=DEV)
    acc[:, :d] = torch.from_numpy(x0).to(DEV)
    for rep in range(2):                                   # twice: the kernel ADDS (2 x product), nothing else is touched
        L.kr_oproj_heads(ptr(wsd), n_split, ptr(Wd), 0, ptr(acc), d + pad, M, d, H, 0)
    torch.cuda.synchronize()
    got = acc.cpu().numpy()
    assert not got[:, d:].any()
    want = ref + (ref - x0)
    np.testing.assert_allclose(got[:, :d], want, rtol=2e-5, atol=2e-4 * np.abs(want).max())


def test_oproj_heads_fp8_weights(L):
    if not L.experiments:
        pytest.skip("experiment entry point: -DKR_EXPERIMENTS builds only (include/karanta_hip_experiments.h)")
    rng = np.random.default_rng(5)
    H, d, n_split, M, hd = 12, 1536, 8, 8, 128
    from karanta_ocr_amd.weights import fp8_e4m3_to_f32, pack_w16x64_fp8, quantize_fp8_rows
    ws = np.zeros((M, H, n_split, hd + 4), np.float32)
    ws[..., :hd] = rng.standard_normal((M, H, n_split, hd)).astype(np.float32)
    ws[..., hd] = rng.uniform(-3, 3, (M, H, n_split)).astype(np.float32)
    ws[..., hd + 1] = rng.uniform(0.5, 4, (M, H, n_split)).astype(np.float32)
    q, sc8 = quantize_fp8_rows(rnd(rng, d, H * hd, scale=(H * hd) ** -0.5))
    Wf = fp8_e4m3_to_f32(q) * sc8[:, None]
    mm = ws[..., hd].max(-1, keepdims=True)
    sc = np.exp2(ws[..., hd] - mm)
    merged = bf16_round(((ws[..., :hd] * sc[..., None]).sum(2) / (ws[..., hd + 1] * sc).sum(-1)[..., None]).astype(np.float32))
    ref = merged.reshape(M, H * hd).astype(np.float64) @ Wf.astype(np.float64).T
    acc = torch.zeros(M, d, dtype=torch.float32, device=DEV)
    qd = torch.from_numpy(np.ascontiguousarray(pack_w16x64_fp8(q))).to(DEV)
    sd = torch.from_numpy(sc8.astype(np.float32)).to(DEV)
    L.kr_oproj_heads(ptr(torch.from_numpy(ws).to(DEV)), n_split, ptr(qd), ptr(sd), ptr(acc), d, M, d, H, 0)
    torch.cuda.synchronize()
    np.testing.assert_allclose(acc.cpu().numpy(), ref, rtol=2e-5, atol=2e-4 * np.abs(ref).max())


@pytest.mark.parametrize("M,K,blocks,waves", [(8, 1536, 256, 5), (16, 3584, 9, 8), (3, 512, 2, 3), (21, 1536, 256, 8)])
def test_linear_wide_f32_rows_equal_the_rounded_bf16_rows(L, M, K, blocks, waves):
    """kr_linear_decode_wide_x32 (x rows from the f32 residual accumulator) = kr_linear_decode_wide on their bf16 rounding,
    bit for bit; workgroup 0 leaves the rounded rows in x_out."""
    if not L.experiments:
        pytest.skip("experiment entry point: -DKR_EXPERIMENTS builds only (include/karanta_hip_experiments.h)")
    rng = np.random.default_rng(M + K)
    ff = 16 * 37
    xf = (rng.standard_normal((M, K)) * 2).astype(np.float32)
    W = rnd(rng, 2 * ff, K, scale=K ** -0.5)
    nw = bf16_round(1 + 0.1 * rng.standard_normal(K).astype(np.float32))
    Wd, nwd = dev_bf16(pack_w16x64(W)), dev_bf16(nw)
    xfd = torch.from_numpy(np.concatenate([xf, np.zeros((M, 8), np.float32)], 1)).to(DEV)      # ldx = K + 8 floats
    xbd = dev_bf16(bf16_round(xf))
    out_a = torch.zeros(M, ff, dtype=torch.bfloat16, device=DEV); out_b = torch.zeros_like(out_a)
    x_out = torch.zeros(M, K + 16, dtype=torch.bfloat16, device=DEV)
    av = torch.zeros(M, blocks * waves, device=DEV); ai = torch.zeros(M, blocks * waves, dtype=torch.int32, device=DEV)
    L.kr_linear_decode_wide_x32(DEC_SILU8, ptr(xfd), K + 8, ptr(x_out), K + 16, ptr(Wd), 0, ptr(nwd), 1e-6, ptr(out_a), 0, ff, M,
                                2 * ff, K, blocks, waves, ptr(av), ptr(ai), 0)
    L.kr_linear_decode_wide(DEC_SILU8, ptr(xbd), K, ptr(Wd), 0, ptr(nwd), 1e-6, 0, 0, ptr(out_b), 0, ff, M, 2 * ff, K, blocks, waves,
                            ptr(av), ptr(ai), 0)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(host(out_a), host(out_b))
    np.testing.assert_array_equal(host(x_out)[:, :K], bf16_round(xf))
    assert not host(x_out)[:, K:].any()


@pytest.mark.parametrize("M,K,parts", [(8, 1536, True), (8, 1536, False), (21, 2048, True), (16, 3584, True)])
def test_linear_narrow_x32_also_leaves_x_new_as_f32(L, M, K, parts):
    """kr_linear_decode_narrow_x32: the same product as kr_linear_decode_narrow, and workgroup 0 stores x_new (the bf16
    residual row it normalises) as f32 too — the start value of the fast-residual accumulator — inside ldxf only."""
    if not L.experiments:
        pytest.skip("experiment entry point: -DKR_EXPERIMENTS builds only (include/karanta_hip_experiments.h)")
    rng = np.random.default_rng(900 + M + K)
    N = 16 * 9
    x, W, nw = rnd(rng, M, K, scale=2.0), rnd(rng, N, K, scale=K ** -0.5), bf16_round(1 + 0.1 * rnd(rng, K))
    p = (rng.standard_normal((2, M, K)) * 0.5).astype(np.float32)
    xd, Wd, nd, pd = dev_bf16(x), dev_bf16(pack_w16x64(W)), dev_bf16(nw), torch.from_numpy(p).to(DEV)
    xo = torch.full((M, K), 5.0, dtype=torch.bfloat16, device=DEV)
    xf = torch.full((M, K + 4), 7.0, dtype=torch.float32, device=DEV)
    out_a = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV); out_b = torch.zeros_like(out_a)
    L.kr_linear_decode_narrow_x32(DEC_PLAIN, ptr(xd), K, ptr(pd) if parts else 0, 2 if parts else 0, ptr(xo) if parts else 0,
                                  K if parts else 0, ptr(xf), K + 4, ptr(Wd), 0, 0, ptr(nd), 1e-6, 0, 0, ptr(out_a), 0, N, M, N, K, 8, 1,
                                  0, 0, 0, 0, 0, 0, 0, 0, 0, 64, None, 0, 0, 0, 0)   # (no opts, no prefetch workgroups)
    narrow_call(L, DEC_PLAIN, ptr(xd), ptr(Wd), M, N, K, out=ptr(out_b), ldc=N, norm_w=ptr(nd), part_in=ptr(pd) if parts else 0,
                x_out=ptr(xo) if parts else 0)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(host(out_a), host(out_b))
    got = xf.cpu().numpy()
    np.testing.assert_array_equal(got[:, :K], host(xo) if parts else x)
    assert (got[:, K:] == 7.0).all()
