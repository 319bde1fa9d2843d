"""The decode linears of 17..32-row batches (kr_linear_decode32 and the packed-activation producers) against the <= 16-row
kernels they must agree with BIT FOR BIT: a page's tokens do not depend on the size of the batch it decodes in
(the reference's callers fill the server's slots one request at a time: bulk_processing/workers/inference_worker.py:331-339).
Through the C-ABI, on a real MI355X."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from karanta_ocr_amd import weights as WT  # noqa: E402
from karanta_ocr_amd._lib import (DEC_OUT_XP, DEC_PLAIN, DEC_ROPE_KV, DEC_SILU8, Dec32, KarantaHipError, lib, narrow_opts,  # noqa: E402
                                  ptr)
from karanta_ocr_amd.weights import bf16_round, pack_rows32, pack_w16x64, unpack_rows32  # noqa: E402

DEV = "cuda:0"


@pytest.fixture(scope="module")
def L():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return lib()


def dev_bf16(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV).to(torch.bfloat16).contiguous()


def host(t):
    torch.cuda.synchronize()
    return t.float().cpu().numpy()


def rnd(rng, *shape, scale=1.0):
    return bf16_round(rng.standard_normal(shape).astype(np.float32) * np.float32(scale))


def ints(rng, *shape, lo=-1, hi=2):
    return rng.integers(lo, hi, size=shape).astype(np.float32)


def dec32(L, mode, xp, W, M, N, K, waves_ref, ksplit=1, out=None, out_f32=None, ldc=0, bias=None, res=None, ldr=0, w_scale=None,
          atomic=False, tiles=0, zero=None, zero_bytes=0, cs=None, cs_stride=0, plen=None, ctx=None, q=None, kc=None, vc=None,
          heads=0, kv_heads=0, s_max=64, gs=False):
    a = Dec32(ptr(xp), ptr(W), ptr(w_scale), ptr(bias), ptr(res), ldr, ptr(out), ptr(out_f32), ldc, M, N, K, waves_ref, ksplit,
              1 if atomic else 0, tiles, 1 if gs else 0, 0, ptr(zero), zero_bytes, ptr(cs), cs_stride, ptr(plen), ptr(ctx), ptr(q), ptr(kc), ptr(vc),
              heads, kv_heads, s_max)
    L.kr_linear_decode32(mode, C.byref(a), 0)


def narrow(L, mode, x, W, M, N, K, out=0, out_f32=0, ldc=0, bias=0, res=0, ldr=0, waves=8, ksplit=1, cs=0, cs_stride=0, plen=0,
           ctx=0, q_out=0, kc=0, vc=0, heads=0, kv_heads=0, s_max=64, opts=None, w_scale=None):
    if w_scale is not None:
        L.kr_linear_decode_narrow_fp8(mode, x, K, 0, 0, 0, K, W, ptr(w_scale), bias, 0, 1e-6, res, ldr, out, out_f32, ldc, M, N, K, waves,
                                      ksplit, cs, cs_stride, plen, ctx, q_out, kc, vc, heads, kv_heads, s_max, opts, 0)
    else:
        L.kr_linear_decode_narrow(mode, x, K, 0, 0, 0, K, W, bias, 0, 1e-6, res, ldr, out, out_f32, ldc, M, N, K, waves, ksplit, cs,
                                  cs_stride, plen, ctx, q_out, kc, vc, heads, kv_heads, s_max, opts, 0)


@pytest.mark.parametrize("M,K", [(1, 64), (17, 192), (32, 1536), (21, 3584)])
def test_pack_rows32_is_the_documented_layout(L, M, K):
    rng = np.random.default_rng(M + K)
    x = rnd(rng, M, K)
    xd = dev_bf16(np.pad(x, ((0, 0), (0, 8))))            # a row stride larger than K
    xp = torch.full((32 * K,), 7.0, dtype=torch.bfloat16, device=DEV)
    L.kr_pack_rows32(ptr(xd), K + 8, M, K, ptr(xp), 0)
    np.testing.assert_array_equal(host(xp), pack_rows32(x))
    np.testing.assert_array_equal(unpack_rows32(host(xp), K)[:M], x)
    with pytest.raises(KarantaHipError):
        L.kr_pack_rows32(ptr(xd), K + 8, 33, K, ptr(xp), 0)


# (N, K, waves of the <= 16-row launch, ksplit): the shapes of the decoder's narrow linears at the 2B and 7B widths
SHAPES = [(1536, 1536, 8, 1),      # o_proj 2B
          (96, 8960, 16, 2),       # down_proj 2B (a slice of its 1536 rows): 2 x 16 atoms of 4-5 chunks, two per wave here
          (96, 8960, 16, 1),       # the last layer's unsplit down_proj
          (224, 3584, 8, 1),       # o_proj 7B
          (448, 18944, 8, 2),      # down_proj 7B: two tiles per workgroup at >= 192 tile pairs
          (48, 256, 8, 1), (16, 64, 8, 1), (80, 64 * 19, 16, 1), (32, 64 * 5, 16, 2)]   # ragged: waves without chunks


@pytest.mark.parametrize("M", [17, 23, 32])
@pytest.mark.parametrize("N,K,waves,ksplit", SHAPES)
@pytest.mark.parametrize("tiles", [0, 1, 2])
def test_linear_decode32_gives_the_narrow_launch_bits(L, M, N, K, waves, ksplit, tiles):
    """Row for row the bits of kr_linear_decode_narrow (run on each 16-row range of the batch, the launch batches of up to
    16 rows use): PLAIN with bias + residual, deferred split-K into two slabs and into one atomically accumulated slab.
    Random operands (any change of the summation order shows in the low bits) and small integers (exact)."""
    rng = np.random.default_rng(M + N + K + waves + ksplit)
    for exact in (True, False):
        x = ints(rng, M, K) if exact else rnd(rng, M, K, scale=1.5)
        W = ints(rng, N, K) if exact else rnd(rng, N, K, scale=K ** -0.5)
        bias, res = rnd(rng, N, scale=0.2), rnd(rng, M, N)
        xd, Wd, bd, rd = dev_bf16(x), dev_bf16(pack_w16x64(W)), dev_bf16(bias), dev_bf16(res)
        xp = dev_bf16(pack_rows32(x))
        ranges = [(0, 16), (16, M - 16)]
        if ksplit == 1:
            ref = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
            for r0, m in ranges:
                narrow(L, DEC_PLAIN, ptr(xd[r0:]), ptr(Wd), m, N, K, out=ptr(ref[r0:]), ldc=N, bias=ptr(bd), res=ptr(rd[r0:]), ldr=N, waves=waves)
            got = torch.full((M, N), 3.0, dtype=torch.bfloat16, device=DEV)
            dec32(L, DEC_PLAIN, xp, Wd, M, N, K, waves, out=got, ldc=N, bias=bd, res=rd, ldr=N, tiles=tiles)
            np.testing.assert_array_equal(host(got), host(ref))
            if exact:
                np.testing.assert_array_equal(host(got), bf16_round(x @ W.T + bias + res))
            gf = torch.zeros(M, N, dtype=torch.float32, device=DEV)
            dec32(L, DEC_PLAIN, xp, Wd, M, N, K, waves, out_f32=gf, ldc=N, tiles=tiles)           # f32 output, no epilogue operands
            rf = torch.zeros(M, N, dtype=torch.float32, device=DEV)
            for r0, m in ranges:
                narrow(L, DEC_PLAIN, ptr(xd[r0:]), ptr(Wd), m, N, K, out_f32=ptr(rf[r0:]), ldc=N, waves=waves)
            np.testing.assert_array_equal(gf.cpu().numpy(), rf.cpu().numpy())
        else:
            ref = torch.zeros(ksplit, M, N, dtype=torch.float32, device=DEV)
            for r0, m in ranges:
                part = torch.zeros(ksplit, m, N, dtype=torch.float32, device=DEV)
                narrow(L, DEC_PLAIN, ptr(xd[r0:]), ptr(Wd), m, N, K, out_f32=ptr(part), ldc=N, waves=waves, ksplit=ksplit)
                ref[:, r0:r0 + m] = part
            got = torch.full((ksplit, M, N), 3.0, dtype=torch.float32, device=DEV)
            dec32(L, DEC_PLAIN, xp, Wd, M, N, K, waves, ksplit=ksplit, out_f32=got, ldc=N, tiles=tiles)
            np.testing.assert_array_equal(got.cpu().numpy(), ref.cpu().numpy())
            acc = torch.zeros(M, N, dtype=torch.float32, device=DEV)
            other = torch.full((M * N + 8,), 3.0, dtype=torch.float32, device=DEV)
            dec32(L, DEC_PLAIN, xp, Wd, M, N, K, waves, ksplit=ksplit, out_f32=acc, ldc=N, atomic=True, tiles=tiles, zero=other,
                  zero_bytes=M * N * 4)
            r = ref.cpu().numpy()
            np.testing.assert_array_equal(acc.cpu().numpy(), r[0] + r[1])
            oc = other.cpu().numpy()
            assert not oc[:M * N].any() and (oc[M * N:] == 3.0).all(), "zeroing job: exactly the requested range"
            # group split: each K range's atoms in two workgroups adding into the range's slab = the narrow launch's slab of
            # that range (its fold is (first half of the waves) + (second half))
            if ksplit == 2 and tiles == 0:
                for tw in (0, 2, 4):                      # 0: the launch's own choice (4 at 8-atom partitions, else 2)
                    if (N // 16) % (tw or (4 if waves == 8 else 2)):
                        continue
                    pair = torch.zeros(2, M, N, dtype=torch.float32, device=DEV)
                    dec32(L, DEC_PLAIN, xp, Wd, M, N, K, waves, ksplit=2, out_f32=pair, ldc=N, atomic=True, gs=True, tiles=tw)
                    np.testing.assert_array_equal(pair.cpu().numpy(), r)


@pytest.mark.parametrize("M", [19, 32])
@pytest.mark.parametrize("N,K,waves,ksplit", [(1536, 1536, 8, 1), (96, 8960, 16, 2), (448, 18944, 8, 2), (48, 256, 8, 1)])
def test_linear_decode32_fp8_gives_the_narrow_fp8_launch_bits(L, M, N, K, waves, ksplit):
    rng = np.random.default_rng(M + N + K)
    x, W = rnd(rng, M, K, scale=1.5), rnd(rng, N, K, scale=K ** -0.5)
    codes, scale = WT.quantize_fp8_rows(W)
    Wd = torch.from_numpy(WT.pack_w16x64_fp8(codes)).to(DEV)
    sd = torch.from_numpy(scale).to(DEV)
    xd, xp = dev_bf16(x), dev_bf16(pack_rows32(x))
    res = rnd(rng, M, N)
    rd = dev_bf16(res)
    ranges = [(0, 16), (16, M - 16)]
    if ksplit == 1:
        ref = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
        for r0, m in ranges:
            narrow(L, DEC_PLAIN, ptr(xd[r0:]), ptr(Wd), m, N, K, out=ptr(ref[r0:]), ldc=N, res=ptr(rd[r0:]), ldr=N, waves=waves, w_scale=sd)
        got = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
        dec32(L, DEC_PLAIN, xp, Wd, M, N, K, waves, out=got, ldc=N, res=rd, ldr=N, w_scale=sd)
        np.testing.assert_array_equal(host(got), host(ref))
        deq = WT.fp8_e4m3_to_f32(codes) * scale[:, None]
        assert np.abs(host(got) - (x @ deq.T + res)).max() < 0.05 * np.abs(x @ deq.T + res).max()
    else:
        ref = torch.zeros(M, N, dtype=torch.float32, device=DEV)
        for r0, m in ranges:
            part = torch.zeros(m, N, dtype=torch.float32, device=DEV)
            narrow(L, DEC_PLAIN, ptr(xd[r0:]), ptr(Wd), m, N, K, out_f32=ptr(part), ldc=N, waves=waves, ksplit=ksplit, w_scale=sd,
                   opts=narrow_opts(atomic_out=True))
            ref[r0:r0 + m] = part
        acc = torch.zeros(M, N, dtype=torch.float32, device=DEV)
        dec32(L, DEC_PLAIN, xp, Wd, M, N, K, waves, ksplit=ksplit, out_f32=acc, ldc=N, atomic=True, w_scale=sd)
        np.testing.assert_array_equal(acc.cpu().numpy(), ref.cpu().numpy())


@pytest.mark.parametrize("H,KVH,K,B,fp8", [(12, 2, 1536, 32, False), (28, 4, 3584, 21, False), (4, 2, 256, 17, False), (12, 2, 1536, 25, True)])
def test_resnorm32_then_qkv32_gives_the_resnorm_and_direct_qkv_bits(L, H, KVH, K, B, fp8):
    """kr_decode_resnorm32 = kr_decode_resnorm with h packed; kr_linear_decode32(ROPE_KV) on it = the direct narrow qkv launch
    (which round 3 pinned, bit for bit, to the fused launch batches of up to 16 rows run): x_new, q, K cache rows, V^T columns."""
    rng = np.random.default_rng(H + K + B)
    hd, s_max, T = 128, 256, 7
    N = (H + 2 * KVH) * hd
    plen = rng.integers(0, 190, size=B).astype(np.int32)
    step = rng.integers(0, T, size=B).astype(np.int32)
    ctxs = plen + step
    x, W = rnd(rng, B, K, scale=2.0), rnd(rng, N, K, scale=K ** -0.5)
    bias, nw = rnd(rng, N, scale=0.1), bf16_round(1 + 0.1 * rnd(rng, K))
    p = (rng.standard_normal((1, B, K)) * 0.5).astype(np.float32)
    ang = rng.uniform(0, 6.28, size=(B, T, 64)).astype(np.float32)
    cs = np.concatenate([bf16_round(np.cos(ang)), bf16_round(np.sin(ang))], -1).astype(np.float32)
    kc, vt = rnd(rng, B, KVH, s_max, hd), rnd(rng, B, KVH, s_max // 64, hd, 64)
    xd, bd, nd = dev_bf16(x), dev_bf16(bias), dev_bf16(nw)
    sd = None
    if fp8:
        codes, scale = WT.quantize_fp8_rows(W)
        Wd, sd = torch.from_numpy(WT.pack_w16x64_fp8(codes)).to(DEV), torch.from_numpy(scale).to(DEV)
    else:
        Wd = dev_bf16(pack_w16x64(W))
    pd = torch.from_numpy(p).to(DEV)
    cs_d, ctx_d, pl_d = torch.from_numpy(cs).to(DEV), torch.from_numpy(ctxs).to(DEV), torch.from_numpy(plen).to(DEV)
    out = {}
    for form in ("rows", "packed"):
        kc_d, vt_d = dev_bf16(kc), dev_bf16(vt)
        q_d = torch.zeros(B, H, hd, dtype=torch.bfloat16, device=DEV)
        xo = torch.full((B, K), 5.0, dtype=torch.bfloat16, device=DEV)
        if form == "rows":
            h = torch.zeros(B, K, dtype=torch.bfloat16, device=DEV)
            L.kr_decode_resnorm(ptr(xd), K, ptr(pd), 1, B, ptr(xo), K, ptr(nd), 1e-6, ptr(h), K, B, K, 0)
            for r0, m in [(0, 16), (16, B - 16)]:
                narrow(L, DEC_ROPE_KV, ptr(h[r0:]), ptr(Wd), m, N, K, bias=ptr(bd), cs=ptr(cs_d[r0:]), cs_stride=T, plen=ptr(pl_d[r0:]),
                       ctx=ptr(ctx_d[r0:]), q_out=ptr(q_d[r0:]), kc=ptr(kc_d[r0:]), vc=ptr(vt_d[r0:]), heads=H, kv_heads=KVH, s_max=s_max,
                       w_scale=sd)
            hrows = host(h)
        else:
            hp = torch.full((32 * K,), 9.0, dtype=torch.bfloat16, device=DEV)
            L.kr_decode_resnorm32(ptr(xd), K, ptr(pd), 1, B, ptr(xo), K, ptr(nd), 1e-6, ptr(hp), B, K, 0, 0)
            np.testing.assert_array_equal(unpack_rows32(host(hp), K)[:B], hrows)
            dec32(L, DEC_ROPE_KV, hp, Wd, B, N, K, 8, bias=bd, cs=cs_d, cs_stride=T, plen=pl_d, ctx=ctx_d, q=q_d, kc=kc_d, vc=vt_d, heads=H,
                  kv_heads=KVH, s_max=s_max, w_scale=sd)
        out[form] = (host(xo), host(q_d), host(kc_d), host(vt_d))
    for a, b, what in zip(out["rows"], out["packed"], ("x_new", "q", "K cache", "V^T cache")):
        np.testing.assert_array_equal(a, b, err_msg=what)


@pytest.mark.parametrize("B,H,ns", [(17, 12, 16), (32, 28, 16), (21, 4, 4), (32, 12, 5)])
def test_attn_decode_merge32_is_the_merge_packed(L, B, H, ns):
    rng = np.random.default_rng(B + H + ns)
    ws = rng.standard_normal((B * H, ns, 132)).astype(np.float32)
    ws[..., 129] = np.abs(ws[..., 129]) + 0.5          # l > 0
    wd = torch.from_numpy(ws).to(DEV)
    rows = torch.zeros(B, H * 128, dtype=torch.bfloat16, device=DEV)
    pk = torch.full((32 * H * 128,), 7.0, dtype=torch.bfloat16, device=DEV)
    L.kr_attn_decode_merge(ptr(wd), ptr(rows), B, H, 128, ns, 0)
    L.kr_attn_decode_merge32(ptr(wd), ptr(pk), B, H, 128, ns, 0)
    got = unpack_rows32(host(pk), H * 128)
    np.testing.assert_array_equal(got[:B], host(rows))
    assert (got[B:] == 7.0).all()                       # rows beyond the batch are not touched


@pytest.mark.parametrize("M,N,K,blocks,waves", [(17, 16 * 2 * 64, 1536, 256, 8), (32, 16 * 40, 1536, 5, 8), (21, 16 * 48, 3584, 3, 8),
                                                (32, 16 * 16, 2048, 2, 8)])
@pytest.mark.parametrize("fp8", [False, True])
def test_gate_up_packed_output_is_the_row_major_output_packed(L, M, N, K, blocks, waves, fp8):
    """kr_linear_decode_wide(KR_DEC_SILU8 | KR_DEC_OUT_XP): the same values, stored in the packed layout (K = 1536 / 2048: all
    rows staged, K = 3584: the K-halves kernel)."""
    rng = np.random.default_rng(M + N + K)
    x, W, nw = rnd(rng, M, K, scale=2.0), rnd(rng, N, K, scale=K ** -0.5), bf16_round(1 + 0.1 * rnd(rng, K))
    xd, nd = dev_bf16(x), dev_bf16(nw)
    nc = N // 2
    rows = torch.zeros(M, nc, dtype=torch.bfloat16, device=DEV)
    pk = torch.full((32 * nc,), 7.0, dtype=torch.bfloat16, device=DEV)
    if fp8:
        codes, scale = WT.quantize_fp8_rows(W)
        Wd, sd = torch.from_numpy(WT.pack_w16x64_fp8(codes)).to(DEV), torch.from_numpy(scale).to(DEV)
        for mode, o in ((DEC_SILU8, rows), (DEC_SILU8 | DEC_OUT_XP, pk)):
            L.kr_linear_decode_wide_fp8(mode, ptr(xd), K, ptr(Wd), ptr(sd), 0, ptr(nd), 1e-6, 0, 0, ptr(o), 0, nc, M, N, K, blocks, waves, 0, 0, 0)
    else:
        Wd = dev_bf16(pack_w16x64(W))
        for mode, o in ((DEC_SILU8, rows), (DEC_SILU8 | DEC_OUT_XP, pk)):
            L.kr_linear_decode_wide(mode, ptr(xd), K, ptr(Wd), 0, ptr(nd), 1e-6, 0, 0, ptr(o), 0, nc, M, N, K, blocks, waves, 0, 0, 0)
    got = unpack_rows32(host(pk), nc)
    np.testing.assert_array_equal(got[:M], host(rows))
    assert (got[M:] == 7.0).all()
    with pytest.raises(KarantaHipError):                 # packed output is the 17..32-row form
        L.kr_linear_decode_wide(DEC_SILU8 | DEC_OUT_XP, ptr(xd), K, ptr(Wd), 0, ptr(nd), 1e-6, 0, 0, ptr(pk), 0, nc, 16, N, K, blocks, waves, 0, 0, 0)


def test_linear_decode32_rejects_bad_arguments(L):
    z = torch.zeros(32 * 64, dtype=torch.bfloat16, device=DEV)
    o = torch.zeros(32, 16, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(KarantaHipError, match="waves_ref"):
        dec32(L, DEC_PLAIN, z, z, 32, 16, 64, 4, out=o, ldc=16)
    with pytest.raises(KarantaHipError, match="M="):
        dec32(L, DEC_PLAIN, z, z, 33, 16, 64, 8, out=o, ldc=16)
    with pytest.raises(KarantaHipError, match="atomic_out"):
        dec32(L, DEC_PLAIN, z, z, 32, 16, 64, 8, out=o, ldc=16, atomic=True)
    with pytest.raises(KarantaHipError, match="mode"):
        dec32(L, DEC_SILU8, z, z, 32, 16, 64, 8, out=o, ldc=16)


@pytest.mark.parametrize("B,K", [(32, 1536), (21, 3584), (17, 256)])
def test_resnorm32_sum_slabs_first_is_the_one_slab_sum(L, B, K):
    """x + (slab 0 + slab 1) (the slabs of a group-split down_proj) = x + the one atomically accumulated slab of <= 16 rows."""
    rng = np.random.default_rng(B + K)
    x, nw = rnd(rng, B, K), rnd(rng, K)
    s0, s1 = (rng.standard_normal((B, K)).astype(np.float32) for _ in range(2))
    xd, nd = dev_bf16(x), dev_bf16(nw)
    two = torch.from_numpy(np.stack([s0, s1])).to(DEV)
    one = torch.from_numpy(s0 + s1).to(DEV)
    outs = []
    for part, n, first in ((two, 2, 1), (one, 1, 0)):
        xo, hp = torch.zeros(B, K, dtype=torch.bfloat16, device=DEV), torch.zeros(32 * K, dtype=torch.bfloat16, device=DEV)
        L.kr_decode_resnorm32(ptr(xd), K, ptr(part), n, B, ptr(xo), K, ptr(nd), 1e-6, ptr(hp), B, K, first, 0)
        outs.append((host(xo), host(hp)))
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    with pytest.raises(KarantaHipError):
        L.kr_decode_resnorm32(ptr(xd), K, ptr(one), 1, B, ptr(xo), K, ptr(nd), 1e-6, ptr(hp), B, K, 1, 0)
