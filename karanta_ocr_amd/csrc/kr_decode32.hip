// Decode linears for batches of 17..32 rows (two 16-row MFMA column tiles): the configuration the continuous server and
// BASELINE config 3's 32-rows-per-GPU variant run.
//
// Why a family of its own.  The <= 16-row narrow kernel (kr_decode.hip, dec_narrow_kernel) reads its x fragments straight
// from L2 in MFMA fragment shape: one wave-instruction covers 16 rows x 64 bytes = sixteen HALF cache lines, and at two
// column tiles a wave moves 4 KiB of x per 2 KiB of weights.  r3 kernel traces: 2B down_proj 14.5 us at 32 rows against 8.2
// at 8 rows for the same 27.5 MB of weights; an XCD's L2 feeds a CU at ~70 GB/s (MI355X_MICROARCH.md, "Indexed rows"), only
// three times what the CU takes from HBM when the whole chip streams.  Here:
//
//   * PACKED ACTIVATIONS (XP layout).  Every producer of a decode linear's input at > 16 rows (kr_decode_resnorm,
//     kr_attn_decode_merge, the SiLU*mul epilogue of the gate/up launch) writes [K/64][2 column tiles][2 k-steps][64 lanes][8]
//     bf16: the 1 KiB a wave needs for one MFMA operand is contiguous (eight whole lines per wave-instruction, the shape of a
//     weight load).  kr_pack_rows32 builds the same layout from row-major rows (tests, generic callers).
//   * ONE K PARTITION, defined by K and the <= 16-row launch's geometry alone (its `waves` and `ksplit`): the same atoms (a
//     wave's contiguous chunk range), each accumulated from zero in ascending k by the same MFMA sequence, folded in the same
//     order.  A wave here may own several atoms (VW "virtual waves": 2B down_proj's 2 x 16 atoms run on 8-wave workgroups) and
//     several weight tiles (NT: one x fragment feeds NT x 2 column tiles), neither of which changes a sum.  A page's tokens
//     therefore do not depend on whether it decodes in a batch of 8 or of 32 (tests: test_gpu_kernels.py, bit equality of
//     kr_linear_decode32 with kr_linear_decode_narrow row for row).
//   * epilogues spread over (column tile, weight tile) waves instead of wave 0 alone; bias / residual / rotary operands are
//     requested ahead of the weight ring as in the narrow kernel.
#include "kr_common.h"

namespace {

constexpr int D32_PLAIN = 0, D32_ROPE_KV = 2, D32_PARTIAL = 16;
constexpr int MT = 2;   // 16-row column tiles of the batch

// One 64-wide K chunk of a 16-row weight tile in registers (formats of kr_decode.hip's WChunk) and where its x operand sits
// inside a column tile's 2 KiB of a packed chunk.
template <bool W8> struct WCh;
template <> struct WCh<false> {
    static constexpr int BYTES = 2048;
    bf16x8 v[2];
    __device__ __forceinline__ void load(const char* p, int64_t c) {
        v[0] = ld8_nt(reinterpret_cast<const kr_bf16*>(p + c * BYTES));
        v[1] = ld8_nt(reinterpret_cast<const kr_bf16*>(p + c * BYTES + 1024));
    }
    __device__ __forceinline__ bf16x8 frag(int h) const { return v[h]; }
    // lane (fr, fg) of k-step h holds k = 32h + 8fg .. +7: block h, lane's own 16 bytes
    static __device__ __forceinline__ int xp_off(int h, int fr, int fg) { return h * 1024 + (fg * 16 + fr) * 16; }
};
template <> struct WCh<true> {
    static constexpr int BYTES = 1024;
    u32x4 q;
    __device__ __forceinline__ void load(const char* p, int64_t c) {
        q = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + c * BYTES));
    }
    __device__ __forceinline__ bf16x8 frag(int h) const {
        u32x4 o;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int w = (int)q[2 * h + i];
            o[2 * i + 0] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, false));
            o[2 * i + 1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, true));
        }
        return __builtin_bit_cast(bf16x8, o);
    }
    // fp8 chunk: lane (fr, fg) of k-step h holds k = 16fg + 8h .. +7 = block (fg >> 1), lane group 2 (fg & 1) + h
    static __device__ __forceinline__ int xp_off(int h, int fr, int fg) { return (fg >> 1) * 1024 + ((((fg & 1) << 1) | h) * 16 + fr) * 16; }
};

struct D32Args {
    const float* w_scale;
    const kr_bf16* bias;
    const kr_bf16* residual; int64_t ldr;
    kr_bf16* out; float* out_f32; int64_t ldc;
    int ksplit, part_atomic;
    float* zero_ptr; int zero_n16;
    const float* cs_table; const int32_t* prompt_len; int cs_stride; const int32_t* ctx_len;
    kr_bf16* q_out; kr_bf16* kcache; kr_bf16* vtcache; int heads, kv_heads, s_max;
};

// WAVES physical waves, each owning VW consecutive atoms of the WAVES * VW the reference partition has; U = ring depth in
// chunks (x fragments and weights of a chunk travel together); NT weight tiles per workgroup.
// GS ("group split", atomic split-K slabs only): the workgroup owns ONE HALF of a K range's atoms — blockIdx.y = 2 * ks + half —
// folds them in wave order and adds the half's sum into slab ks, which the other half's workgroup adds into as well: two
// addends per slab element, so the slab holds (first half) + (second half) = the narrow kernel's fold whichever arrives
// first.  Half the x bytes per weight byte at the same number of workgroups (NT doubles).
template <int NT, int EPI, int WAVES, int VW, int U, bool W8, bool GS = false>
__global__ void __launch_bounds__(WAVES * 64) dec32_kernel(const char* hxp, const char* hwp, int hM, int hN, int hK, int hcpb,
                                                           const D32Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using WC = WCh<W8>;
    constexpr int WLOC = WAVES * VW;                 // atoms of this workgroup
    constexpr int WREF = GS ? 2 * WLOC : WLOC;       // atoms of the reference partition of one K range
    static_assert(!GS || EPI == D32_PARTIAL, "group split feeds atomic slabs");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int g = blockIdx.x, ks = GS ? (int)blockIdx.y >> 1 : (int)blockIdx.y, a0 = GS ? ((int)blockIdx.y & 1) * WLOC : 0;
    const int M = hM, nchunks = hK >> 6, ntiles = hN >> 4;
    const int cb0 = min(ks * hcpb, nchunks), cb1 = min(cb0 + hcpb, nchunks), nblk = cb1 - cb0;
    int ab[VW + 1];   // atom boundaries of this wave (the narrow kernel's c0 / c1 with WAVES = WREF)
#pragma unroll
    for (int v = 0; v <= VW; ++v) ab[v] = cb0 + ((a0 + wave * VW + v) * nblk) / WREF;
    const int c0 = ab[0], c1 = ab[VW];

    int tile[NT];
    if (EPI == D32_ROPE_KV) {   // rotary tile pair: channels i and i + 64 of one head
        tile[0] = (g >> 2) * 8 + (g & 3);
        tile[NT - 1] = tile[0] + 4;
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) tile[t] = g * NT + t;
    }
    const char* wp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wp[t] = hwp + ((int64_t)min(tile[t], ntiles - 1) * nchunks) * WC::BYTES + lane * 16;
    const char* xq[2];   // + c * 4096 + mt * 2048
#pragma unroll
    for (int h = 0; h < 2; ++h) xq[h] = hxp + WC::xp_off(h, fr, fg);

    // ---- 1. x fragments of the first U chunks, then the epilogue's scalars, then the weight ring (loads return in issue order)
    bf16x8 xf[U][MT][2];
    WC wbuf[U][NT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int c = min(max(min(c0 + u, c1 - 1), cb0), nchunks - 1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int h = 0; h < 2; ++h) xf[u][mt][h] = *reinterpret_cast<const bf16x8*>(xq[h] + (int64_t)c * 4096 + mt * 2048);
    }
    // epilogue role of this wave: column tile emt, weight tile et (ROPE_KV: the pair)
    constexpr int EW = EPI == D32_ROPE_KV ? MT : MT * NT;     // waves that run an epilogue
    const int emt = wave & 1, et = EPI == D32_ROPE_KV ? 0 : ((wave >> 1) % NT);
    const int eb = fr + 16 * emt, erb = min(eb, M - 1);
    int pos = 0, plen = 0;
    if (EPI == D32_ROPE_KV) {
        pos = a.ctx_len[erb];
        plen = a.prompt_len[erb];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int c = min(max(min(c0 + u, c1 - 1), cb0), nchunks - 1);
#pragma unroll
        for (int t = 0; t < NT; ++t) wbuf[u][t].load(wp[t], c);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (a.zero_ptr) {   // zero the split-K accumulator the NEXT down_proj adds into (behind this launch's own requests)
        const int zstep = (int)gridDim.x * (int)gridDim.y * (WAVES * 64);
        for (int zi = ((int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x) * (WAVES * 64) + tid; zi < a.zero_n16; zi += zstep)
            reinterpret_cast<f32x4*>(a.zero_ptr)[zi] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // ---- 2. epilogue operands, in flight during the K loop (unconditional loads from always-valid addresses)
    float csv[8];
    bf16x4 bias0 = {}, bias1 = {}, res_pre = {};
#pragma unroll
    for (int j = 0; j < 8; ++j) csv[j] = 0.f;
    const int en = min(tile[et], ntiles - 1) * 16 + fg * 4;
    if (EPI == D32_ROPE_KV) {
        const int i0 = (tile[0] & 7) * 16 + fg * 4;
        const float* cs = a.cs_table + ((int64_t)erb * a.cs_stride + (pos - plen)) * 128;
        const f32x4 cv = *reinterpret_cast<const f32x4*>(cs + i0), sv = *reinterpret_cast<const f32x4*>(cs + 64 + i0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            csv[j] = cv[j];
            csv[4 + j] = sv[j];
        }
        bias0 = *reinterpret_cast<const bf16x4*>(a.bias + tile[0] * 16 + fg * 4);
        bias1 = *reinterpret_cast<const bf16x4*>(a.bias + tile[NT - 1] * 16 + fg * 4);
    } else if (EPI == D32_PLAIN) {
        bias0 = *reinterpret_cast<const bf16x4*>(a.bias ? reinterpret_cast<const char*>(a.bias + en) : hxp);
        res_pre = *reinterpret_cast<const bf16x4*>(a.residual ? reinterpret_cast<const char*>(a.residual + (int64_t)erb * a.ldr + en) : hxp);
    }

    // ---- 3. K loop: atoms in order, each from zero
    f32x4 acc[VW][MT][NT];
#pragma unroll
    for (int v = 0; v < VW; ++v)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[v][mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int cc = c0; cc < c1; cc += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = cc + u;
            if (c < c1) {
                auto mac = [&](f32x4 (&ac)[MT][NT]) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const bf16x8 w0 = wbuf[u][t].frag(0), w1 = wbuf[u][t].frag(1);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            ac[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, xf[u][mt][0], ac[mt][t], 0, 0, 0);
                            ac[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, xf[u][mt][1], ac[mt][t], 0, 0, 0);
                        }
                    }
                };
                if constexpr (VW == 1) {
                    mac(acc[0]);
                } else if constexpr (VW == 2) {
                    if (c < ab[1]) mac(acc[0]);
                    else mac(acc[1]);
                } else {
                    if (c < ab[1]) mac(acc[0]);
                    else if (c < ab[2]) mac(acc[1]);
                    else if (c < ab[VW - 1]) mac(acc[VW - 2]);
                    else mac(acc[VW - 1]);
                }
                const int cn = c + U;
                if (cn < c1) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int h = 0; h < 2; ++h) xf[u][mt][h] = *reinterpret_cast<const bf16x8*>(xq[h] + (int64_t)cn * 4096 + mt * 2048);
#pragma unroll
                    for (int t = 0; t < NT; ++t) wbuf[u][t].load(wp[t], cn);
                }
            }
        }
    }
    // ---- 4. fold of the atoms in reference order: red[mt][atom][t][lane] f32x4
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int v = 0; v < VW; ++v)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                *reinterpret_cast<f32x4*>(red + ((((mt * WLOC) + wave * VW + v) * NT + t) * 64 + lane) * 4) = acc[v][mt][t];
    __syncthreads();
    if constexpr (GS) {   // roles (column tile, weight tile) dealt over the waves; every role: fold of this half, add into slab ks
        for (int role = wave; role < MT * NT; role += WAVES) {
            const int rmt = role & 1, rt = role >> 1, rb = fr + 16 * rmt;
            f32x4 sm = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < WLOC; ++w) {
                const f32x4 p = *reinterpret_cast<const f32x4*>(red + ((((rmt * WLOC) + w) * NT + rt) * 64 + lane) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) sm[j] += p[j];
            }
            if (rb < M && tile[rt] < ntiles) {
                float* dst = a.out_f32 + ((int64_t)ks * M + rb) * a.ldc + tile[rt] * 16 + fg * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) atomicAdd(dst + j, sm[j]);
            }
        }
        return;
    }
    static_assert(GS || (EPI == D32_ROPE_KV ? MT : MT * NT) <= WAVES, "one epilogue role per wave");
    if (wave >= EW) return;
    constexpr int ET = EPI == D32_ROPE_KV ? NT : 1;   // tiles this epilogue wave folds
    f32x4 sum[ET];
#pragma unroll
    for (int e = 0; e < ET; ++e) {
        const int t = EPI == D32_ROPE_KV ? e : et;
        // the narrow kernel's fold: wave order inside each half of the atoms, then the two halves added (GS: this workgroup's
        // atoms ARE one half; the slab add is the second level)
        constexpr int NH = GS ? 1 : 2;
        f32x4 half[NH];
#pragma unroll
        for (int hfi = 0; hfi < NH; ++hfi) {
            half[hfi] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = hfi * (WLOC / NH); w < (hfi + 1) * (WLOC / NH); ++w) {
                const f32x4 p = *reinterpret_cast<const f32x4*>(red + ((((emt * WLOC) + w) * NT + t) * 64 + lane) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) half[hfi][j] += p[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) sum[e][j] = GS ? half[0][j] : half[0][j] + half[NH - 1][j];
        if (a.w_scale) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(a.w_scale + min(tile[t], ntiles - 1) * 16 + fg * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) sum[e][j] *= sc[j];
        }
    }
    if (eb >= M) return;
    if constexpr (EPI == D32_ROPE_KV) {
        const int hh = tile[0] >> 3;                   // head index in [q heads | k heads | v heads]
        const int i0 = (tile[0] & 7) * 16 + fg * 4;    // channel in [0, 64)
        float lo[4], hi[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            lo[j] = bfround(sum[0][j] + bf2f(bias0[j]));       // the projection output is a bf16 tensor
            hi[j] = bfround(sum[NT - 1][j] + bf2f(bias1[j]));
        }
        if (hh < a.heads + a.kv_heads) {
            bf16x4 o0, o1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o0[j] = f2bf(lo[j] * csv[j] - hi[j] * csv[4 + j]);
                o1[j] = f2bf(hi[j] * csv[j] + lo[j] * csv[4 + j]);
            }
            kr_bf16* dst = hh < a.heads ? a.q_out + ((int64_t)eb * a.heads + hh) * 128
                                        : a.kcache + (((int64_t)eb * a.kv_heads + (hh - a.heads)) * a.s_max + pos) * 128;
            *reinterpret_cast<bf16x4*>(dst + i0) = o0;
            *reinterpret_cast<bf16x4*>(dst + 64 + i0) = o1;
        } else {
            const int kvh = hh - a.heads - a.kv_heads;
            kr_bf16* vt = a.vtcache + (((int64_t)eb * a.kv_heads + kvh) * (a.s_max >> 6) + (pos >> 6)) * (128 * 64) + kr_vt_off(0, pos & 63, 128);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                vt[(i0 + j) * 32] = __builtin_bit_cast(kr_bf16, f2bf(lo[j]));
                vt[(64 + i0 + j) * 32] = __builtin_bit_cast(kr_bf16, f2bf(hi[j]));
            }
        }
    } else {
        if (tile[et] >= ntiles) return;
        const int n = tile[et] * 16 + fg * 4;
        if constexpr (EPI == D32_PARTIAL) {
            if (a.part_atomic) {
                float* dst = a.out_f32 + (int64_t)eb * a.ldc + n;
#pragma unroll
                for (int j = 0; j < 4; ++j) atomicAdd(dst + j, sum[0][j]);
            } else {
                *reinterpret_cast<f32x4*>(a.out_f32 + ((int64_t)ks * M + eb) * a.ldc + n) = sum[0];
            }
        } else {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = sum[0][j];
            if (a.bias) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += bf2f(bias0[j]);
            }
            if (a.residual) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += bf2f(res_pre[j]);
            }
            if (a.out_f32) {
                *reinterpret_cast<f32x4*>(a.out_f32 + (int64_t)eb * a.ldc + n) = (f32x4){v[0], v[1], v[2], v[3]};
            } else {
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = f2bf(v[j]);
                *reinterpret_cast<bf16x4*>(a.out + (int64_t)eb * a.ldc + n) = o;
            }
        }
    }
}

template <int NT, int EPI, int WAVES, int VW, int U, bool W8>
int launch32(const kr_dec32& q, const D32Args& a, int groups, kr_stream s) {
    const int nchunks = q.K >> 6, cpb = (nchunks + q.ksplit - 1) / q.ksplit;
    const size_t lds = (size_t)MT * WAVES * VW * NT * 1024;
    KR_CHECK_ARG(lds <= 160 * 1024, "kr_linear_decode32: %zu bytes of LDS", lds);
    auto fn = &dec32_kernel<NT, EPI, WAVES, VW, U, W8>;
    static KrPerDeviceOnce attr;
    if (attr.need()) {
        KR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    fn<<<dim3(groups, q.ksplit), WAVES * 64, lds, kr_hs(s)>>>(reinterpret_cast<const char*>(q.xp), reinterpret_cast<const char*>(q.w_packed),
                                                              q.M, q.N, q.K, cpb, a);
    KR_CHECK_LAUNCH();
    return KR_OK;
}

// ring depth by the busiest wave's share of K (every chunk requested up front when it fits); register budgets: 8 waves
// 256 VGPRs (x 16 + weights 8 NT per chunk, fp8 4 NT), 16 waves 128
template <int NT, int EPI, int WAVES, int VW, bool W8>
int launch32_u(const kr_dec32& q, const D32Args& a, int groups, kr_stream s) {
    const int nchunks = q.K >> 6, cpb = (nchunks + q.ksplit - 1) / q.ksplit;
    const int share = (cpb * VW + WAVES * VW - 1) / (WAVES * VW);   // chunks the busiest wave owns (all its atoms)
    if constexpr (WAVES == 16) {
        return launch32<NT, EPI, WAVES, VW, 3, W8>(q, a, groups, s);
    } else if constexpr (NT == 2) {
        if (share <= 3) return launch32<NT, EPI, WAVES, VW, 3, W8>(q, a, groups, s);
        return launch32<NT, EPI, WAVES, VW, 5, W8>(q, a, groups, s);
    } else {
        if (share <= 3) return launch32<NT, EPI, WAVES, VW, 3, W8>(q, a, groups, s);
        if (share <= 5) return launch32<NT, EPI, WAVES, VW, 5, W8>(q, a, groups, s);
        return launch32<NT, EPI, WAVES, VW, 8, W8>(q, a, groups, s);
    }
}

// group split: grid.y = 2 * ksplit (K range, half of its atoms); bf16 weights (a row scale would have to multiply the SUM of
// the two halves).  8-atom partitions run 4-wave workgroups with NT tiles, 16-atom partitions 8-wave workgroups.
template <int NT, int WAVES, int U>
int launch32_gs(const kr_dec32& q, const D32Args& a, int groups, kr_stream s) {
    const int nchunks = q.K >> 6, cpb = (nchunks + q.ksplit - 1) / q.ksplit;
    const size_t lds = (size_t)MT * WAVES * NT * 1024;
    auto fn = &dec32_kernel<NT, D32_PARTIAL, WAVES, 1, U, false, true>;
    fn<<<dim3(groups, 2 * q.ksplit), WAVES * 64, lds, kr_hs(s)>>>(reinterpret_cast<const char*>(q.xp), reinterpret_cast<const char*>(q.w_packed),
                                                                  q.M, q.N, q.K, cpb, a);
    KR_CHECK_LAUNCH();
    return KR_OK;
}

template <int NT, int EPI, bool W8>
int launch32_w(const kr_dec32& q, const D32Args& a, int groups, kr_stream s) {
    // physical waves: 8; a 16-atom reference partition runs as two atoms per wave
    if (q.waves_ref == 16) return launch32_u<NT, EPI, 8, 2, W8>(q, a, groups, s);
    return launch32_u<NT, EPI, 8, 1, W8>(q, a, groups, s);
}

template <int NT, int EPI>
int launch32_q(const kr_dec32& q, const D32Args& a, int groups, kr_stream s) {
    return q.w_scale ? launch32_w<NT, EPI, true>(q, a, groups, s) : launch32_w<NT, EPI, false>(q, a, groups, s);
}

// row-major rows -> XP layout: thread = one 16-byte piece (row b, 8 consecutive k)
__global__ void __launch_bounds__(256) pack_rows32_kernel(const kr_bf16* __restrict__ x, int64_t ldx, int M, int K, kr_bf16* __restrict__ xp) {
    const int kc = K >> 3, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 32 * kc) return;
    const int b = i / kc, c = i - b * kc;
    bf16x8 v = {};
    if (b < M) v = ld8(x + (int64_t)b * ldx + c * 8);
    *reinterpret_cast<bf16x8*>(reinterpret_cast<char*>(xp) + kr_xp_byte_offset(b, c * 8)) = v;
}

}  // namespace

extern "C" int kr_pack_rows32(const kr_bf16* x, int64_t ldx, int M, int K, kr_bf16* xp, kr_stream s) {
    KR_CHECK_ARG(x && xp && M >= 1 && M <= 32 && K > 0 && K % 64 == 0 && ldx >= K && (ldx & 7) == 0, "kr_pack_rows32: M=%d K=%d ldx=%lld", M, K,
                 (long long)ldx);
    pack_rows32_kernel<<<(32 * (K >> 3) + 255) / 256, 256, 0, kr_hs(s)>>>(x, ldx, M, K, xp);
    KR_CHECK_LAUNCH();
    return KR_OK;
}

extern "C" int kr_linear_decode32(int mode, const kr_dec32* qp, kr_stream s) {
    KR_CHECK_ARG(qp, "kr_linear_decode32: null args");
    const kr_dec32& q = *qp;
    KR_CHECK_ARG(q.xp && q.w_packed && ((uintptr_t)q.xp & 15) == 0, "kr_linear_decode32: null / unaligned pointer");
    KR_CHECK_ARG(q.M >= 1 && q.M <= 32, "kr_linear_decode32: M=%d must be in 1..32", q.M);
    KR_CHECK_ARG(q.N > 0 && q.N % 16 == 0 && q.K > 0 && q.K % 64 == 0, "kr_linear_decode32: N=%d K=%d (N%%16, K%%64)", q.N, q.K);
    KR_CHECK_ARG(q.waves_ref == 8 || q.waves_ref == 16, "kr_linear_decode32: waves_ref=%d (8 or 16: the <= 16-row launch's waves)", q.waves_ref);
    KR_CHECK_ARG(q.ksplit >= 1 && q.ksplit <= 8 && q.ksplit <= (q.K >> 6), "kr_linear_decode32: ksplit=%d", q.ksplit);
    KR_CHECK_ARG((q.zero_ptr || q.zero_bytes == 0) && ((uintptr_t)q.zero_ptr & 15) == 0 && (q.zero_bytes & 15) == 0 && q.zero_bytes < (1u << 30),
                 "kr_linear_decode32: zero range");
    KR_CHECK_ARG(q.tiles_per_wg >= 0 && (q.tiles_per_wg <= 2 || (q.group_split && q.tiles_per_wg == 4)),
                 "kr_linear_decode32: tiles_per_wg=%d (0 = automatic, 1, 2; 4 with group_split)", q.tiles_per_wg);
    KR_CHECK_ARG(!q.group_split || (mode == D32_PLAIN && q.ksplit == 2), "kr_linear_decode32: group_split is for PLAIN, ksplit 2");
    D32Args a{};
    a.w_scale = q.w_scale; a.bias = q.bias; a.residual = q.residual; a.ldr = q.ldr;
    a.out = q.out; a.out_f32 = q.out_f32; a.ldc = q.ldc; a.ksplit = q.ksplit; a.part_atomic = q.atomic_out ? 1 : 0;
    a.zero_ptr = q.zero_bytes ? q.zero_ptr : nullptr; a.zero_n16 = (int)(q.zero_bytes >> 4);
    a.cs_table = q.cs_table; a.prompt_len = q.prompt_len; a.cs_stride = q.cs_stride; a.ctx_len = q.ctx_len;
    a.q_out = q.q_out; a.kcache = q.kcache; a.vtcache = q.vtcache; a.heads = q.heads; a.kv_heads = q.kv_heads; a.s_max = q.s_max;
    const int ntiles = q.N >> 4;
    switch (mode) {
        case D32_PLAIN: {
            KR_CHECK_ARG(q.ldc >= q.N && (q.ldc & 3) == 0 && (!q.residual || (q.ldr & 3) == 0), "kr_linear_decode32: PLAIN ldc / ldr");
            // two weight tiles per workgroup share the x fragments (x : weight bytes 1 : 1 instead of 2 : 1) where that still
            // leaves a workgroup for most compute units
            const bool nt2 = q.tiles_per_wg ? q.tiles_per_wg == 2 : ((ntiles & 1) == 0 && ntiles / 2 * q.ksplit >= 192);
            if (q.ksplit > 1) {
                KR_CHECK_ARG(q.out_f32 && !q.out && !q.bias && !q.residual, "kr_linear_decode32: split-K writes f32 slabs only (no bias / residual)");
                KR_CHECK_ARG(!a.part_atomic || q.ksplit == 2, "kr_linear_decode32: atomic_out is for ksplit 2 (order-free sum)");
                if (q.group_split) {
                    KR_CHECK_ARG(a.part_atomic && !q.w_scale, "kr_linear_decode32: group_split needs atomic_out and bf16 weights");
                    const int cpb = ((q.K >> 6) + q.ksplit - 1) / q.ksplit, share = (cpb + q.waves_ref - 1) / q.waves_ref;
                    const int tw = q.tiles_per_wg ? q.tiles_per_wg : (q.waves_ref == 8 ? 4 : 2);
                    KR_CHECK_ARG(ntiles % tw == 0 && (tw == 2 || tw == 4), "kr_linear_decode32: group_split tiles_per_wg=%d over %d tiles", tw, ntiles);
                    if (q.waves_ref == 8) {
                        if (tw == 4) return launch32_gs<4, 4, 4>(q, a, ntiles / 4, s);
                        return share <= 5 ? launch32_gs<2, 4, 5>(q, a, ntiles / 2, s) : launch32_gs<2, 4, 8>(q, a, ntiles / 2, s);
                    }
                    if (tw == 4) return launch32_gs<4, 8, 3>(q, a, ntiles / 4, s);
                    return share <= 3 ? launch32_gs<2, 8, 3>(q, a, ntiles / 2, s) : launch32_gs<2, 8, 5>(q, a, ntiles / 2, s);
                }
                return nt2 ? launch32_q<2, D32_PARTIAL>(q, a, (ntiles + 1) / 2, s) : launch32_q<1, D32_PARTIAL>(q, a, ntiles, s);
            }
            KR_CHECK_ARG(q.out || q.out_f32, "kr_linear_decode32: PLAIN output");
            KR_CHECK_ARG(!a.part_atomic, "kr_linear_decode32: atomic_out needs ksplit 2");
            return nt2 ? launch32_q<2, D32_PLAIN>(q, a, (ntiles + 1) / 2, s) : launch32_q<1, D32_PLAIN>(q, a, ntiles, s);
        }
        case D32_ROPE_KV:
            KR_CHECK_ARG(q.bias && q.cs_table && q.prompt_len && q.ctx_len && q.q_out && q.kcache && q.vtcache && q.cs_stride > 0,
                         "kr_linear_decode32: ROPE_KV pointers");
            KR_CHECK_ARG(q.N == (q.heads + 2 * q.kv_heads) * 128 && q.s_max % 64 == 0 && q.ksplit == 1 && q.waves_ref == 8,
                         "kr_linear_decode32: ROPE_KV needs head_dim 128, ksplit 1, waves_ref 8");
            return launch32_q<2, D32_ROPE_KV>(q, a, ntiles / 2, s);
        default:
            kr_set_error("kr_linear_decode32: mode %d not supported (PLAIN, ROPE_KV)", mode);
            return KR_ERR_ARG;
    }
}
