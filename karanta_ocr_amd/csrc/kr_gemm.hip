// Dense linears on the MI355X matrix cores.
//
//  gemm_kernel : C[M,N] = epi(A[M,K] W[N,K]^T + bias) (+R)   M large (ViT, merger, decoder prefill)
//                128x128x64 block tile, 4 waves (2x2, 64x64 each) or 256x256x64, 8 waves (2x4, 128x64 each),
//                v_mfma_f32_16x16x32_bf16,
//                both operands K-contiguous, staged HBM->LDS with 16-byte LDS-DMA
//                (global_load_lds_dwordx4) into an XOR-swizzled image (swizzle applied on the
//                per-lane SOURCE address, linear LDS destination), double buffered.
//                MFMA-bound; roofline = 2.5 PFLOP/s dense bf16.
//  gemv_kernel : same contract for M <= 16 (decode).  Weights go HBM -> VGPR exactly once with
//                non-temporal 16-byte loads, 8+ loads in flight per wave; x (<= 16 rows) is staged
//                (optionally RMS-normalised) in LDS.  HBM-bound; roofline = 8 TB/s.
#include "kr_common.h"

// =====================================================================================
// GEMM
// =====================================================================================
namespace {

constexpr int BK = 64;
// Tile geometries.  G128: 128x128 block, 4 waves (2x2) of 64x64 — 2-3 workgroups per CU hide the staging
// latency of each other; the shape for small / ragged M.  G256: 256x256 block, 8 waves (2x4) of 128x64 — half the
// LDS-DMA traffic and a quarter of the barriers per flop, 0.375 instead of 0.5 LDS fragment reads per MFMA, one
// workgroup per CU (128 KiB of LDS, 128 accumulator registers); for the large-M GEMMs of the ViT and the prefill.
struct G128 { static constexpr int BM = 128, BN = 128, WM = 2, WN = 2; };
struct G256 { static constexpr int BM = 256, BN = 256, WM = 2, WN = 4; };

// LDS image of a [128 rows][64 k] bf16 tile: 128-byte rows of eight 16-byte chunks; chunk c of
// row r is stored at chunk position c ^ ((r >> 1) & 7): conflict-free for the ds_read_b128
// fragment reads below (16 distinct rows x 4 k-chunks per instruction).
__device__ __forceinline__ int lds_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

// Stage one [ROWS][64] tile (rows row0.., k k0..) of a row-major [rows_total][ld] matrix.
// Each wave-instruction writes 1 KiB contiguous LDS (= 8 tile rows).
// PACKED: the matrix is stored in the decode layout [rows/16][ld/32][4][16][8] (one MFMA fragment
// block = 1 KiB, see kr_decode.hip): chunk c (8 k) of row r lives at block (r/16, k/32), group
// (k%32)/8, row r%16 — a wave-instruction still fetches eight 128-byte segments.
template <bool PACKED, int ROWS, int NTHR>
__device__ __forceinline__ void stage_tile(const kr_bf16* __restrict__ g, int64_t ld, int64_t row0, int64_t rows_total,
                                           int k0, char* lds_tile, int tid, int wave) {
#pragma unroll
    for (int p = 0; p < ROWS * 8 / NTHR; ++p) {
        const int idx = p * NTHR + tid;  // chunk index inside the tile image
        const int r = idx >> 3, cp = idx & 7;
        const int c = cp ^ ((r >> 1) & 7);  // which global chunk lands at this LDS position
        int64_t gr = row0 + r;
        gr = gr < rows_total ? gr : rows_total - 1;
        const kr_bf16* src = PACKED ? g + ((((gr >> 4) * (ld >> 5) + (k0 >> 5) + (c >> 2)) * 4 + (c & 3)) * 16 + (gr & 15)) * 8
                                    : g + gr * ld + k0 + c * 8;
        char* dst = lds_tile + (p * NTHR + wave * 64) * 16;  // wave-uniform; hardware adds lane*16
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
}

template <int EPI, bool WPACK, typename G>
__global__ void __launch_bounds__(G::WM * G::WN * 64) gemm_kernel(const kr_bf16* __restrict__ A, int64_t lda,
                                                   const kr_bf16* __restrict__ W, const kr_bf16* __restrict__ bias,
                                                   const kr_bf16* __restrict__ R, int64_t ldr, kr_bf16* __restrict__ C,
                                                   int64_t ldc, int64_t M, int N, int K, int tiles_n, unsigned nwg) {
    constexpr int BM = G::BM, BN = G::BN, NTHR = G::WM * G::WN * 64;
    constexpr int WTM = BM / G::WM, WTN = BN / G::WN;  // wave tile
    constexpr int MT = WTM / 16, NT = WTN / 16;
    constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2, STAGE = A_BYTES + W_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 x (A tile, W tile)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / G::WN, wc = wave % G::WN;

    const unsigned wg = xcd_remap(blockIdx.x, nwg);
    const int64_t m0 = (int64_t)(wg / tiles_n) * BM;
    const int n0 = (int)(wg % tiles_n) * BN;

    f32x4 acc[NT][MT];  // [nt][mt]: rows (regs) = n, col (lane&15) = m
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
    stage_tile<false, BM, NTHR>(A, lda, m0, M, 0, smem, tid, wave);
    stage_tile<WPACK, BN, NTHR>(W, K, n0, N, 0, smem + A_BYTES, tid, wave);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int fr = lane & 15, fg = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        char* cur = smem + (kt & 1) * STAGE;
        if (kt + 1 < nk) {
            char* nxt = smem + ((kt + 1) & 1) * STAGE;
            stage_tile<false, BM, NTHR>(A, lda, m0, M, (kt + 1) * BK, nxt, tid, wave);
            stage_tile<WPACK, BN, NTHR>(W, K, n0, N, (kt + 1) * BK, nxt + A_BYTES, tid, wave);
        }
        const char* At = cur;
        const char* Wt = cur + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 xa[MT], wb[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t)
                wb[t] = *reinterpret_cast<const bf16x8*>(Wt + lds_off(wc * WTN + t * 16 + fr, ks * 4 + fg));
#pragma unroll
            for (int t = 0; t < MT; ++t)
                xa[t] = *reinterpret_cast<const bf16x8*>(At + lds_off(wr * WTM + t * 16 + fr, ks * 4 + fg));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[nt], xa[mt], acc[nt][mt], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---------------- epilogue: lane holds 4 consecutive n for one m
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int64_t m = m0 + wr * WTM + mt * 16 + fr;
        if (EPI != KR_EPI_SILU_MUL8 && m >= M) continue;  // (SILU_MUL8 shuffles across lanes: no early exit)
        if (EPI == KR_EPI_SILU_MUL8) {
            // gate/up interleaved in groups of 8 rows: a 16-row tile holds gate rows in lane groups 0,1
            // and the matching up rows in lane groups 2,3 (lane ^ 32)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = n0 + wc * WTN + nt * 16;
                float g[4], u[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] = acc[nt][mt][j];
                if (bias && n < N) {  // bias rows are interleaved like the weight rows: every lane adds its own
                    const bf16x4 bv = *reinterpret_cast<const bf16x4*>(bias + n + fg * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) g[j] += bf2f(bv[j]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) u[j] = __shfl_xor(g[j], 32, 64);
                if (fg < 2 && n < N && m < M) {
                    bf16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = f2bf(act_silu(g[j]) * u[j]);
                    *reinterpret_cast<bf16x4*>(C + m * ldc + (n >> 1) + fg * 4) = o;
                }
            }
        } else if (EPI == KR_EPI_SILU_MUL) {
#pragma unroll
            for (int pr = 0; pr < NT / 2; ++pr) {
                const int n = n0 + wc * WTN + pr * 32 + fg * 4;  // gate row index in W'
                if (n >= N) continue;
                const int oc = ((n0 + wc * WTN) >> 1) + pr * 16 + fg * 4;
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = f2bf(act_silu(acc[2 * pr][mt][j]) * acc[2 * pr + 1][mt][j]);
                *reinterpret_cast<bf16x4*>(C + m * ldc + oc) = o;
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = n0 + wc * WTN + nt * 16 + fg * 4;
                if (n >= N) continue;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[nt][mt][j];
                if (bias) {
                    const bf16x4 bv = *reinterpret_cast<const bf16x4*>(bias + n);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += bf2f(bv[j]);
                }
                if (EPI == KR_EPI_QUICK_GELU) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = act_quick_gelu(v[j]);
                } else if (EPI == KR_EPI_GELU_ERF) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = act_gelu_erf(v[j]);
                }
                if (R) {
                    const bf16x4 rv = *reinterpret_cast<const bf16x4*>(R + m * ldr + n);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += bf2f(rv[j]);
                }
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = f2bf(v[j]);
                *reinterpret_cast<bf16x4*>(C + m * ldc + n) = o;
            }
        }
    }
}

template <int EPI, bool WPACK, typename G>
int launch_gemm3(const kr_bf16* A, int64_t lda, const kr_bf16* W, const kr_bf16* bias, const kr_bf16* R, int64_t ldr,
                 kr_bf16* C, int64_t ldc, int64_t M, int N, int K, kr_stream s) {
    constexpr int LDS = 2 * (G::BM + G::BN) * BK * 2;
    const int64_t tiles_m = (M + G::BM - 1) / G::BM;
    const int tiles_n = (N + G::BN - 1) / G::BN;
    const int64_t nwg = tiles_m * tiles_n;
    KR_CHECK_ARG(nwg < (1ll << 31), "kr_gemm_bf16: grid too large");
    static bool attr_set = false;
    if (!attr_set) {
        KR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<EPI, WPACK, G>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set = true;
    }
    gemm_kernel<EPI, WPACK, G><<<(unsigned)nwg, G::WM * G::WN * 64, LDS, kr_hs(s)>>>(A, lda, W, bias, R, ldr, C, ldc, M, N, K,
                                                                                      tiles_n, (unsigned)nwg);
    KR_CHECK_LAUNCH();
    return KR_OK;
}

// 0 = automatic, 128 / 256 = forced (KARANTA_GEMM_TILE, tuning sweeps)
inline int gemm_tile_choice(int64_t M, int N, int K) {
    const char* env = getenv("KARANTA_GEMM_TILE");  // read per call: the tests flip it between launches
    const int forced = env ? atoi(env) : 0;
    if (forced == 128 || forced == 256) return forced;
    // the 256x256 tile needs enough workgroups to fill 256 CUs a few times over, no N padding, and either several
    // N tiles or a long K (measured on the ViT / prefill shapes, gemm_microbench.py: +10-20 % there, -3 % on
    // N = 1280, K = 1280)
    const int64_t wgs = ((M + 255) / 256) * ((N + 255) / 256);
    return (wgs >= 512 && N % 256 == 0 && (N >= 2048 || K >= 4096)) ? 256 : 128;
}

template <int EPI, bool WPACK>
int launch_gemm2(const kr_bf16* A, int64_t lda, const kr_bf16* W, const kr_bf16* bias, const kr_bf16* R, int64_t ldr,
                 kr_bf16* C, int64_t ldc, int64_t M, int N, int K, kr_stream s) {
    if (gemm_tile_choice(M, N, K) == 256) return launch_gemm3<EPI, WPACK, G256>(A, lda, W, bias, R, ldr, C, ldc, M, N, K, s);
    return launch_gemm3<EPI, WPACK, G128>(A, lda, W, bias, R, ldr, C, ldc, M, N, K, s);
}

template <int EPI>
int launch_gemm(const kr_bf16* A, int64_t lda, const kr_bf16* W, const kr_bf16* bias, const kr_bf16* R, int64_t ldr,
                kr_bf16* C, int64_t ldc, int64_t M, int N, int K, int w_packed, kr_stream s) {
    return w_packed ? launch_gemm2<EPI, true>(A, lda, W, bias, R, ldr, C, ldc, M, N, K, s)
                    : launch_gemm2<EPI, false>(A, lda, W, bias, R, ldr, C, ldc, M, N, K, s);
}

}  // namespace

extern "C" int kr_gemm_bf16(const kr_bf16* A, int64_t lda, const kr_bf16* W, const kr_bf16* bias,
                            const kr_bf16* residual, int64_t ldr, kr_bf16* C, int64_t ldc, int64_t M, int N, int K,
                            int epilogue, int w_packed, kr_stream s) {
    KR_CHECK_ARG(A && W && C, "kr_gemm_bf16: null pointer");
    KR_CHECK_ARG(M >= 0 && N > 0 && K > 0, "kr_gemm_bf16: bad sizes M=%lld N=%d K=%d", (long long)M, N, K);
    KR_CHECK_ARG(K % BK == 0, "kr_gemm_bf16: K=%d must be a multiple of %d", K, BK);
    KR_CHECK_ARG(N % 16 == 0, "kr_gemm_bf16: N=%d must be a multiple of 16", N);
    KR_CHECK_ARG(lda >= K && (lda & 7) == 0, "kr_gemm_bf16: lda=%lld", (long long)lda);
    KR_CHECK_ARG((ldc & 3) == 0 && (residual == nullptr || (ldr & 3) == 0), "kr_gemm_bf16: ldc/ldr alignment");
    KR_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0 && ((uintptr_t)C & 7) == 0,
                 "kr_gemm_bf16: pointer alignment");
    if (M == 0) return KR_OK;
    switch (epilogue) {
        case KR_EPI_NONE:
            KR_CHECK_ARG(ldc >= N, "kr_gemm_bf16: ldc < N");
            return launch_gemm<KR_EPI_NONE>(A, lda, W, bias, residual, ldr, C, ldc, M, N, K, w_packed, s);
        case KR_EPI_QUICK_GELU:
            KR_CHECK_ARG(ldc >= N, "kr_gemm_bf16: ldc < N");
            return launch_gemm<KR_EPI_QUICK_GELU>(A, lda, W, bias, residual, ldr, C, ldc, M, N, K, w_packed, s);
        case KR_EPI_GELU_ERF:
            KR_CHECK_ARG(ldc >= N, "kr_gemm_bf16: ldc < N");
            return launch_gemm<KR_EPI_GELU_ERF>(A, lda, W, bias, residual, ldr, C, ldc, M, N, K, w_packed, s);
        case KR_EPI_SILU_MUL:
            KR_CHECK_ARG(N % 32 == 0 && ldc >= N / 2 && !bias && !residual,
                         "kr_gemm_bf16: SILU_MUL needs N%%32==0, no bias/residual");
            return launch_gemm<KR_EPI_SILU_MUL>(A, lda, W, bias, residual, ldr, C, ldc, M, N, K, w_packed, s);
        case KR_EPI_SILU_MUL8:
            KR_CHECK_ARG(ldc >= N / 2 && !residual, "kr_gemm_bf16: SILU_MUL8 takes no residual");
            return launch_gemm<KR_EPI_SILU_MUL8>(A, lda, W, bias, residual, ldr, C, ldc, M, N, K, w_packed, s);
        default:
            kr_set_error("kr_gemm_bf16: unknown epilogue %d", epilogue);
            return KR_ERR_ARG;
    }
}

// =====================================================================================
// GEMV (M <= 16): decode linears
// =====================================================================================
namespace {

constexpr int GV_U = 4;  // K-chunks (64 k each) in flight per wave -> NT*2*U 16-byte loads outstanding

// NT   : 16-row weight tiles per block (2 for SILU_MUL pairs / wide N, 1 for narrow N)
// XLDS : x staged in LDS (required for NORM); otherwise x fragments come from global (L2)
template <int NT, int EPI, bool XLDS, bool OUTF32>
__global__ void __launch_bounds__(256) gemv_kernel(const kr_bf16* __restrict__ x, int64_t ldx,
                                                   const kr_bf16* __restrict__ W, const kr_bf16* __restrict__ bias,
                                                   const kr_bf16* __restrict__ R, int64_t ldr, kr_bf16* __restrict__ out,
                                                   float* __restrict__ outf, int64_t ldc, int M, int N, int K,
                                                   const kr_bf16* __restrict__ norm_w, float norm_eps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int n_base = blockIdx.x * (NT * 16);
    const int nchunks = K >> 6;
    const int xrow_bytes = K * 2 + 16;  // +16: rows land on different LDS slots
    float* red = reinterpret_cast<float*>(smem + (XLDS ? ((M * xrow_bytes + 127) & ~127) : 0));  // [4][NT][64][4]
    float* rstd_s = red + 4 * NT * 256;  // [16]

    // ---- weight row pointers (rows past N are clamped; their results are never stored)
    const kr_bf16* wp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int n = n_base + t * 16 + fr;
        n = n < N ? n : N - 1;
        wp[t] = W + (int64_t)n * K + fg * 16;
    }
    // ---- put the first U chunks of weights in flight before touching x
    bf16x8 wbuf[GV_U][NT][2];
#pragma unroll
    for (int u = 0; u < GV_U; ++u) {
        const int c = wave + 4 * u;
        if (c < nchunks) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                wbuf[u][t][0] = ld8_nt(wp[t] + c * 64);
                wbuf[u][t][1] = ld8_nt(wp[t] + c * 64 + 8);
            }
        }
    }

    // ---- x: (optional RMSNorm) -> LDS
    if (XLDS) {
        const int kc = K >> 3;
        if (norm_w) {
            for (int b = wave; b < M; b += 4) {
                float ss = 0.f;
                for (int c = lane; c < kc; c += 64) {
                    const bf16x8 v = ld8(x + (int64_t)b * ldx + c * 8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) ss += bf2f(v[j]) * bf2f(v[j]);
                }
                ss = wave_sum(ss);
                if (lane == 0) rstd_s[b] = rsqrtf(ss / (float)K + norm_eps);
            }
            __syncthreads();
        }
        for (int i = tid; i < M * kc; i += 256) {
            const int b = i / kc, c = i - b * kc;
            bf16x8 v = ld8(x + (int64_t)b * ldx + c * 8);
            if (norm_w) {
                const bf16x8 nw = ld8(norm_w + c * 8);
                const float rs = rstd_s[b];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = f2bf(bf2f(nw[j]) * bfround(bf2f(v[j]) * rs));
            }
            *reinterpret_cast<bf16x8*>(smem + b * xrow_bytes + c * 16) = v;
        }
        __syncthreads();
    }
    const int xb = fr < M ? fr : 0;
    const char* xl = smem + xb * xrow_bytes + fg * 32;       // XLDS
    const kr_bf16* xg = x + (int64_t)xb * ldx + fg * 16;     // !XLDS

    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int c0 = wave; c0 < nchunks; c0 += 4 * GV_U) {
#pragma unroll
        for (int u = 0; u < GV_U; ++u) {
            const int c = c0 + 4 * u;
            if (c < nchunks) {
                bf16x8 x0, x1;
                if (XLDS) {
                    x0 = *reinterpret_cast<const bf16x8*>(xl + c * 128);
                    x1 = *reinterpret_cast<const bf16x8*>(xl + c * 128 + 16);
                } else {
                    x0 = ld8(xg + c * 64);
                    x1 = ld8(xg + c * 64 + 8);
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wbuf[u][t][0], x0, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wbuf[u][t][1], x1, acc[t], 0, 0, 0);
                }
                const int cn = c + 4 * GV_U;
                if (cn < nchunks) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        wbuf[u][t][0] = ld8_nt(wp[t] + cn * 64);
                        wbuf[u][t][1] = ld8_nt(wp[t] + cn * 64 + 8);
                    }
                }
            }
        }
    }

    // ---- cross-wave K reduction through LDS, then wave t finishes tile t
#pragma unroll
    for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4*>(red + ((wave * NT + t) * 64 + lane) * 4) = acc[t];
    __syncthreads();
    if (wave >= (EPI == KR_EPI_SILU_MUL ? 1 : NT)) return;
    f32x4 sum[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        sum[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const f32x4 p = *reinterpret_cast<const f32x4*>(red + ((w * NT + t) * 64 + lane) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) sum[t][j] += p[j];
        }
    }
    if (fr >= M) return;
    if (EPI == KR_EPI_SILU_MUL) {
        // tile 0 = gate rows, tile 1 = up rows of the same 16 features
        const int n = n_base + fg * 4;
        if (n >= N) return;
        const int oc = (n_base >> 1) + fg * 4;
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = f2bf(act_silu(sum[0][j]) * sum[NT - 1][j]);
        *reinterpret_cast<bf16x4*>(out + (int64_t)fr * ldc + oc) = o;
        return;
    }
    const int t = wave;  // wave-uniform tile index
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
        if (tt != t) continue;
        const int n = n_base + tt * 16 + fg * 4;
        if (n >= N) continue;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = sum[tt][j];
        if (bias) {
            const bf16x4 bv = *reinterpret_cast<const bf16x4*>(bias + n);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += bf2f(bv[j]);
        }
        if (EPI == KR_EPI_QUICK_GELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = act_quick_gelu(v[j]);
        } else if (EPI == KR_EPI_GELU_ERF) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = act_gelu_erf(v[j]);
        }
        if (R) {
            const bf16x4 rv = *reinterpret_cast<const bf16x4*>(R + (int64_t)fr * ldr + n);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += bf2f(rv[j]);
        }
        if (OUTF32) {
            *reinterpret_cast<f32x4*>(outf + (int64_t)fr * ldc + n) = (f32x4){v[0], v[1], v[2], v[3]};
        } else {
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = f2bf(v[j]);
            *reinterpret_cast<bf16x4*>(out + (int64_t)fr * ldc + n) = o;
        }
    }
}

template <int NT, int EPI, bool XLDS, bool OUTF32>
int launch_gemv(const kr_bf16* x, int64_t ldx, const kr_bf16* W, const kr_bf16* bias, const kr_bf16* R, int64_t ldr,
                kr_bf16* out, float* outf, int64_t ldc, int M, int N, int K, const kr_bf16* norm_w, float eps,
                kr_stream s) {
    const int grid = (N + NT * 16 - 1) / (NT * 16);
    const size_t xbytes = XLDS ? (((size_t)M * (K * 2 + 16) + 127) & ~(size_t)127) : 0;
    const size_t lds = xbytes + (size_t)4 * NT * 256 * 4 + 64;
    KR_CHECK_ARG(lds <= 160 * 1024, "kr_gemv_bf16: LDS %zu too large", lds);
    auto fn = &gemv_kernel<NT, EPI, XLDS, OUTF32>;
    static size_t max_set = 0;
    if (lds > 48 * 1024 && lds > max_set) {
        KR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         160 * 1024));
        max_set = 160 * 1024;
    }
    fn<<<grid, 256, lds, kr_hs(s)>>>(x, ldx, W, bias, R, ldr, out, outf, ldc, M, N, K, norm_w, eps);
    KR_CHECK_LAUNCH();
    return KR_OK;
}

}  // namespace

extern "C" int kr_gemv_bf16(const kr_bf16* x, int64_t ldx, const kr_bf16* W, const kr_bf16* bias,
                            const kr_bf16* residual, int64_t ldr, kr_bf16* out, float* out_f32, int64_t ldc, int M,
                            int N, int K, int epilogue, const kr_bf16* norm_w, float norm_eps, kr_stream s) {
    KR_CHECK_ARG(x && W && (out || out_f32), "kr_gemv_bf16: null pointer");
    KR_CHECK_ARG(M >= 1 && M <= 16, "kr_gemv_bf16: M=%d must be in 1..16", M);
    KR_CHECK_ARG(N > 0 && N % 16 == 0, "kr_gemv_bf16: N=%d must be a multiple of 16", N);
    KR_CHECK_ARG(K > 0 && K % 64 == 0, "kr_gemv_bf16: K=%d must be a multiple of 64", K);
    KR_CHECK_ARG(ldx >= K && (ldx & 7) == 0 && (ldc & 3) == 0, "kr_gemv_bf16: ldx/ldc");
    KR_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)W & 15) == 0, "kr_gemv_bf16: pointer alignment");
    const bool xlds = (size_t)M * (K * 2 + 16) <= 148 * 1024;
    KR_CHECK_ARG(!norm_w || xlds, "kr_gemv_bf16: fused RMSNorm needs M*(2K+16) <= 148 KiB of LDS (M=%d K=%d)", M, K);
    if (epilogue == KR_EPI_SILU_MUL) {
        KR_CHECK_ARG(N % 32 == 0 && !bias && !residual && out && !out_f32 && ldc >= N / 2, "kr_gemv_bf16: SILU_MUL args");
        return xlds ? launch_gemv<2, KR_EPI_SILU_MUL, true, false>(x, ldx, W, bias, residual, ldr, out, out_f32, ldc, M, N,
                                                                   K, norm_w, norm_eps, s)
                    : launch_gemv<2, KR_EPI_SILU_MUL, false, false>(x, ldx, W, bias, residual, ldr, out, out_f32, ldc, M,
                                                                    N, K, norm_w, norm_eps, s);
    }
    KR_CHECK_ARG(epilogue == KR_EPI_NONE, "kr_gemv_bf16: epilogue %d not supported for M<=16", epilogue);
    KR_CHECK_ARG(ldc >= N, "kr_gemv_bf16: ldc < N");
    // wide N: two tiles per block share the x fragments; narrow N: one tile per block for parallelism
    const bool wide = N >= 16 * 2 * 512;
    if (out_f32) {
        KR_CHECK_ARG(!residual, "kr_gemv_bf16: fp32 output has no residual path");
        if (wide)
            return xlds ? launch_gemv<2, KR_EPI_NONE, true, true>(x, ldx, W, bias, residual, ldr, out, out_f32, ldc, M, N, K,
                                                                  norm_w, norm_eps, s)
                        : launch_gemv<2, KR_EPI_NONE, false, true>(x, ldx, W, bias, residual, ldr, out, out_f32, ldc, M, N,
                                                                   K, norm_w, norm_eps, s);
        return xlds ? launch_gemv<1, KR_EPI_NONE, true, true>(x, ldx, W, bias, residual, ldr, out, out_f32, ldc, M, N, K,
                                                              norm_w, norm_eps, s)
                    : launch_gemv<1, KR_EPI_NONE, false, true>(x, ldx, W, bias, residual, ldr, out, out_f32, ldc, M, N, K,
                                                               norm_w, norm_eps, s);
    }
    if (wide)
        return xlds ? launch_gemv<2, KR_EPI_NONE, true, false>(x, ldx, W, bias, residual, ldr, out, out_f32, ldc, M, N, K,
                                                               norm_w, norm_eps, s)
                    : launch_gemv<2, KR_EPI_NONE, false, false>(x, ldx, W, bias, residual, ldr, out, out_f32, ldc, M, N, K,
                                                                norm_w, norm_eps, s);
    return xlds ? launch_gemv<1, KR_EPI_NONE, true, false>(x, ldx, W, bias, residual, ldr, out, out_f32, ldc, M, N, K,
                                                           norm_w, norm_eps, s)
                : launch_gemv<1, KR_EPI_NONE, false, false>(x, ldx, W, bias, residual, ldr, out, out_f32, ldc, M, N, K,
                                                            norm_w, norm_eps, s);
}
