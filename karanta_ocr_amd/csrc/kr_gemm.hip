// Dense linears on the MI355X matrix cores.
//
//  gemm_kernel      : C[M,N] = epi(A[M,K] W[N,K]^T + bias) (+R)   M large (ViT, merger, decoder prefill)
//                     128x128x64 block tile, 4 waves (2x2, 64x64 each) — small / ragged M, 2-3 workgroups per CU —
//                     or 256x256x64, 8 waves (kept for A/B runs); v_mfma_f32_16x16x32_bf16, both operands
//                     K-contiguous, staged HBM->LDS with 16-byte LDS-DMA (global_load_lds_dwordx4) into an
//                     XOR-swizzled image (swizzle on the per-lane SOURCE address, linear LDS destination), double
//                     buffered, one drained barrier pair per K-step.
//  gemm_pipe_kernel : the 256x256 tile for every GEMM with >= 192 tiles: four K = 32 LDS buffers, counted vmcnt
//                     (staging in flight across barriers), staggered wave rows; bf16 W (row-major or decode layout)
//                     or fp8 W codes + row scales (kr_gemm_fp8).  Epilogue through LDS: full-line stores.
//                     MFMA-bound; roofline = 2.5 PFLOP/s dense bf16 (the chip clocks near 1.5 GHz at this duty).
//  gemv_kernel      : same contract for M <= 16 (decode).  Weights go HBM -> VGPR exactly once with
//                     non-temporal 16-byte loads, 8+ loads in flight per wave; x (<= 16 rows) is staged
//                     (optionally RMS-normalised) in LDS.  HBM-bound; roofline = 8 TB/s.
#include <algorithm>
#include <mutex>

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

// Epilogue shared by the GEMM kernels: lane (fr, fg) of a wave holds, per (nt, mt) accumulator, 4 consecutive n
// (n_base + nt*16 + fg*4 ..) of one m (m_base + mt*16 + fr).
// wlds != nullptr: the wave's own MT*16 x 128-byte LDS scratch (the staging buffers are free by now).  The bf16
// results go through it and leave as 16-byte-per-lane, row-contiguous stores (whole 128-byte lines per 8 lanes)
// instead of 8-byte pieces of 16 different rows per instruction: the direct form wrote C at ~2 TB/s with one
// workgroup per CU and nothing to overlap it with (vit fc1: 233 of 642 us).  Values are identical either way.
// LDSEPI / HAS_R are compile-time: runtime selects inside the unrolled (mt, nt) loops turn into a branch per accumulator.
// CH > 0 (LDSEPI only): the scratch holds CH m-tiles (CH * 16 rows) at a time and is flushed after every CH of them — the persistent
// kernel's epilogue, whose scratch is the 4 KiB per wave the staging ring leaves over.  Same values, same stores.
template <int EPI, int NT, int MT, bool LDSEPI, bool HAS_R, int CH = 0>
__device__ __forceinline__ void gemm_epilogue_impl(f32x4 (&acc)[NT][MT], const kr_bf16* __restrict__ bias,
                                                   const kr_bf16* __restrict__ R, int64_t ldr, kr_bf16* __restrict__ C, int64_t ldc,
                                                   int64_t M, int N, int64_t m_base, int n_base, int fr, int fg, char* wlds_) {
    constexpr bool HALF = EPI == KR_EPI_SILU_MUL8;       // output is N / 2 wide: 64-byte scratch rows
    constexpr int ROWB = HALF ? 64 : 128;
    char* const wlds = wlds_;
    // scratch chunk (16 B) c of row r sits at c ^ swz(r): 2-way conflicts on the 8-byte writes, none on the reads
    auto swz = [](int r) { return HALF ? ((r >> 1) & 3) : (r & 7); };
    // scratch rows [0, nmt * 16) = the m-tiles mt0 .. mt0 + nmt - 1: out as 16-byte-per-lane, row-contiguous stores.  Same wave,
    // in-order LDS: its reads see its writes, and the next m-tiles' writes come after these reads
    auto flush = [&](int mt0, int nmt) {
        constexpr int LPR = ROWB / 16, RPI = 64 / LPR;   // lanes per row, rows per instruction
        const int lane = fg * 16 + fr;
        const int n_out = HALF ? (n_base >> 1) : n_base, n_lim = HALF ? (N >> 1) : N;
#pragma unroll
        for (int it = 0; it < MT * 16 / RPI; ++it) {
            if (it < nmt * 16 / RPI) {
                const int r = it * RPI + lane / LPR, c = lane % LPR;
                const u32x4 v = *reinterpret_cast<const u32x4*>(wlds + r * ROWB + ((c ^ swz(r)) << 4));
                const int64_t m = m_base + mt0 * 16 + r;
                const int n = n_out + c * 8;
                if (m < M && n < n_lim) *reinterpret_cast<u32x4*>(C + m * ldc + n) = v;
            }
        }
    };
    constexpr int RM = CH > 0 ? CH : MT;   // m-tiles the scratch holds
    // ---------------- epilogue: lane holds 4 consecutive n for one m
    // Operand loads are unconditional (clamped addresses) and hoisted: a guarded load inside the (mt, nt) loops costs
    // one dependent L2 round trip per accumulator — 32 of them in a row were 8-13 us per tile.
    if (EPI == KR_EPI_SILU_MUL8) {
        // gate/up interleaved in groups of 8 rows: a 16-row tile holds gate rows in lane groups 0,1
        // and the matching up rows in lane groups 2,3 (lane ^ 32)
        bf16x4 bv[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {   // bias rows are interleaved like the weight rows: every lane adds its own
            const int n = min(n_base + nt * 16, N - 16);
            bv[nt] = bias ? *reinterpret_cast<const bf16x4*>(bias + n + fg * 4) : bf16x4{};
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int64_t m = m_base + mt * 16 + fr;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = n_base + nt * 16;
                float g[4], u[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] = acc[nt][mt][j] + bf2f(bv[nt][j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) u[j] = __shfl_xor(g[j], 32, 64);
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = f2bf(act_silu(g[j]) * u[j]);
                if (LDSEPI) {
                    const int r = (mt % RM) * 16 + fr;
                    if (fg < 2) *reinterpret_cast<bf16x4*>(wlds + r * ROWB + ((nt ^ swz(r)) << 4) + (fg & 1) * 8) = o;
                } else if (fg < 2 && n < N && m < M) {
                    *reinterpret_cast<bf16x4*>(C + m * ldc + (n >> 1) + fg * 4) = o;
                }
            }
            if (LDSEPI && CH > 0 && (mt % RM) == RM - 1) flush(mt - (RM - 1), RM);
        }
    } else if (EPI == KR_EPI_SILU_MUL) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int64_t m = m_base + mt * 16 + fr;
            if (m >= M) continue;
#pragma unroll
            for (int pr = 0; pr < NT / 2; ++pr) {
                const int n = n_base + pr * 32 + fg * 4;  // gate row index in W'
                if (n >= N) continue;
                const int oc = ((n_base) >> 1) + pr * 16 + fg * 4;
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = f2bf(act_silu(acc[2 * pr][mt][j]) * acc[2 * pr + 1][mt][j]);
                *reinterpret_cast<bf16x4*>(C + m * ldc + oc) = o;
            }
        }
    } else {
        bf16x4 bv[NT];
        int ncl[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            ncl[nt] = min(n_base + nt * 16 + fg * 4, N - 4);
            bv[nt] = bias ? *reinterpret_cast<const bf16x4*>(bias + ncl[nt]) : bf16x4{};
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int64_t m = m_base + mt * 16 + fr;
            const int64_t mc = m < M ? m : M - 1;
            bf16x4 rv[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) rv[nt] = HAS_R ? *reinterpret_cast<const bf16x4*>(R + mc * ldr + ncl[nt]) : bf16x4{};
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = n_base + nt * 16 + fg * 4;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[nt][mt][j];
                if (bias) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += bf2f(bv[nt][j]);
                }
                if (EPI == KR_EPI_QUICK_GELU) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = act_quick_gelu(v[j]);
                } else if (EPI == KR_EPI_GELU_ERF) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = act_gelu_erf(v[j]);
                }
                if (HAS_R) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += bf2f(rv[nt][j]);
                }
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = f2bf(v[j]);
                if (LDSEPI) {
                    const int r = (mt % RM) * 16 + fr;
                    *reinterpret_cast<bf16x4*>(wlds + r * ROWB + (((nt * 2 + (fg >> 1)) ^ swz(r)) << 4) + (fg & 1) * 8) = o;
                } else if (m < M && n < N) {
                    *reinterpret_cast<bf16x4*>(C + m * ldc + n) = o;
                }
            }
            if (LDSEPI && CH > 0 && (mt % RM) == RM - 1) flush(mt - (RM - 1), RM);
        }
    }
    if (LDSEPI && CH == 0) flush(0, MT);
}

// LDSEPI kernels (the 256x256 tiles) are only launched with ldc % 8 == 0 and a 16-byte aligned C.
template <int EPI, int NT, int MT, bool LDSEPI, int CH = 0>
__device__ __forceinline__ void gemm_epilogue(f32x4 (&acc)[NT][MT], const kr_bf16* __restrict__ bias,
                                              const kr_bf16* __restrict__ R, int64_t ldr, kr_bf16* __restrict__ C, int64_t ldc,
                                              int64_t M, int N, int64_t m_base, int n_base, int fr, int fg, char* wlds) {
    constexpr bool L = LDSEPI && EPI != KR_EPI_SILU_MUL;
    if constexpr (EPI == KR_EPI_SILU_MUL || EPI == KR_EPI_SILU_MUL8) {
        gemm_epilogue_impl<EPI, NT, MT, L, false, CH>(acc, bias, R, ldr, C, ldc, M, N, m_base, n_base, fr, fg, wlds);
    } else {
        if (R) gemm_epilogue_impl<EPI, NT, MT, L, true, CH>(acc, bias, R, ldr, C, ldc, M, N, m_base, n_base, fr, fg, wlds);
        else gemm_epilogue_impl<EPI, NT, MT, L, false, CH>(acc, bias, R, ldr, C, ldc, M, N, m_base, n_base, fr, fg, wlds);
    }
}

// Tile id -> (m tile, n tile) of the 256x256 tile list.  group_m <= 1: m-major rows.  group_m = g: the list walks DOWN g
// m tiles before it moves to the next n tile (groups of g x tiles_n tiles), so the ~32 workgroups that run side by side on
// one XCD (a contiguous run of ids, xcd_remap) cover g x 32/g tiles and share g A blocks + 32/g W blocks in that XCD's L2
// instead of 1 + 32 (prefill gate/up, 70 n tiles: every m row re-streamed all of W through the 4 MB L2).
__device__ __forceinline__ void tile_coords(unsigned id, unsigned tiles_m, unsigned tiles_n, unsigned group_m, unsigned& tm,
                                            unsigned& tn) {
    if (group_m <= 1) {
        tm = id / tiles_n;
        tn = id - tm * tiles_n;
        return;
    }
    const unsigned per_group = group_m * tiles_n;
    const unsigned g = id / per_group, first = g * group_m;
    const unsigned gsz = min(tiles_m - first, group_m);
    const unsigned r = id - g * per_group;
    tn = r / gsz;
    tm = first + (r - tn * gsz);
}

// STAGES = 2: the tile of K step k+1 is requested at the top of step k and drained (vmcnt(0)) at its end — fine with
// several workgroups per CU covering for each other.  STAGES = 4 (launches of at most one workgroup per CU: the tail
// quarters of a 256-tile launch, small GEMMs of a one-page admission): three K steps in flight behind a COUNTED wait and
// raw barriers, because a lone 4-wave workgroup on a CU has nobody to hide a memory round trip per K step behind (the 8
// tail tiles of prefill down_proj, K = 8960: 32 workgroups x 140 steps x 1.2 us = 166 us for 3 % of the GEMM's work).
template <int EPI, bool WPACK, typename G, int STAGES = 2>
__global__ void __launch_bounds__(G::WM * G::WN * 64) gemm_kernel(const kr_bf16* __restrict__ A, int64_t lda,
                                                   const kr_bf16* __restrict__ W, const kr_bf16* __restrict__ bias,
                                                   const kr_bf16* __restrict__ R, int64_t ldr, kr_bf16* __restrict__ C,
                                                   int64_t ldc, int64_t M, int N, int K, int tiles_n, unsigned nwg,
                                                   int ptiles_n, unsigned qbase, int group_m, int ksplit, int ks_mode,
                                                   float* __restrict__ ws) {
    constexpr int BM = G::BM, BN = G::BN, NTHR = G::WM * G::WN * 64;
    constexpr int WTM = BM / G::WM, WTN = BN / G::WN;  // wave tile
    constexpr int MT = WTM / 16, NT = WTN / 16;
    constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2, STAGE = A_BYTES + W_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 x (A tile, W tile)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / G::WN, wc = wave % G::WN;

    unsigned wg = xcd_remap(blockIdx.x, nwg);
    int64_t m0 = (int64_t)(wg / tiles_n) * BM;
    int n0 = (int)(wg % tiles_n) * BN;
    // Tail with a K split (ks_mode): TWO launches, the kernel boundary between them is the hand-off.
    //   ks_mode 1: grid = quarters x ksplit; this workgroup walks its share of the K steps and leaves its f32 accumulators in
    //              the workspace, image [quarter][split][thread][NT * MT] f32x4 (thread-private: same fragment map everywhere);
    //   ks_mode 2: grid = quarters; sums the ksplit partials in split order and runs the epilogue.  No K loop.
    // A lone 128x128 workgroup streams its 32 KB per K step at the ~45 GB/s ONE CU pulls from HBM (measured: 0.82 us per
    // step with three steps in flight), so a long-K tail wants its bytes on many CUs, not a deeper ring.
    unsigned qid = 0, sidx = 0;
    if (ptiles_n > 0) {
        // TAIL of a 256x256-tile launch (launch_gemm_pipe): this grid is the quarters (128x128) of the 256x256 tiles
        // qbase, qbase + 1, ... of that launch's tile list — the tiles of its last, mostly empty round
        if (ks_mode == 1) {
            wg = blockIdx.x;
            qid = wg / (unsigned)ksplit;
            sidx = wg - qid * (unsigned)ksplit;
            wg = qid;
        } else {
            qid = wg;
        }
        const unsigned parent = qbase + (wg >> 2), sub = wg & 3u;
        unsigned tm, tn;
        tile_coords(parent, (unsigned)((M + 255) >> 8), (unsigned)ptiles_n, (unsigned)group_m, tm, tn);
        m0 = (int64_t)tm * 256 + (sub >> 1) * 128;
        n0 = (int)tn * 256 + (int)(sub & 1u) * 128;
        if (m0 >= M) return;  // the parent straddled the end of M (whole workgroup: before any barrier)
    }

    f32x4 acc[NT][MT];  // [nt][mt]: rows (regs) = n, col (lane&15) = m
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int kt0 = 0, nk = K / BK;
    if (ks_mode == 1) {  // this workgroup's K steps: an even share, the remainder to the first splits
        const int q = nk / ksplit, r = nk - q * ksplit;
        kt0 = (int)sidx * q + min((int)sidx, r);
        nk = kt0 + q + ((int)sidx < r ? 1 : 0);
    }
    const int fr = lane & 15, fg = lane >> 4;
    constexpr int FRAGS = NT * MT;
    if (ks_mode == 2) {
        for (int sp = 0; sp < ksplit; ++sp) {
            const f32x4* part = reinterpret_cast<const f32x4*>(ws) + (((size_t)qid * ksplit + sp) * NTHR + tid) * FRAGS;
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    const f32x4 pv = part[i * MT + j];
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[i][j][c] += pv[c];
                }
        }
        nk = 0;   // straight to the epilogue
    }
    constexpr int LPT = (BM + BN) * 8 / NTHR;   // LDS-DMA instructions per thread and K step
    auto stage = [&](int kt, int buf) {
        stage_tile<false, BM, NTHR>(A, lda, m0, M, kt * BK, smem + buf * STAGE, tid, wave);
        stage_tile<WPACK, BN, NTHR>(W, K, n0, N, kt * BK, smem + buf * STAGE + A_BYTES, tid, wave);
    };
    if constexpr (STAGES == 2) {
        if (kt0 < nk) {
            stage(kt0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    } else {
        static_assert(STAGES == 4, "ring depth");
#pragma unroll
        for (int p = 0; p < STAGES - 1; ++p)
            if (kt0 + p < nk) stage(kt0 + p, p);
    }

    for (int kt = kt0; kt < nk; ++kt) {
        const int ki = kt - kt0;
        char* cur = smem + (ki % STAGES) * STAGE;
        if constexpr (STAGES == 2) {
            if (kt + 1 < nk) stage(kt + 1, (ki + 1) & 1);
        } else {
            // buffer (ki + 3) % 4 = (ki - 1) % 4 was read in step ki - 1, which every wave left at its trailing barrier
            if (kt + STAGES - 1 < nk) stage(kt + STAGES - 1, (ki + STAGES - 1) % STAGES);
            // K step kt has landed once at most the steps requested after it are outstanding (loads retire in order)
            const int later = min(STAGES - 1, nk - 1 - kt);
            if (later == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LPT) : "memory");
            else if (later == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");
            else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * LPT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();   // raw: a __syncthreads() would drain the staging
            asm volatile("" ::: "memory");
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
        if constexpr (STAGES == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        } else {
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();   // every wave has read this step's buffer: the next step may restage its ring slot
            asm volatile("" ::: "memory");
        }
    }

    if (ks_mode == 1) {
        f32x4* part = reinterpret_cast<f32x4*>(ws) + (((size_t)qid * ksplit + sidx) * NTHR + tid) * FRAGS;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) part[i * MT + j] = acc[i][j];
        return;
    }
    gemm_epilogue<EPI, NT, MT, G::BM == 256>(acc, bias, R, ldr, C, ldc, M, N, m0 + wr * WTM, n0 + wc * WTN, fr, fg,
                                             smem + wave * (WTM * 128));
}

// -------------------------------------------------------------------------------------
// gemm_pipe_kernel: the 256x256 tile with the staging kept in flight ACROSS barriers.
// gemm_kernel drains its LDS-DMA (vmcnt(0)) at every K-step's barrier: one 64 KiB stage is requested at the top of
// a step and must have landed at its end, so the bytes in flight saw-tooth between 64 KiB and nothing and the MFMA
// pipe waits on memory latency (the "two barriers per K-step" ceiling, ~900 TFLOP/s measured here).  This kernel:
//   * K-tiles of 32 (one MFMA k-step), FOUR LDS buffers of [A 256x32][W 256x32] = 32 KiB each (128 KiB);
//   * two phases per K-tile, each {fragment ds_reads + 2 LDS-DMA per wave -> barrier -> 16 MFMA -> barrier}:
//     phase 0 reads W (4 n-tiles) + A rows 0..63 of the wave and stages the W half-tiles of K-tile k+3,
//     phase 1 reads A rows 64..127 and stages the A half-tiles of K-tile k+3;
//   * a COUNTED wait once per K-tile (vmcnt(8): K-tile k+1 landed, k+2 and k+3 still in flight), never 0 in the
//     steady state; raw s_barrier (a __syncthreads would drain the DMA);
//   * the two wave rows (wr = 0 / 1) run one barrier apart, so one half of the workgroup issues loads while the
//     other half is in its MFMA burst (s_setprio(1) around the burst).
// Hazards (DMA is ordered for a ds_read only by the issuing wave's vmcnt + a barrier the reader has passed):
//   RAW  K-tile k+1 is waited for in phase (k,1) before its first barrier and first read in phase (k+1,0);
//   WAR  buffer (k+3)&3 = (k-1)&3: its W rows were last read in phase (k-1,0), its A rows in phase (k-1,1);
//        they are restaged in phases (k,0) / (k,1), two phases (four barriers) later — one barrier of stagger fits.
// LDS image: 64-byte rows (32 k); chunk c (8 k) of row r sits at slot c ^ (((r >> 3) & 1) << 1): the four 16-lane
// groups of a ds_read_b128 ({0-3,12-15,20-27}, ...) then touch 16 distinct 16-byte slots of the 256-byte bank row.
constexpr int PK = 32, PBUF = 2 * 256 * PK * 2, PSTAGES = 4;

template <bool PACKED>
__device__ __forceinline__ void stage_rows32(const kr_bf16* __restrict__ g, int64_t ld, int64_t row0, int64_t rows_total, int k0,
                                             char* lds_op, int lane, int wave) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {  // the two 128-row halves: each wave fills one 16-row block (1 KiB) of each
        const int r = h * 128 + wave * 16 + (lane >> 2);
        const int c = (lane & 3) ^ (((lane >> 5) & 1) << 1);  // source chunk for this LDS slot
        int64_t gr = row0 + r;
        gr = gr < rows_total ? gr : rows_total - 1;
        const kr_bf16* src = PACKED ? g + ((((gr >> 4) * (ld >> 5) + (k0 >> 5)) * 4 + c) * 16 + (gr & 15)) * 8
                                    : g + gr * ld + k0 + c * 8;
        char* dst = lds_op + (h * 128 + wave * 16) * 64;  // wave-uniform; the hardware adds lane * 16
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
}

// fp8 weights in the decode layout [N/16][K/64][4 k-groups][16 rows][16 B] (weights.pack_w16x64_fp8): the 32 k of a
// K-tile are half a 1 KiB block (k-groups 2j, 2j+1 = 512 contiguous bytes), so a 256-row W tile is 8 KiB and ONE
// LDS-DMA per wave (lanes 0-31: row block 2*wave, lanes 32-63: row block 2*wave + 1).
__device__ __forceinline__ void stage_w8_rows32(const uint8_t* __restrict__ g, int K, int n0, int N, int k0, char* lds_op, int lane,
                                                int wave) {
    int rb = (n0 >> 4) + 2 * wave + (lane >> 5);
    rb = min(rb, (N >> 4) - 1);
    const uint8_t* src = g + ((int64_t)rb * (K >> 6) + (k0 >> 6)) * 1024 + ((k0 & 32) ? 512 : 0) + (lane & 31) * 16;
    char* dst = lds_op + wave * 1024;  // wave-uniform; the hardware adds lane * 16
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

// fp8 ACTIVATIONS (W8A8 prefill: kr_quantize_rows_fp8 + kr_gemm_fp8a): row-major e4m3 codes [M][K], so the 32 k of a K-tile are
// 32 bytes of a row and a 256-row A tile is 8 KiB — ONE LDS-DMA per wave (32 rows x two 16-byte halves).  LDS image: 32-byte
// rows; the 16-byte half q of row r holds SOURCE half q ^ ((r >> 3) & 1), so that the 32 lanes of a ds_read_b64 group
// (rows 0..15 x two 8-byte granules) touch 32 distinct 8-byte slots of the 256-byte bank row.
__device__ __forceinline__ void stage_a8_rows32(const uint8_t* __restrict__ g, int64_t ld, int64_t row0, int64_t rows_total, int k0,
                                                char* lds_op, int lane, int wave) {
    const int r = wave * 32 + (lane >> 1);
    const int p = (lane & 1) ^ ((r >> 3) & 1);
    int64_t gr = row0 + r;
    gr = gr < rows_total ? gr : rows_total - 1;
    const uint8_t* src = g + gr * ld + k0 + p * 16;
    char* dst = lds_op + wave * 1024;  // wave-uniform; the hardware adds lane * 16
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
}

// 8 e4m3 codes (two 32-bit words) -> 8 bf16, natural order (cvt_scalef32_pk_bf16_fp8 converts one 16-bit half of a
// word; its 2 x bf16 result is moved as a 32-bit word: element-wise extraction is mis-lowered by this compiler)
__device__ __forceinline__ bf16x8 fp8x8_to_bf16(u32x2 q) {
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        o[2 * i + 0] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)q[i], 1.0f, false));
        o[2 * i + 1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)q[i], 1.0f, true));
    }
    return __builtin_bit_cast(bf16x8, o);
}

// W8: W is fp8 codes + one f32 scale per output row (applied to the accumulators before the epilogue); activations,
// accumulation and outputs as in the bf16 kernel.
// A8 (with W8): the activations are e4m3 codes too (one f32 scale per ROW of A: dynamic per-token quantisation,
// kr_quantize_rows_fp8) and the products run on the fp8 matrix instruction, v_mfma_f32_16x16x32_fp8_fp8 — no conversion on
// the way from LDS to the MFMA, half the A bytes through LDS; both scales multiply the f32 accumulators before the epilogue.
// PERSIST (r3 EXPERIMENT, -DKR_GEMM_PERSIST_EXPERIMENT): the workgroup walks the tile list (tiles blockIdx.x,
// + gridDim.x, ...: the same tile every round as the one-tile-per-workgroup launch gives that CU) and requests the NEXT tile's first
// three K-tiles before it runs the current tile's epilogue, whose LDS scratch is the 4 KiB per wave the ring leaves over (two m-tiles
// at a time): the pipeline fill of a tile and the workgroup turnover sit under the previous tile's C write-out instead of in front
// of every tile.  Same values as the product launch (GPU test on the variant library).  MEASURED (tools/gemm_microbench.py,
// profiles/r03_gemm_persistent.txt): ViT qkv 355.9 us against 345.0, fc1 487.9 / 483.5, proj 122.2 / 121.1, fc2 411.3 / 407.8, prefill
// gate/up 540.5 / 540.4 — no gain: with one workgroup per CU the hardware already starts the next workgroup while the finished
// one's C stores drain, whereas the loop has to wait for them (vmcnt(0): loads and stores share the counter) before it may trust
// the prefetched K-tiles.  What the K = 1280 shapes lose is inside the loop (0.0254 us per k and round against 0.0161 at the MFMA
// rate) and in the epilogue itself, not in the turnover.
template <int EPI, bool WPACK, bool W8, bool A8 = false, bool PERSIST = false>
__global__ void __launch_bounds__(512) gemm_pipe_kernel(const kr_bf16* __restrict__ A, int64_t lda, const kr_bf16* __restrict__ W,
                                                        const kr_bf16* __restrict__ bias, const kr_bf16* __restrict__ R,
                                                        int64_t ldr, kr_bf16* __restrict__ C, int64_t ldc, int64_t M, int N, int K,
                                                        int tiles_n, unsigned nwg, const float* __restrict__ w_scale, int group_m,
                                                        const float* __restrict__ a_scale = nullptr, int stagger = 0) {
    static_assert(!A8 || W8, "fp8 activations go with fp8 weights");
    // STAGGER (KARANTA_GEMM_STAGGER = n, experiment): every second workgroup of the FIRST round (one per CU) starts n x ~3.4 us
    // late, so that for the rest of the launch half of the CUs are in their main loop while the other half run their
    // prologue / epilogue bursts (all tiles take the same time: without it every round's 33 MB of C stores and its
    // pipeline fills hit the memory system together)
    if (stagger > 0 && blockIdx.x < 256u && ((blockIdx.x >> 3) & 1u)) {
        for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }
    constexpr int NT = 4, MT = 8, A_BYTES = 256 * PK * 2;
    const uint8_t* W8p = reinterpret_cast<const uint8_t*>(W);
    const uint8_t* A8p = reinterpret_cast<const uint8_t*>(A);
    extern __shared__ __attribute__((aligned(16))) char smem[];  // the ONLY LDS object (a second one makes hipcc drain the DMA)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fg = lane >> 4;
    // tile `t` of the list: PERSIST: round r = t / gridDim.x takes ids [r * G, r * G + G_r) with the XCD remap inside the round
    int64_t m0;
    int n0;
    auto set_tile = [&](unsigned t) {
        unsigned id;
        if constexpr (PERSIST) {
            const unsigned r0 = (t / gridDim.x) * gridDim.x;
            id = r0 + xcd_remap(t - r0, min(gridDim.x, nwg - r0));
        } else {
            id = xcd_remap(t, nwg);
        }
        unsigned tm, tn;
        tile_coords(id, (unsigned)((M + 255) >> 8), (unsigned)tiles_n, (unsigned)group_m, tm, tn);
        m0 = (int64_t)tm * 256;
        n0 = (int)tn * 256;
    };
    unsigned tile = blockIdx.x;
    set_tile(tile);

    f32x4 acc[NT][MT];

    const int nk = K / PK;
    // fragment read offsets inside a buffer (row * 64 + swizzled chunk * 16)
    const int sw = ((fg ^ (((fr >> 3) & 1) << 1)) << 4);
    const int a_off = A8 ? (wr * 128 + fr) * 32 + ((((fg >> 1) ^ ((fr >> 3) & 1))) << 4) + ((fg & 1) << 3)   // 8 codes of row fr
                         : (wr * 128 + fr) * 64 + sw;
    const int w_off = W8 ? A_BYTES + wc * 4 * 512 + (fg >> 1) * 256 + fr * 16 + (fg & 1) * 8   // 8 codes of row fr
                         : A_BYTES + (wc * 64 + fr) * 64 + sw;
    auto stage_w = [&](int k0, char* buf) {
        if constexpr (W8) stage_w8_rows32(W8p, K, n0, N, k0, buf + A_BYTES, lane, wave);
        else stage_rows32<WPACK>(W, K, n0, N, k0, buf + A_BYTES, lane, wave);
    };
    auto stage_a = [&](int k0, char* buf) {
        if constexpr (A8) stage_a8_rows32(A8p, lda, m0, M, k0, buf, lane, wave);
        else stage_rows32<false>(A, lda, m0, M, k0, buf, lane, wave);
    };

    // ---- prologue: K-tiles 0..2 requested, K-tile 0 landed and visible
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        if (t < nk) {
            char* buf = smem + t * PBUF;
            stage_w(t * PK, buf);
            stage_a(t * PK, buf);
        }
    }
    // loads per K-tile and wave: 2 (bf16 A) or 1 (fp8 A) + 2 (bf16 W) or 1 (fp8 W); the counted waits leave two / one K-tile in flight
    auto wait_two = [] {
        if constexpr (A8) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if constexpr (W8) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    };
    auto wait_one = [] {
        if constexpr (A8) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if constexpr (W8) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    };
    if (nk >= 3) wait_two();
    else if (nk == 2) wait_one();
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    bf16x8 wb[NT], xa[4];
    u32x2 xq[4];   // A8: the A fragments as 8 codes
    for (;;) {   // one pass per tile (PERSIST: until the list ends)
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (wr == 1) __builtin_amdgcn_s_barrier();  // stagger: wave row 1 runs one barrier behind wave row 0
    for (int k = 0; k < nk; ++k) {
        const char* cur = smem + (k & 3) * PBUF;
        char* nxt = smem + ((k + 3) & 3) * PBUF;
        const bool more = k + 3 < nk;
        // ---------------- phase 0: W fragments + A rows 0..63
        u32x2 wq[NT];
        if constexpr (W8) {
#pragma unroll
            for (int t = 0; t < NT; ++t) wq[t] = *reinterpret_cast<const u32x2*>(cur + w_off + t * 512);
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) wb[t] = *reinterpret_cast<const bf16x8*>(cur + w_off + t * 16 * 64);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (A8) {
#pragma unroll
            for (int t = 0; t < 4; ++t) xq[t] = *reinterpret_cast<const u32x2*>(cur + a_off + t * 16 * 32);
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) xa[t] = *reinterpret_cast<const bf16x8*>(cur + a_off + t * 16 * 64);
        }
        if (more) stage_w((k + 3) * PK, nxt);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (W8 && !A8) {
#pragma unroll
            for (int t = 0; t < NT; ++t) wb[t] = fp8x8_to_bf16(wq[t]);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if constexpr (A8)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(__builtin_bit_cast(long, wq[nt]), __builtin_bit_cast(long, xq[mt]),
                                                                             acc[nt][mt], 0, 0, 0);
                else
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[nt], xa[mt], acc[nt][mt], 0, 0, 0);
            }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
        // ---------------- phase 1: A rows 64..127; the counted wait of this K-tile
        if constexpr (A8) {
#pragma unroll
            for (int t = 0; t < 4; ++t) xq[t] = *reinterpret_cast<const u32x2*>(cur + a_off + (4 + t) * 16 * 32);
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) xa[t] = *reinterpret_cast<const bf16x8*>(cur + a_off + (4 + t) * 16 * 64);
        }
        if (more) stage_a((k + 3) * PK, nxt);
        // K-tile k+1 must have landed (this wave's share); k+2 and k+3 may stay in flight
        if (more) wait_two();
        else if (k + 3 == nk) wait_one();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if constexpr (A8)
                    acc[nt][4 + mt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(__builtin_bit_cast(long, wq[nt]), __builtin_bit_cast(long, xq[mt]),
                                                                                 acc[nt][4 + mt], 0, 0, 0);
                else
                    acc[nt][4 + mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[nt], xa[mt], acc[nt][4 + mt], 0, 0, 0);
            }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();  // pairs with wave row 1's last barrier
    const int64_t m0_cur = m0;
    const int n0_cur = n0;
    bool more_tiles = false;
    if constexpr (PERSIST) {
        // every fragment read of this tile is behind the last barrier: the ring is free.  The next tile's K-tiles 0..2 are
        // requested now and land under the epilogue below
        tile += gridDim.x;
        more_tiles = tile < nwg;
        if (more_tiles) {
            set_tile(tile);
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                if (t < nk) {
                    char* buf = smem + t * PBUF;
                    stage_w(t * PK, buf);
                    stage_a(t * PK, buf);
                }
            }
        }
    }
    if constexpr (W8) {  // row scales of the quantised weights: lane holds 4 consecutive n per accumulator
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(w_scale + min(n0_cur + wc * 64 + nt * 16 + fg * 4, N - 4));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[nt][mt][j] *= sc[j];
        }
    }
    if constexpr (A8) {  // per-token scales of the quantised activations: the lane's batch row is the accumulator's column
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int64_t m = m0_cur + wr * 128 + mt * 16 + fr;
            const float as = a_scale[m < M ? m : M - 1];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[nt][mt][j] *= as;
        }
    }
    if constexpr (PERSIST) {
        gemm_epilogue<EPI, NT, MT, true, 2>(acc, bias, R, ldr, C, ldc, M, N, m0_cur + wr * 128, n0_cur + wc * 64, fr, fg,
                                            smem + PSTAGES * PBUF + wave * 4096);
        if (!more_tiles) break;
        // the next tile's first K-tiles have landed (this wave's share) and the C stores are out; everybody's share after the barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    } else {
        gemm_epilogue<EPI, NT, MT, true>(acc, bias, R, ldr, C, ldc, M, N, m0_cur + wr * 128, n0_cur + wc * 64, fr, fg,
                                         smem + wave * (128 * 128));
        break;
    }
    }  // tiles
}

// -------------------------------------------------------------------------------------
// gemm_pipe_mx_kernel: W8A8 on the BLOCK-SCALED fp8 matrix instruction, v_mfma_scale_f32_32x32x64_f8f6f4 — the CDNA4 fp8 pipe
// at twice the bf16 rate (64 cycles for a 32x32x64 product against 32 for 32x32x16 bf16: MI355X_MICROARCH.md, matrix cores).
// The same pipeline as gemm_pipe_kernel (256x256 tile, 8 waves x (128 x 64), four 32 KiB LDS buffers, two phases per K-tile,
// counted vmcnt, staggered wave rows), with K-tiles of 64 codes: an e4m3 row of a K-tile is 64 bytes — the bf16 kernel's LDS
// geometry — and one MFMA consumes the whole K-tile.  Block scales: both operands carry E8M0 127 (= 2^0) for every 32-code
// block; the REAL scales (one f32 per row of A from kr_quantize_rows_fp8, one per row of W) multiply the f32 accumulators before
// the epilogue, as in kr_gemm_fp8a.
//   A image : 64-byte rows, 16-byte chunk c of row r at slot c ^ ((r >> 2) & 3): lane (row r = lane & 31, h = lane >> 5) reads
//             chunks 2h, 2h+1 with two ds_read_b128, whose 16-lane groups ({0-3,12-15,20-27}, ...) then hit 16 distinct slots;
//   W image : the decode layout's 1 KiB blocks as they are ([16-row block][k-group][row][16 B]): lane (r, h) reads k-groups 2h, 2h+1
//             of block r >> 4 — conflict-free as stored.  Lane (., h) holds the same 32 k of both operands (k-groups 2h, 2h+1),
//             which is all a dot product needs.
//   C layout: 32x32 accumulators, column (the A row m) on the lane, rows (the W row n) 8 (i >> 2) + 4 h + (i & 3) in register i —
//             a 16-row group of n keeps its SiLU gate rows (4h + j) and their up rows (8 + 4h + j) in the same lane.
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ void stage_a8_rows64(const uint8_t* __restrict__ g, int64_t ld, int64_t row0, int64_t rows_total, int k0,
                                                char* lds_op, int lane, int wave) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {  // the two 128-row halves: each wave fills one 16-row block (1 KiB) of each
        const int r = h * 128 + wave * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((lane >> 4) & 3);   // = (r >> 2) & 3: the 16-row block base is a multiple of 16
        int64_t gr = row0 + r;
        gr = gr < rows_total ? gr : rows_total - 1;
        const uint8_t* src = g + gr * ld + k0 + c * 16;
        char* dst = lds_op + (h * 128 + wave * 16) * 64;  // wave-uniform; the hardware adds lane * 16
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
}

__device__ __forceinline__ void stage_w8_rows64(const uint8_t* __restrict__ g, int K, int n0, int N, int k0, char* lds_op, int lane,
                                                int wave) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {  // 16 row blocks of 1 KiB: wave w stages blocks 2w, 2w + 1
        int rb = (n0 >> 4) + 2 * wave + h;
        rb = min(rb, (N >> 4) - 1);
        const uint8_t* src = g + ((int64_t)rb * (K >> 6) + (k0 >> 6)) * 1024 + lane * 16;
        char* dst = lds_op + (2 * wave + h) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
}

template <int EPI, bool TWO>
__global__ void __launch_bounds__(512) gemm_pipe_mx_kernel(const uint8_t* __restrict__ A8p, int64_t lda, const uint8_t* __restrict__ W8p,
                                                           const kr_bf16* __restrict__ bias, const kr_bf16* __restrict__ R, int64_t ldr,
                                                           kr_bf16* __restrict__ C, int64_t ldc, int64_t M, int N, int K, int tiles_n,
                                                           unsigned nwg, const float* __restrict__ w_scale, int group_m,
                                                           const float* __restrict__ a_scale) {
    constexpr int NT = 2, MT = 4, XK = 64, A_BYTES = 256 * XK, XBUF = 2 * A_BYTES;   // 32 KiB per stage, 4 stages
    constexpr int ONE = 0x7f7f7f7f;   // E8M0 127 in every byte: block scale 2^0
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int cl = lane & 31, h = lane >> 5;
    const unsigned wg = xcd_remap(blockIdx.x, nwg);
    unsigned tm, tn;
    tile_coords(wg, (unsigned)((M + 255) >> 8), (unsigned)tiles_n, (unsigned)group_m, tm, tn);
    const int64_t m0 = (int64_t)tm * 256;
    const int n0 = (int)tn * 256;

    f32x16 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = K / XK;
    // fragment read offsets inside a buffer
    const int a_off = (wr * 128 + cl) * 64 + ((((2 * h) ^ ((cl >> 2) & 3))) << 4);      // chunk 2h; chunk 2h+1 is the slot ^ 1
    const int a_off2 = (wr * 128 + cl) * 64 + ((((2 * h + 1) ^ ((cl >> 2) & 3))) << 4);
    const int w_off = A_BYTES + (wc * 4 + (cl >> 4)) * 1024 + (2 * h) * 256 + (cl & 15) * 16;   // k-group 2h; 2h+1 is + 256

    if constexpr (TWO) {
        // TWO K-tiles per barrier pair (K % 128 == 0): the matrix bursts between two barriers are 8 MFMAs = 512 cycles instead of
        // 256, against the same ~185 cycles a barrier interval costs besides its burst (both pipelined kernels measured 0.55-0.60
        // MFMA-busy with 256-cycle bursts: profiles/r03_pmc_sq_fp8_gemm.txt).  The four LDS buffers are two PAIRS: an iteration reads
        // the pair holding K-tiles 2i, 2i+1 and, in its first phase, restages the other pair (last read one iteration ago: its W
        // rows five barriers, its last A rows three barriers earlier) with K-tiles 2i+2, 2i+3 — one whole iteration to land, drained
        // (vmcnt(0): nothing else is in flight) before the iteration's last barrier.
        const int ni = nk >> 1;
        {
            stage_w8_rows64(W8p, K, n0, N, 0, smem + A_BYTES, lane, wave);
            stage_a8_rows64(A8p, lda, m0, M, 0, smem, lane, wave);
            stage_w8_rows64(W8p, K, n0, N, XK, smem + XBUF + A_BYTES, lane, wave);
            stage_a8_rows64(A8p, lda, m0, M, XK, smem + XBUF, lane, wave);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wr == 1) __builtin_amdgcn_s_barrier();  // stagger: wave row 1 runs one barrier behind wave row 0
        for (int i = 0; i < ni; ++i) {
            const char* cur = smem + (i & 1) * 2 * XBUF;
            char* oth = smem + ((i + 1) & 1) * 2 * XBUF;
            const bool more = i + 1 < ni;
            // ---------------- phase 0: W fragments of both K-tiles + A m-tiles 0, 1 of both; the other pair is restaged
            u32x4 wq[2][NT][2], xq[2][2][2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    wq[kt][t][0] = *reinterpret_cast<const u32x4*>(cur + kt * XBUF + w_off + t * 2048);
                    wq[kt][t][1] = *reinterpret_cast<const u32x4*>(cur + kt * XBUF + w_off + t * 2048 + 256);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    xq[kt][t][0] = *reinterpret_cast<const u32x4*>(cur + kt * XBUF + a_off + t * 32 * 64);
                    xq[kt][t][1] = *reinterpret_cast<const u32x4*>(cur + kt * XBUF + a_off2 + t * 32 * 64);
                }
            if (more) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    stage_w8_rows64(W8p, K, n0, N, (2 * i + 2 + kt) * XK, oth + kt * XBUF + A_BYTES, lane, wave);
                    stage_a8_rows64(A8p, lda, m0, M, (2 * i + 2 + kt) * XK, oth + kt * XBUF, lane, wave);
                }
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            i32x8 wf[2][NT];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        wf[kt][t][e] = (int)wq[kt][t][0][e];
                        wf[kt][t][4 + e] = (int)wq[kt][t][1][e];
                    }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    i32x8 xf;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        xf[e] = (int)xq[kt][mt][0][e];
                        xf[4 + e] = (int)xq[kt][mt][1][e];
                    }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[nt][mt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[kt][nt], xf, acc[nt][mt], 0, 0, 0, ONE, 0, ONE);
                }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
            // ---------------- phase 1: A m-tiles 2, 3 of both K-tiles; the restaged pair must have landed
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    xq[kt][t][0] = *reinterpret_cast<const u32x4*>(cur + kt * XBUF + a_off + (2 + t) * 32 * 64);
                    xq[kt][t][1] = *reinterpret_cast<const u32x4*>(cur + kt * XBUF + a_off2 + (2 + t) * 32 * 64);
                }
            // WAR: the OTHER wave row restages this pair right after the barrier below (its phase 0 of the next iteration runs
            // one barrier ahead or behind): these reads — the pair's last — must have COMPLETED, not just been issued, when
            // this wave arrives, so the LDS wait sits in front of the barrier here (the reading row has the slack: its
            // partner is in a 512-cycle burst)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    i32x8 xf;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        xf[e] = (int)xq[kt][mt][0][e];
                        xf[4 + e] = (int)xq[kt][mt][1][e];
                    }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[nt][2 + mt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[kt][nt], xf, acc[nt][2 + mt], 0, 0, 0, ONE, 0, ONE);
                }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
        }
    } else {
    // ---- prologue: K-tiles 0..2 requested, K-tile 0 landed and visible (4 loads per K-tile and wave)
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        if (t < nk) {
            char* buf = smem + t * XBUF;
            stage_w8_rows64(W8p, K, n0, N, t * XK, buf + A_BYTES, lane, wave);
            stage_a8_rows64(A8p, lda, m0, M, t * XK, buf, lane, wave);
        }
    }
    if (nk >= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (nk == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();  // stagger: wave row 1 runs one barrier behind wave row 0

    for (int k = 0; k < nk; ++k) {
        const char* cur = smem + (k & 3) * XBUF;
        char* nxt = smem + ((k + 3) & 3) * XBUF;
        const bool more = k + 3 < nk;
        // ---------------- phase 0: W fragments (2 n-tiles of 32) + A m-tiles 0, 1
        u32x4 wq[NT][2], xq[2][2];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            wq[t][0] = *reinterpret_cast<const u32x4*>(cur + w_off + t * 2048);
            wq[t][1] = *reinterpret_cast<const u32x4*>(cur + w_off + t * 2048 + 256);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            xq[t][0] = *reinterpret_cast<const u32x4*>(cur + a_off + t * 32 * 64);
            xq[t][1] = *reinterpret_cast<const u32x4*>(cur + a_off2 + t * 32 * 64);
        }
        if (more) stage_w8_rows64(W8p, K, n0, N, (k + 3) * XK, nxt + A_BYTES, lane, wave);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        i32x8 wf[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                wf[t][e] = (int)wq[t][0][e];
                wf[t][4 + e] = (int)wq[t][1][e];
            }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            i32x8 xf;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xf[e] = (int)xq[mt][0][e];
                xf[4 + e] = (int)xq[mt][1][e];
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[nt][mt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[nt], xf, acc[nt][mt], 0, 0, 0, ONE, 0, ONE);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
        // ---------------- phase 1: A m-tiles 2, 3; the counted wait of this K-tile
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            xq[t][0] = *reinterpret_cast<const u32x4*>(cur + a_off + (2 + t) * 32 * 64);
            xq[t][1] = *reinterpret_cast<const u32x4*>(cur + a_off2 + (2 + t) * 32 * 64);
        }
        if (more) stage_a8_rows64(A8p, lda, m0, M, (k + 3) * XK, nxt, lane, wave);
        if (more) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          // K-tile k+1 landed; k+2 and k+3 may stay in flight
        else if (k + 3 == nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            i32x8 xf;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xf[e] = (int)xq[mt][0][e];
                xf[4 + e] = (int)xq[mt][1][e];
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[nt][2 + mt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wf[nt], xf, acc[nt][2 + mt], 0, 0, 0, ONE, 0, ONE);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
    }
    }   // !TWO
    if (wr == 0) __builtin_amdgcn_s_barrier();  // pairs with wave row 1's last barrier

    // ---- epilogue: scales, bias / residual or SiLU*mul, bf16, through the wave's 16 KiB of LDS, whole-line stores
    constexpr bool HALF = EPI == KR_EPI_SILU_MUL8;
    constexpr int ROWB = HALF ? 64 : 128;
    char* const wlds = smem + wave * (128 * 128);
    const int64_t m_base = m0 + wr * 128;
    const int n_base = n0 + wc * 64;
    auto swz = [](int r) { return HALF ? ((r >> 1) & 3) : (r & 7); };
    float as[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int64_t m = m_base + mt * 32 + cl;
        as[mt] = a_scale[m < M ? m : M - 1];
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {       // register quad q: n = n_base + nt*32 + 8q + 4h + j
            const int nq = n_base + nt * 32 + 8 * q + 4 * h;
            const int ncl = min(nq, N - 4);
            const f32x4 ws4 = *reinterpret_cast<const f32x4*>(w_scale + ncl);
            const bf16x4 bv = bias ? *reinterpret_cast<const bf16x4*>(bias + ncl) : bf16x4{};
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[nt][mt][4 * q + j] = acc[nt][mt][4 * q + j] * ws4[j] * as[mt] + bf2f(bv[j]);
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int r = mt * 32 + cl;               // row inside the wave's 128
        const int64_t m = m_base + r;
        const int64_t mc = m < M ? m : M - 1;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if constexpr (HALF) {
#pragma unroll
                for (int k16 = 0; k16 < 2; ++k16) {   // 16-row group: gate rows in quad 2 k16, their up rows in quad 2 k16 + 1
                    bf16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = f2bf(act_silu(acc[nt][mt][8 * k16 + j]) * acc[nt][mt][8 * k16 + 4 + j]);
                    const int cc = nt * 2 + k16;      // 16-byte chunk (8 outputs) of the 64-byte scratch row; this lane's half: h
                    *reinterpret_cast<bf16x4*>(wlds + r * ROWB + ((cc ^ swz(r)) << 4) + h * 8) = o;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int nq = n_base + nt * 32 + 8 * q + 4 * h;
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[nt][mt][4 * q + j];
                    if (R) {
                        const bf16x4 rv = *reinterpret_cast<const bf16x4*>(R + mc * ldr + min(nq, N - 4));
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] += bf2f(rv[j]);
                    }
                    bf16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = f2bf(v[j]);
                    const int cc = nt * 4 + q;
                    *reinterpret_cast<bf16x4*>(wlds + r * ROWB + ((cc ^ swz(r)) << 4) + h * 8) = o;
                }
            }
        }
    }
    {   // same wave, in-order LDS: its reads see its writes
        constexpr int LPR = ROWB / 16, RPI = 64 / LPR;   // lanes per row, rows per instruction
        const int n_out = HALF ? (n_base >> 1) : n_base, n_lim = HALF ? (N >> 1) : N;
#pragma unroll
        for (int it = 0; it < 128 / RPI; ++it) {
            const int r = it * RPI + lane / LPR, c = lane % LPR;
            const u32x4 v = *reinterpret_cast<const u32x4*>(wlds + r * ROWB + ((c ^ swz(r)) << 4));
            const int64_t m = m_base + r;
            const int n = n_out + c * 8;
            if (m < M && n < n_lim) *reinterpret_cast<u32x4*>(C + m * ldc + n) = v;
        }
    }
}

template <int EPI, bool WPACK>
int launch_gemm_tail(const kr_bf16* A, int64_t lda, const kr_bf16* W, const kr_bf16* bias, const kr_bf16* R, int64_t ldr,
                     kr_bf16* C, int64_t ldc, int64_t M, int N, int K, int ptiles_n, unsigned tile0, unsigned n_tiles, int group_m,
                     kr_stream s);
inline int kr_cu_count();

// One 512-thread workgroup per CU at a time: a launch of T tiles takes ceil(T / CUs) ROUNDS, and the page-sized GEMMs
// sit badly on that grid — ViT proj / fc2 at 8 pages are 770 tiles = 3 rounds + 2 tiles (a fourth round for 0.3 % of
// the work), prefill o_proj / down_proj 264 tiles = 1 round + 8.  When the last round would be at most half full its
// tiles are cut into 128x128 quarters and run as a second launch of the two-barrier kernel (same k order: the same
// bits), which fills the chip with 4x as many, co-resident, short workgroups.  KARANTA_GEMM_TAIL=0 disables it.
template <int EPI, bool WPACK, bool W8 = false, bool A8 = false>
int launch_gemm_pipe(const kr_bf16* A, int64_t lda, const kr_bf16* W, const kr_bf16* bias, const kr_bf16* R, int64_t ldr,
                     kr_bf16* C, int64_t ldc, int64_t M, int N, int K, kr_stream s, const float* w_scale = nullptr,
                     const float* a_scale = nullptr) {
    constexpr int LDS = PSTAGES * PBUF;
    const int64_t tiles_m = (M + 255) / 256;
    const int tiles_n = (N + 255) / 256;
    int64_t nwg = tiles_m * tiles_n;
    KR_CHECK_ARG(nwg < (1ll << 31), "kr_gemm_bf16: grid too large");
    unsigned tail = 0;
    if constexpr (!W8) {
        const int cus = kr_cu_count();
        const char* env = getenv("KARANTA_GEMM_TAIL");
        const int64_t t = nwg % cus;
        // measured (tools/gemm_microbench.py, r2): a lone last round is cheaper than a full one (its few tiles have the
        // chip's clocks and memory system to themselves), so the split pays most where a tile is long — prefill down_proj
        // 414 -> 353 us, o_proj 81 -> 74, ViT fc2 460 -> 442, merger fc1 469 -> 435.  Round 3 (tools/gemm_tail_probe.py, the
        // tail on the 4-deep ring kernel, admission batches of 1..8 pages): at K = 1280 too every tail of <= cus / 2 tiles is
        // neutral or better split — proj 3 pages (34 tail tiles) 64.7 -> 59.3 us, qkv 1 page (44) 65.0 -> 58.4, fc1 2 pages
        // (12) 146.2 -> 136.5, fc1 8 pages (8) 499.7 -> 481.9 — so the round-2 condition "K >= 1536 or <= 4 tail tiles" is gone
        const char* kenv = getenv("KARANTA_GEMM_TAIL_MINK");   // tuning sweeps: the K from which any tail of <= cus / 2 tiles is split
        const int min_k = kenv ? atoi(kenv) : 0;
        if (nwg > cus && t > 0 && 2 * t <= cus && (K >= min_k || t <= 4) && !(env && atoi(env) == 0)) {
            tail = (unsigned)t;
            nwg -= t;
        }
    }
    static KrPerDeviceOnce attr_set;
    if (attr_set.need()) {
        KR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pipe_kernel<EPI, WPACK, W8, A8>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
#ifdef KR_GEMM_PERSIST_EXPERIMENT
        KR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pipe_kernel<EPI, WPACK, W8, A8, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS + 8 * 4096));
#endif
    }
    // measured (tools/gemm_microbench.py, r2): groups of 8 m tiles against m-major rows — ViT qkv (15 n tiles) 373 -> 350 us,
    // fc1 (20) 518 -> 496, prefill gate/up (70) 581 -> 562, 8192^3 856 -> 1390 TFLOP/s; with 5-6 n tiles an m-major run of
    // 32 ids is already a compact patch and groups are neutral (proj, down_proj) or worse (ViT fc2, K = 5120: 467 -> 495)
    const char* genv = getenv("KARANTA_GEMM_GROUP_M");
    const int group_m = genv ? atoi(genv) : (tiles_n >= 8 ? 8 : 1);
    const char* senv = getenv("KARANTA_GEMM_STAGGER");
    const int stagger = senv ? atoi(senv) : 0;
#ifdef KR_GEMM_PERSIST_EXPERIMENT
    // experiment build (tools/build_variant.py persist kr_gemm.hip -DKR_GEMM_PERSIST_EXPERIMENT): KARANTA_GEMM_PERSIST=0 (read per
    // call: A/B, tests) selects the product launch in the same library
    const char* penv = getenv("KARANTA_GEMM_PERSIST");
    const bool persist = (penv ? atoi(penv) != 0 : true) && nwg > kr_cu_count();
    if (persist)
        gemm_pipe_kernel<EPI, WPACK, W8, A8, true><<<(unsigned)kr_cu_count(), 512, LDS + 8 * 4096, kr_hs(s)>>>(
            A, lda, W, bias, R, ldr, C, ldc, M, N, K, tiles_n, (unsigned)nwg, w_scale, group_m, a_scale, 0);
    else
#endif
        gemm_pipe_kernel<EPI, WPACK, W8, A8><<<(unsigned)nwg, 512, LDS, kr_hs(s)>>>(A, lda, W, bias, R, ldr, C, ldc, M, N, K, tiles_n,
                                                                                    (unsigned)nwg, w_scale, group_m, a_scale, stagger);
    KR_CHECK_LAUNCH();
    if constexpr (!W8) {
        if (tail)
            return launch_gemm_tail<EPI, WPACK>(A, lda, W, bias, R, ldr, C, ldc, M, N, K, tiles_n, (unsigned)nwg, tail, group_m, s);
    }
    return KR_OK;
}

inline int kr_cu_count();

constexpr bool MX_TWO_DEFAULT = false;   // set by measurement (profiles/r03_fp8_gemm_bench.txt)

template <int EPI>
int launch_gemm_pipe_mx(const uint8_t* A8, int64_t lda, const uint8_t* W8, const kr_bf16* bias, const kr_bf16* R, int64_t ldr, kr_bf16* C,
                        int64_t ldc, int64_t M, int N, int K, kr_stream s, const float* w_scale, const float* a_scale) {
    constexpr int LDS = 4 * 32768;
    const int64_t tiles_m = (M + 255) / 256;
    const int tiles_n = (N + 255) / 256;
    const int64_t nwg = tiles_m * tiles_n;
    KR_CHECK_ARG(nwg < (1ll << 31), "kr_gemm_fp8a: grid too large");
    static KrPerDeviceOnce attr_set;
    if (attr_set.need()) {
        KR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pipe_mx_kernel<EPI, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        KR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pipe_mx_kernel<EPI, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    }
    const char* genv = getenv("KARANTA_GEMM_GROUP_M");
    const int group_m = genv ? atoi(genv) : (tiles_n >= 8 ? 8 : 1);
    const char* tenv = getenv("KARANTA_FP8_MX2");   // 1: two K-tiles per barrier pair where K % 128 == 0 (A/B, tests flip it)
    const bool two = (tenv ? atoi(tenv) != 0 : MX_TWO_DEFAULT) && (K % 128) == 0;
    if (two)
        gemm_pipe_mx_kernel<EPI, true><<<(unsigned)nwg, 512, LDS, kr_hs(s)>>>(A8, lda, W8, bias, R, ldr, C, ldc, M, N, K, tiles_n, (unsigned)nwg,
                                                                              w_scale, group_m, a_scale);
    else
        gemm_pipe_mx_kernel<EPI, false><<<(unsigned)nwg, 512, LDS, kr_hs(s)>>>(A8, lda, W8, bias, R, ldr, C, ldc, M, N, K, tiles_n, (unsigned)nwg,
                                                                               w_scale, group_m, a_scale);
    KR_CHECK_LAUNCH();
    return KR_OK;
}

template <int EPI, bool WPACK, typename G, int STAGES>
int launch_gemm_kernel(unsigned nwg, const kr_bf16* A, int64_t lda, const kr_bf16* W, const kr_bf16* bias, const kr_bf16* R,
                       int64_t ldr, kr_bf16* C, int64_t ldc, int64_t M, int N, int K, int tiles_n, int ptiles_n, unsigned qbase,
                       int group_m, kr_stream s, int ksplit = 1, int ks_mode = 0, float* ws = nullptr) {
    constexpr int LDS = STAGES * (G::BM + G::BN) * BK * 2;
    static KrPerDeviceOnce attr_set;
    if (attr_set.need()) {
        KR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<EPI, WPACK, G, STAGES>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    }
    gemm_kernel<EPI, WPACK, G, STAGES><<<nwg, G::WM * G::WN * 64, LDS, kr_hs(s)>>>(A, lda, W, bias, R, ldr, C, ldc, M, N, K, tiles_n,
                                                                                  nwg, ptiles_n, qbase, group_m, ksplit, ks_mode, ws);
    KR_CHECK_LAUNCH();
    return KR_OK;
}

// 128x128 launches of at most one workgroup per CU take the 4-deep ring (see gemm_kernel); KARANTA_GEMM_STAGES=2 forces
// the two-buffer form (A/B, tests).
inline bool gemm_deep_ring(int64_t nwg) {
    const char* env = getenv("KARANTA_GEMM_STAGES");
    if (env) return atoi(env) == 4;
    return nwg <= kr_cu_count();
}

template <int EPI, bool WPACK, typename G>
int launch_gemm3(const kr_bf16* A, int64_t lda, const kr_bf16* W, const kr_bf16* bias, const kr_bf16* R, int64_t ldr,
                 kr_bf16* C, int64_t ldc, int64_t M, int N, int K, kr_stream s) {
    const int64_t tiles_m = (M + G::BM - 1) / G::BM;
    const int tiles_n = (N + G::BN - 1) / G::BN;
    const int64_t nwg = tiles_m * tiles_n;
    KR_CHECK_ARG(nwg < (1ll << 31), "kr_gemm_bf16: grid too large");
    if constexpr (G::BM == 128) {
        if (gemm_deep_ring(nwg))
            return launch_gemm_kernel<EPI, WPACK, G, 4>((unsigned)nwg, A, lda, W, bias, R, ldr, C, ldc, M, N, K, tiles_n, 0, 0u, 0, s);
    }
    return launch_gemm_kernel<EPI, WPACK, G, 2>((unsigned)nwg, A, lda, W, bias, R, ldr, C, ldc, M, N, K, tiles_n, 0, 0u, 0, s);
}

// Split-K scratch of the tail launches: CALLER-OWNED (kr_gemm_bf16_ws: 128x128 f32 per partial workgroup, KR_GEMM_SCRATCH_BYTES in
// all).  The library allocates nothing: a call without scratch (kr_gemm_bf16) runs its tail unsplit, a call with scratch splits
// it — the same call gives the same bits whatever ran before it, in a stream capture or not (the lazily
// allocated per-stream scratch made the accumulation order depend on call history).  The scratch of the current entry-point
// call, valid only while that call is on this thread's stack:
constexpr int TAIL_MAX_WGS = 512, TAIL_WG_BYTES = 256 * 16 * 16;   // 128x128 f32 per workgroup
static_assert((size_t)TAIL_MAX_WGS * TAIL_WG_BYTES == KR_GEMM_SCRATCH_BYTES, "karanta_hip.h: KR_GEMM_SCRATCH_BYTES");
static thread_local float* t_call_scratch = nullptr;
struct CallScratch {
    explicit CallScratch(float* p) { t_call_scratch = p; }
    ~CallScratch() { t_call_scratch = nullptr; }
};

// The quarters of the 256x256 tiles [tile0, tile0 + n_tiles) of a launch_gemm_pipe tile list, as 128x128 workgroups; with a
// long K (>= 4096) and few tail tiles each quarter is cut into K ranges (partials launch + reduce-and-epilogue launch, see
// gemm_kernel) so that the tail's bytes are pulled by ~256-512 workgroups instead of 4 per tail tile.
template <int EPI, bool WPACK>
int launch_gemm_tail(const kr_bf16* A, int64_t lda, const kr_bf16* W, const kr_bf16* bias, const kr_bf16* R, int64_t ldr,
                     kr_bf16* C, int64_t ldc, int64_t M, int N, int K, int ptiles_n, unsigned tile0, unsigned n_tiles, int group_m,
                     kr_stream s) {
    const unsigned nq = 4 * n_tiles;
    const char* env = getenv("KARANTA_GEMM_TAIL_KSPLIT");   // 0 / 1: never split (A/B, tests); n: at most n ranges
    const int cap = env ? atoi(env) : 16;
    int ksplit = 1;
    // one round of workgroups (4-deep ring: one per CU), at least 8 K steps each
    if (cap > 1 && K >= 4096) ksplit = std::max(1, std::min(std::min(cap, (K / BK) / 8), (int)(kr_cu_count() / nq)));
    float* ws = (ksplit > 1 && nq * (unsigned)ksplit <= (unsigned)TAIL_MAX_WGS) ? t_call_scratch : nullptr;
    if (ws) {
        int rc = launch_gemm_kernel<EPI, WPACK, G128, 4>(nq * ksplit, A, lda, W, bias, R, ldr, C, ldc, M, N, K, 1, ptiles_n, tile0,
                                                         group_m, s, ksplit, 1, ws);
        if (rc != KR_OK) return rc;
        return launch_gemm_kernel<EPI, WPACK, G128, 2>(nq, A, lda, W, bias, R, ldr, C, ldc, M, N, K, 1, ptiles_n, tile0, group_m, s,
                                                       ksplit, 2, ws);
    }
    if (gemm_deep_ring(nq))
        return launch_gemm_kernel<EPI, WPACK, G128, 4>(nq, A, lda, W, bias, R, ldr, C, ldc, M, N, K, 1, ptiles_n, tile0, group_m, s);
    return launch_gemm_kernel<EPI, WPACK, G128, 2>(nq, A, lda, W, bias, R, ldr, C, ldc, M, N, K, 1, ptiles_n, tile0, group_m, s);
}

// Compute units of the current device (cached per device): the round size of a one-workgroup-per-CU launch.
inline int kr_cu_count() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cus[dev] == 0) {
        int n = 0;
        cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cus[dev];
}

// 0 = automatic; 128 / 256 / 512 (= pipelined 256x256) forced by KARANTA_GEMM_TILE (tests, tuning sweeps)
inline int gemm_tile_choice(int64_t M, int N, int K) {
    const char* env = getenv("KARANTA_GEMM_TILE");  // read per call: the tests flip it between launches
    const int forced = env ? atoi(env) : 0;
    if (forced == 128 || forced == 256 || forced == 512) return forced;   // 512: the pipelined 256x256 kernel
    // the 256x256 tile (pipelined kernel) wants about one workgroup per CU or more and no N padding; measured on the
    // ViT / prefill / merger shapes (gemm_microbench.py) it beats the 128x128 tile by 10-40 % from 234 workgroups up
    const int64_t wgs = ((M + 255) / 256) * ((N + 255) / 256);
    return (wgs >= 192 && N % 256 == 0) ? 512 : 128;
}

template <int EPI, bool WPACK>
int launch_gemm2(const kr_bf16* A, int64_t lda, const kr_bf16* W, const kr_bf16* bias, const kr_bf16* R, int64_t ldr,
                 kr_bf16* C, int64_t ldc, int64_t M, int N, int K, kr_stream s) {
    int tile = gemm_tile_choice(M, N, K);
    // the 256x256 kernels store C through LDS as 16-byte pieces
    if ((ldc & 7) != 0 || (reinterpret_cast<uintptr_t>(C) & 15) != 0) tile = 128;
    if (tile == 512) return launch_gemm_pipe<EPI, WPACK>(A, lda, W, bias, R, ldr, C, ldc, M, N, K, s);
    if (tile == 256) return launch_gemm3<EPI, WPACK, G256>(A, lda, W, bias, R, ldr, C, ldc, M, N, K, s);
    return launch_gemm3<EPI, WPACK, G128>(A, lda, W, bias, R, ldr, C, ldc, M, N, K, s);
}

template <int EPI>
int launch_gemm(const kr_bf16* A, int64_t lda, const kr_bf16* W, const kr_bf16* bias, const kr_bf16* R, int64_t ldr,
                kr_bf16* C, int64_t ldc, int64_t M, int N, int K, int w_packed, kr_stream s) {
    return w_packed ? launch_gemm2<EPI, true>(A, lda, W, bias, R, ldr, C, ldc, M, N, K, s)
                    : launch_gemm2<EPI, false>(A, lda, W, bias, R, ldr, C, ldc, M, N, K, s);
}

}  // namespace

extern "C" int kr_gemm_bf16_ws(const kr_bf16* A, int64_t lda, const kr_bf16* W, const kr_bf16* bias,
                               const kr_bf16* residual, int64_t ldr, kr_bf16* C, int64_t ldc, int64_t M, int N, int K,
                               int epilogue, int w_packed, float* scratch, size_t scratch_bytes, kr_stream s) {
    KR_CHECK_ARG(scratch == nullptr || (scratch_bytes >= KR_GEMM_SCRATCH_BYTES && ((uintptr_t)scratch & 15) == 0),
                 "kr_gemm_bf16_ws: scratch of %zu bytes (KR_GEMM_SCRATCH_BYTES = %zu, 16-byte aligned)", scratch_bytes,
                 (size_t)KR_GEMM_SCRATCH_BYTES);
    CallScratch guard(scratch);
    return kr_gemm_bf16(A, lda, W, bias, residual, ldr, C, ldc, M, N, K, epilogue, w_packed, s);
}

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

// ---- dynamic per-token fp8 quantisation of activations (the A operand of kr_gemm_fp8a)
namespace {
// One 256-thread workgroup per row: the row in registers (K <= 8 * 256 * QMAX), max |x| over the workgroup, scale = max / 448
// (1 for an all-zero row), codes = e4m3(x / scale) by v_cvt_pk_fp8_f32 (round to nearest even; |x / scale| <= 448 up to one
// rounding of the division, far below the 464 where e4m3 would round up past its largest finite value).  Restated on the
// host by weights.quantize_fp8_rows (same f32 division, same rounding): the codes are bit-identical (GPU test).
template <int QMAX>
__global__ void __launch_bounds__(256) quantize_rows_fp8_kernel(const kr_bf16* __restrict__ x, int64_t ldx, uint8_t* __restrict__ q,
                                                                int64_t ldq, float* __restrict__ scale, int K) {
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kc = K >> 3;
    const kr_bf16* xr = x + (int64_t)blockIdx.x * ldx;
    bf16x8 v[QMAX];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < QMAX; ++i) {
        const int c = tid + i * 256;
        if (c < kc) {
            v[i] = ld8(xr + c * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(bf2f(v[i][j])));
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if (lane == 0) red[wave] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float sc = amax > 0.f ? amax / 448.0f : 1.0f;
    if (tid == 0) scale[blockIdx.x] = sc;
    uint8_t* qr = q + (int64_t)blockIdx.x * ldq;
#pragma unroll
    for (int i = 0; i < QMAX; ++i) {
        const int c = tid + i * 256;
        if (c < kc) {
            u32x2 o;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int w = 0;
                w = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v[i][4 * h + 0]) / sc, bf2f(v[i][4 * h + 1]) / sc, w, false);
                w = __builtin_amdgcn_cvt_pk_fp8_f32(bf2f(v[i][4 * h + 2]) / sc, bf2f(v[i][4 * h + 3]) / sc, w, true);
                o[h] = (unsigned)w;
            }
            *reinterpret_cast<u32x2*>(qr + c * 8) = o;
        }
    }
}
}  // namespace

extern "C" int kr_quantize_rows_fp8(const kr_bf16* x, int64_t ldx, uint8_t* q, int64_t ldq, float* scale, int64_t rows, int K,
                                    kr_stream s) {
    KR_CHECK_ARG(x && q && scale, "kr_quantize_rows_fp8: null pointer");
    KR_CHECK_ARG(rows >= 0 && rows < (1ll << 31) && K > 0 && K % 8 == 0 && K <= 8 * 256 * 12, "kr_quantize_rows_fp8: rows=%lld K=%d (K %% 8, K <= 24576)",
                 (long long)rows, K);
    KR_CHECK_ARG(ldx >= K && (ldx & 7) == 0 && ldq >= K && (ldq & 15) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)q & 15) == 0,
                 "kr_quantize_rows_fp8: strides / alignment");
    if (rows == 0) return KR_OK;
    const int per = ((K >> 3) + 255) / 256;
    if (per <= 2) quantize_rows_fp8_kernel<2><<<(unsigned)rows, 256, 0, kr_hs(s)>>>(x, ldx, q, ldq, scale, K);
    else if (per <= 4) quantize_rows_fp8_kernel<4><<<(unsigned)rows, 256, 0, kr_hs(s)>>>(x, ldx, q, ldq, scale, K);
    else if (per <= 8) quantize_rows_fp8_kernel<8><<<(unsigned)rows, 256, 0, kr_hs(s)>>>(x, ldx, q, ldq, scale, K);
    else quantize_rows_fp8_kernel<12><<<(unsigned)rows, 256, 0, kr_hs(s)>>>(x, ldx, q, ldq, scale, K);
    KR_CHECK_LAUNCH();
    return KR_OK;
}

extern "C" int kr_gemm_fp8a(const uint8_t* A8, int64_t lda, const float* a_scale, const uint8_t* w_packed_fp8, const float* w_scale,
                            const kr_bf16* bias, const kr_bf16* residual, int64_t ldr, kr_bf16* C, int64_t ldc, int64_t M, int N, int K,
                            int epilogue, kr_stream s) {
    KR_CHECK_ARG(A8 && a_scale && w_packed_fp8 && w_scale && C, "kr_gemm_fp8a: null pointer");
    KR_CHECK_ARG(M >= 0 && N > 0 && K > 0, "kr_gemm_fp8a: bad sizes M=%lld N=%d K=%d", (long long)M, N, K);
    KR_CHECK_ARG(K % 64 == 0 && N % 16 == 0, "kr_gemm_fp8a: K=%d must be a multiple of 64, N=%d of 16", K, N);
    KR_CHECK_ARG(lda >= K && (lda & 15) == 0, "kr_gemm_fp8a: lda=%lld (bytes, a multiple of 16)", (long long)lda);
    KR_CHECK_ARG((ldc & 7) == 0 && ((uintptr_t)C & 15) == 0 && (residual == nullptr || (ldr & 3) == 0),
                 "kr_gemm_fp8a: C must be 16-byte aligned with ldc %% 8 == 0 (ldr %% 4 == 0)");
    KR_CHECK_ARG(((uintptr_t)A8 & 15) == 0 && ((uintptr_t)w_packed_fp8 & 15) == 0 && ((uintptr_t)w_scale & 15) == 0,
                 "kr_gemm_fp8a: pointer alignment");
    if (M == 0) return KR_OK;
    const kr_bf16* Ap = reinterpret_cast<const kr_bf16*>(A8);
    const kr_bf16* Wp = reinterpret_cast<const kr_bf16*>(w_packed_fp8);
    // KARANTA_FP8_MX (read per call: the tests flip it): 1 (default) = the block-scaled instruction v_mfma_scale_f32_32x32x64_f8f6f4
    // (twice the bf16 rate); 0 = v_mfma_f32_16x16x32_fp8_fp8 in the bf16 kernel's pipeline (the bf16 rate).  Same products
    // (exact in f32), another summation order.
    const char* mxe = getenv("KARANTA_FP8_MX");
    const bool mx = !(mxe && atoi(mxe) == 0) && N % 32 == 0;
    switch (epilogue) {
        case KR_EPI_NONE:
            KR_CHECK_ARG(ldc >= N, "kr_gemm_fp8a: ldc < N");
            if (mx) return launch_gemm_pipe_mx<KR_EPI_NONE>(A8, lda, w_packed_fp8, bias, residual, ldr, C, ldc, M, N, K, s, w_scale, a_scale);
            return launch_gemm_pipe<KR_EPI_NONE, true, true, true>(Ap, lda, Wp, bias, residual, ldr, C, ldc, M, N, K, s, w_scale, a_scale);
        case KR_EPI_SILU_MUL8:
            KR_CHECK_ARG(ldc >= N / 2 && !residual, "kr_gemm_fp8a: SILU_MUL8 takes no residual");
            if (mx) return launch_gemm_pipe_mx<KR_EPI_SILU_MUL8>(A8, lda, w_packed_fp8, bias, residual, ldr, C, ldc, M, N, K, s, w_scale, a_scale);
            return launch_gemm_pipe<KR_EPI_SILU_MUL8, true, true, true>(Ap, lda, Wp, bias, residual, ldr, C, ldc, M, N, K, s, w_scale, a_scale);
        default:
            kr_set_error("kr_gemm_fp8a: epilogue %d (NONE and SILU_MUL8 only: the decoder's prefill linears)", epilogue);
            return KR_ERR_ARG;
    }
}

extern "C" int kr_gemm_fp8(const kr_bf16* A, int64_t lda, const uint8_t* w_packed_fp8, const float* w_scale, const kr_bf16* bias,
                           const kr_bf16* residual, int64_t ldr, kr_bf16* C, int64_t ldc, int64_t M, int N, int K, int epilogue,
                           kr_stream s) {
    KR_CHECK_ARG(A && w_packed_fp8 && w_scale && C, "kr_gemm_fp8: null pointer");
    KR_CHECK_ARG(M >= 0 && N > 0 && K > 0, "kr_gemm_fp8: bad sizes M=%lld N=%d K=%d", (long long)M, N, K);
    KR_CHECK_ARG(K % 64 == 0 && N % 16 == 0, "kr_gemm_fp8: K=%d must be a multiple of 64, N=%d of 16", K, N);
    KR_CHECK_ARG(lda >= K && (lda & 7) == 0, "kr_gemm_fp8: lda=%lld", (long long)lda);
    KR_CHECK_ARG((ldc & 7) == 0 && ((uintptr_t)C & 15) == 0 && (residual == nullptr || (ldr & 3) == 0),
                 "kr_gemm_fp8: C must be 16-byte aligned with ldc %% 8 == 0 (ldr %% 4 == 0)");
    KR_CHECK_ARG(((uintptr_t)A & 15) == 0 && ((uintptr_t)w_packed_fp8 & 15) == 0 && ((uintptr_t)w_scale & 15) == 0,
                 "kr_gemm_fp8: pointer alignment");
    if (M == 0) return KR_OK;
    const kr_bf16* Wp = reinterpret_cast<const kr_bf16*>(w_packed_fp8);
    switch (epilogue) {
        case KR_EPI_NONE:
            KR_CHECK_ARG(ldc >= N, "kr_gemm_fp8: ldc < N");
            return launch_gemm_pipe<KR_EPI_NONE, true, true>(A, lda, Wp, bias, residual, ldr, C, ldc, M, N, K, s, w_scale);
        case KR_EPI_SILU_MUL8:
            KR_CHECK_ARG(ldc >= N / 2 && !residual, "kr_gemm_fp8: SILU_MUL8 takes no residual");
            return launch_gemm_pipe<KR_EPI_SILU_MUL8, true, true>(A, lda, Wp, bias, residual, ldr, C, ldc, M, N, K, s, w_scale);
        default:
            kr_set_error("kr_gemm_fp8: epilogue %d (NONE and SILU_MUL8 only: the decoder's prefill linears)", epilogue);
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
