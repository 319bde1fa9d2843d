// Attention path: rotary + layout prep, flash-style varlen attention (ViT full / decoder causal
// prefill), decode attention (q_len = 1, GQA, split-KV) — all on v_mfma bf16 with fp32 softmax.
//
// Data layout in HBM (chosen for the matrix cores, not inherited from any framework):
//   Q   : [heads, n, hd]                     rotated, bf16
//   K   : [kv_heads, rows, hd]               rotated, bf16 (decoder: the KV cache itself)
//   V^T : [kv_heads, blocks, 2, hd, 32]      V transposed inside 64-token blocks, each block as two contiguous 32-key halves
//                                            (kr_common.h, kr_vt_off; rounds 1-3: [hd, 64]), so that the
//                                            P*V MFMA operand (8 consecutive keys for one d)
//                                            is one contiguous 16-byte read per lane.
#include "kr_common.h"

namespace {

// =====================================================================================
// rotary helpers
// =====================================================================================

// x*cos + rotate_half(x)*sin on one head vector, 8 elements [c0, c0+8) per call.
// `other` holds x[(c + hd/2) mod hd]; sign = -1 for the first half, +1 for the second.
__device__ __forceinline__ bf16x8 rope8(bf16x8 x, bf16x8 other, const float* cs, const float* sn, float sign) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(x[j]) * cs[j] + sign * bf2f(other[j]) * sn[j]);
    return o;
}

// In-place rotary on [n, heads, hd] (row stride given): standalone operator.
__global__ void __launch_bounds__(256) rope_inplace_kernel(kr_bf16* __restrict__ x, const float* __restrict__ cos,
                                                           const float* __restrict__ sin, int64_t n, int heads, int hd,
                                                           int64_t row_stride) {
    const int half_chunks = hd >> 4;  // chunks of 8 in half a head
    const int64_t total = n * heads * half_chunks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % half_chunks);
        const int64_t t = i / half_chunks;
        const int h = (int)(t % heads);
        const int64_t tok = t / heads;
        kr_bf16* p = x + tok * row_stride + (int64_t)h * hd;
        const int d0 = c * 8, d1 = d0 + (hd >> 1);
        const bf16x8 a = ld8(p + d0), b = ld8(p + d1);
        float c0[8], s0[8], c1[8], s1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            c0[j] = cos[tok * hd + d0 + j];
            s0[j] = sin[tok * hd + d0 + j];
            c1[j] = cos[tok * hd + d1 + j];
            s1[j] = sin[tok * hd + d1 + j];
        }
        st8(p + d0, rope8(a, b, c0, s0, -1.f));
        st8(p + d1, rope8(b, a, c1, s1, 1.f));
    }
}

// =====================================================================================
// prep: rotary + re-layout (+ V transpose through LDS) for 64-token blocks
// =====================================================================================
// grid = (n_blocks, ceil((q_heads + kv_heads) / PREP_G)).  Block i covers tokens blk_tok0[i] .. +blk_ntok[i] (<= 64)
// of the flattened activation `qkv` (row stride ld_qkv); they belong to one segment/sequence and
// start at a multiple of 64 inside it, so the block owns one whole V^T block.  Head slot y of the PREP_G the block takes:
//   y <  q_heads : rotate q head y            -> q_out[y][tok][hd]
//   y >= q_heads : rotate k head, copy V^T    -> k rows (k_row0[i] + j), V^T block vt_blk[i]
// Column offsets of q / k / v inside a qkv row are given in elements.
// A thread keeps the cos / sin of its (token, 8-channel chunk) in registers across the block's head slots: with one
// head per block (r1) the fp32 tables were 4x the bytes of the activations they rotate and the ViT launch ran at
// 2.5 TB/s of useful traffic (240 us per layer for 8 pages).
constexpr int PREP_G = 8;
template <int HD>
__global__ void __launch_bounds__(256) qkv_prep_kernel(const kr_bf16* __restrict__ qkv, int64_t ld_qkv, int q_off,
                                                       int k_off, int v_off, const float* __restrict__ cos,
                                                       const float* __restrict__ sin,
                                                       const int32_t* __restrict__ blk_tok0,
                                                       const int32_t* __restrict__ blk_ntok,
                                                       const int64_t* __restrict__ blk_k_row0,
                                                       const int64_t* __restrict__ blk_vt_blk, kr_bf16* __restrict__ q_out,
                                                       int64_t q_head_stride, kr_bf16* __restrict__ k_out,
                                                       int64_t k_head_stride, kr_bf16* __restrict__ vt_out,
                                                       int64_t vt_head_stride, int q_heads, int n_slots) {
    constexpr int HC = HD / 16;  // 8-element chunks in half a head
    constexpr int VT_RS = 33;    // dwords per V^T row in LDS: 32 token pairs + 1
    __shared__ unsigned vt_s[HD * VT_RS];
    const int i = blockIdx.x;
    const int y0 = blockIdx.y * PREP_G, y1 = min(y0 + PREP_G, n_slots);
    const int tok0 = blk_tok0[i], ntok = blk_ntok[i];
    const int64_t k_row0 = blk_k_row0[i];
    // ---- rotary on 64 tokens x HD, all head slots of the block
    for (int e = threadIdx.x; e < 64 * HC; e += 256) {
        const int j = e / HC, c = e - j * HC;
        if (j >= ntok) continue;
        const int64_t tok = tok0 + j;
        const int d0 = c * 8, d1 = d0 + HD / 2;
        float c0[8], s0[8], c1[8], s1[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            c0[jj] = cos[tok * HD + d0 + jj];
            s0[jj] = sin[tok * HD + d0 + jj];
            c1[jj] = cos[tok * HD + d1 + jj];
            s1[jj] = sin[tok * HD + d1 + jj];
        }
        const kr_bf16* row = qkv + tok * ld_qkv;
#pragma unroll 2
        for (int y = y0; y < y1; ++y) {
            const bool is_q = y < q_heads;
            const int head = is_q ? y : y - q_heads;
            const kr_bf16* p = row + (is_q ? q_off : k_off) + head * HD;
            kr_bf16* dst = is_q ? q_out + (int64_t)head * q_head_stride + tok * HD
                                : k_out + (int64_t)head * k_head_stride + (k_row0 + j) * HD;
            const bf16x8 a = ld8(p + d0), b = ld8(p + d1);
            st8(dst + d0, rope8(a, b, c0, s0, -1.f));
            st8(dst + d1, rope8(b, a, c1, s1, 1.f));
        }
    }
    // ---- V of the block's k-head slots: [64 tok][HD] -> LDS transposed -> V^T block [HD][64], zero padded past ntok.
    // A thread takes one 8-channel chunk of a PAIR of tokens and writes 8 whole dwords (two tokens of one channel): rows
    // of 33 dwords put the 10 / 16 chunks of a pair on distinct banks (r1 / early r2: 2-byte writes, 8-way conflicts).
    for (int y = max(y0, q_heads); y < y1; ++y) {
        const int head = y - q_heads;
        for (int e = threadIdx.x; e < 32 * (HD / 8); e += 256) {
            const int jp = e / (HD / 8), c = e - jp * (HD / 8);
            u32x4 va = (u32x4){0u, 0u, 0u, 0u}, vb = va;
            const kr_bf16* src = qkv + (int64_t)(tok0 + 2 * jp) * ld_qkv + v_off + head * HD + c * 8;
            if (2 * jp < ntok) va = *reinterpret_cast<const u32x4*>(src);
            if (2 * jp + 1 < ntok) vb = *reinterpret_cast<const u32x4*>(src + ld_qkv);
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const unsigned lo = (va[jj >> 1] >> ((jj & 1) * 16)) & 0xffffu, hi = (vb[jj >> 1] >> ((jj & 1) * 16)) & 0xffffu;
                vt_s[(c * 8 + jj) * VT_RS + jp] = lo | (hi << 16);
            }
        }
        __syncthreads();
        kr_bf16* vt = vt_out + (int64_t)head * vt_head_stride + blk_vt_blk[i] * (int64_t)(HD * 64);
        for (int e = threadIdx.x; e < HD * 8; e += 256) {   // e = the 16-byte piece in MEMORY order: consecutive threads, whole lines
            const int half = e >= HD * 4 ? 1 : 0, rem = e - half * (HD * 4);
            const int d = rem >> 2, c = half * 4 + (rem & 3);
            const unsigned* r = vt_s + d * VT_RS + c * 4;
            *reinterpret_cast<u32x4*>(vt + e * 8) = (u32x4){r[0], r[1], r[2], r[3]};      // = vt + kr_vt_off(d, c * 8, HD)
        }
        __syncthreads();
    }
}

// =====================================================================================
// flash-style varlen attention, v_mfma_f32_32x32x16_bf16
// =====================================================================================
// Block = 4 waves = 128 queries of one head; each wave owns 32 queries.  Per 64-key tile:
//   S^T[key][q]  = K (A operand, LDS) x Q^T (B operand, registers)        2 x HD/16 MFMA
//   online softmax in registers: the query is on the lane, its 32 keys in 2x16 accumulator
//   registers; one cross-lane max with lane^32
//   O^T[d][q]   += V^T (A operand, LDS) x P^T (B operand = the S^T accumulators, converted in
//   place to bf16: no LDS round trip, the MFMA k-order permutation is absorbed by reading V^T
//   keys in the same order)                                               HD/32 x 4 MFMA
// Diagnostic build only (-DKR_ATTN_STAMPS, tools/build_variant.py): s_memtime stamps around the segments of one tile
// iteration, summed per wave and left in a buffer of their own; no shipped build contains a stamp.
#ifdef KR_ATTN_STAMPS
__device__ unsigned long long kr_attn_dbg[8 * 8];
#define KR_STAMP(var)                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                             \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");                    \
    __builtin_amdgcn_sched_barrier(0)
extern "C" int kr_attn_debug_read(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(kr_attn_dbg), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : 1;
}
#else
#define KR_STAMP(var)
#endif
#ifndef KR_ATTN_PRESCALE
#define KR_ATTN_PRESCALE 1
#endif
#ifndef KR_ATTN_ROT
#define KR_ATTN_ROT 0   // measured slower (see ROT in the kernel): build with -DKR_ATTN_ROT=1 to repeat the experiment
#endif
#ifndef KR_ATTN_VPRE
#define KR_ATTN_VPRE(HD) ((HD) == 80 ? 3 : 0)
#endif
#ifndef KR_PIPE_VPRE
#define KR_PIPE_VPRE 1   // pipelined kernel: V^T fragments read one step ahead of their PV
#endif
template <int HD>
struct AttnCfg {
    static constexpr int KS = HD / 16;               // QK^T k-steps
    static constexpr int DT = (HD + 31) / 32;        // 32-row tiles of O^T
    static constexpr int KCH = HD / 8;               // 16-byte chunks per K row
    static constexpr int KROW = (HD == 128) ? 256 : (KCH + 1) * 16;  // LDS K row bytes (hd=80: 176, conflict-free)
    static constexpr int VROW = 136;                 // LDS V^T row bytes (64 keys + 8 B pad: conflict-free b64 reads)
};

// Chunk `vi` of a V^T block in MEMORY order — the block is [2 halves][HD][32 keys] (kr_common.h, kr_vt_off), a chunk is 8 keys of
// one channel — goes to row d = channel, 16-byte piece 4 * half + (vi & 3) of the [HD][64 keys] LDS image.  The staging loads
// stay linear over the block (whole lines), only the LDS destination knows about the halves.
template <int HD>
__device__ __forceinline__ int vt_lds_piece(int vi, int vrow) {
    const int half = vi >= HD * 4 ? 1 : 0, rem = vi - half * (HD * 4);
    return (rem >> 2) * vrow + (half * 4 + (rem & 3)) * 16;
}

template <int HD>
__device__ __forceinline__ int k_lds_off(int key, int c) {
    if (HD == 128) return key * 256 + ((c ^ (key & 15)) << 4);
    return key * AttnCfg<HD>::KROW + (c << 4);
}

// NW = waves per workgroup: 4 (128 queries) or 8 (256 queries sharing one K / V^T tile image: half the tile staging per
// query).  Either way the registers (~200-254) allow two waves per SIMD — amdgpu_waves_per_eu(2) makes the compiler keep
// that — and the loop is bound by vector-ALU ISSUE slots (per 64-key tile and wave: 32 v_exp + 32 v_fma + 16 v_max3 +
// 16 v_cvt_pk + addresses, about as many cycles as the tile's 22 (hd 80) / 32 (hd 128) MFMAs hold the matrix pipe), so
// what the r2 work removed were instructions and waits, not bytes: see the comments at the staging lambdas, at ONES
// and at the permlane swap.  -DKR_ATTN_STAMPS (tools/build_variant.py) builds the per-segment cycle stamps this was
// read from.
template <int HD, bool CAUSAL, int NW>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2))) attn_varlen_kernel(const kr_bf16* __restrict__ q, const kr_bf16* __restrict__ k,
                                                          const kr_bf16* __restrict__ vt, kr_bf16* __restrict__ out,
                                                          const int32_t* __restrict__ qblk,
                                                          const int32_t* __restrict__ qblk_len, int64_t nq_total,
                                                          int q_heads, int group, int64_t k_head_stride,
                                                          int64_t vt_head_stride, float scale_log2e) {
    using C = AttnCfg<HD>;
    constexpr int NTHR = NW * 64;
    // staging: the tile's 16-byte chunks, K rows then V^T rows, as ONE list dealt to the threads in whole passes
    constexpr int K_CH = 64 * C::KCH, V_CH = HD * 8, T_CH = K_CH + V_CH, PASSES = (T_CH + NTHR - 1) / NTHR;
    constexpr int VPRE = KR_ATTN_VPRE(HD);  // V^T tiles prefetched across the softmax
    // two tile images [K | V^T]: tile t+1 is written while tile t is read, one barrier per tile
    constexpr int K_BYTES = 64 * C::KROW, V_BYTES = C::DT * 32 * C::VROW, IMG = K_BYTES + V_BYTES;
    __shared__ __attribute__((aligned(16))) char img_s[2 * IMG];
#ifdef KR_ATTN_LDS_PAD
    __shared__ char lds_pad[KR_ATTN_LDS_PAD];
    if (nq_total < 0) lds_pad[threadIdx.x] = 1, out[0] = lds_pad[threadIdx.x ^ 1];
#endif

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, lh = lane >> 5;
    // head is the fastest index of the 1-D grid: the workgroups of one query block start together (the plan lists the
    // blocks heaviest first, so a causal launch is scheduled longest-job-first over ALL heads), and with workgroups
    // dealt round-robin to the 8 XCDs each L2 holds the K / V^T of heads h = xcd (mod 8) only
    const int bi = blockIdx.x / q_heads, head = blockIdx.x - bi * q_heads, kvh = head / group;
    const int64_t q_row0 = qblk[4 * bi + 0];
    const int n_q = qblk[4 * bi + 1];
    const int64_t k_row0 = (int64_t)qblk[4 * bi + 2];
    const int64_t vt_blk0 = (int64_t)qblk[4 * bi + 3];
    const int kv_len_seg = qblk_len[2 * bi + 0];
    const int q_pos0 = qblk_len[2 * bi + 1];
    int kv_len = kv_len_seg;
    if (CAUSAL) kv_len = min(kv_len, q_pos0 + n_q);
    const int n_tiles = (kv_len + 63) >> 6;

    // The padded V^T rows (hd=80: rows 80..95, never staged) are written once: row HD to ONES, the rest to zero.  The PV
    // MFMAs compute those rows anyway, so O^T row HD comes out as sum_k P[k][q] — the softmax denominator, summed over
    // the same bf16 P the numerator uses, rescaled with O, for no instruction at all: the 32 adds per tile it replaces
    // were an eighth of the loop's vector-ALU issue slots, which — not the matrix pipe — bound this kernel.
    constexpr bool ONES = C::DT * 32 > HD;
    if (ONES) {
        for (int e = tid; e < (C::DT * 32 - HD) * C::VROW / 8; e += NTHR) {
            const unsigned v = e < 16 ? 0x3F803F80u : 0u;  // 64 keys x bf16 1.0 = the first 16 8-byte units
            reinterpret_cast<u32x2*>(img_s + K_BYTES + HD * C::VROW)[e] = (u32x2){v, v};
            reinterpret_cast<u32x2*>(img_s + IMG + K_BYTES + HD * C::VROW)[e] = (u32x2){v, v};
        }
    }

    // ---- Q fragments (B operand): lane = query, 8 d per k-step half
    int ql = wave * 32 + lq;
    const bool q_valid = ql < n_q;
    if (!q_valid) ql = n_q - 1;
    const kr_bf16* qp = q + ((int64_t)head * nq_total + q_row0 + ql) * HD + lh * 8;
    bf16x8 qf[C::KS];
#pragma unroll
    for (int s = 0; s < C::KS; ++s) qf[s] = ld8(qp + s * 16);
    // PRESCALE (hd 80, where the registers allow a second accumulator-sized tuple): Q is multiplied by scale * log2(e) ONCE
    // and rounded back to bf16, and the QK^T accumulators START at -m_ref (the lazy reference maximum, a per-query constant
    // held in `cinit`), so that S^T comes out of the matrix pipe as the exp2 argument itself: the 32 v_fma per tile that
    // applied scale and reference were a fifth of the loop's vector-ALU issue slots.  The extra rounding of Q (relative
    // 2^-9 per element) is of the size of the bf16 rounding the HF reference applies to the scores themselves.
    constexpr bool PRESCALE = (HD == 80) && KR_ATTN_PRESCALE;
    if (PRESCALE) {
#pragma unroll
        for (int s = 0; s < C::KS; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[s][j] = f2bf(bf2f(qf[s][j]) * scale_log2e);
    }
    f32x16 cinit;   // PRESCALE: -m_run in every register of the lane (lane = query)
#pragma unroll
    for (int r = 0; r < 16; ++r) cinit[r] = 0.f;
    const int qpos = q_pos0 + wave * 32 + lq;

    const kr_bf16* kbase = k + (int64_t)kvh * k_head_stride + k_row0 * HD;
    const kr_bf16* vbase = vt + (int64_t)kvh * vt_head_stride + vt_blk0 * (int64_t)(HD * 64);

    f32x16 o[C::DT];
#pragma unroll
    for (int t = 0; t < C::DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = PRESCALE ? 0.f : -1e30f, l_run = 0.f;

    // Every thread takes exactly one chunk in every pass, in the loop ALWAYS (a tile index past the end re-loads the last
    // tile, a list index past the end the last chunk: the same bytes to the same place), and a chunk goes to LDS as two
    // 8-byte halves whether it is a K chunk (16-byte aligned) or a V^T chunk (rows of 136 B: 8-byte aligned).  No
    // lane-predicated or tile-count-dependent branch is left around a load or its wait.  With them (r1 / early r2) the
    // compiler had to assume a load into the same registers might still be in flight on the path that skipped the
    // stores, put s_waitcnt vmcnt(0) in front of the LAST load of every tile — i.e. waited for the five loads issued
    // just before it — and the stamped build (-DKR_ATTN_STAMPS) showed 1500-2400 of the ~4300 cycles per tile in
    // "load issue".
    bf16x8 treg[PASSES];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            int idx = p * NTHR + tid;
            idx = idx < T_CH ? idx : T_CH - 1;
            const int key = idx / C::KCH, c = idx - key * C::KCH;
            int kg = t * 64 + key;
            kg = kg < kv_len_seg ? kg : kv_len_seg - 1;
            const kr_bf16* kp = kbase + (int64_t)kg * HD + c * 8;
            const kr_bf16* vp = vbase + (int64_t)t * (HD * 64) + (idx - K_CH) * 8;   // the block in MEMORY order (see vt_lds_piece)
            treg[p] = ld8(idx < K_CH ? kp : vp);
        }
    };
    auto store_tile = [&](char* img) {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            int idx = p * NTHR + tid;
            idx = idx < T_CH ? idx : T_CH - 1;
            const int key = idx / C::KCH, c = idx - key * C::KCH;
            const int off = idx < K_CH ? k_lds_off<HD>(key, c) : K_BYTES + vt_lds_piece<HD>(idx - K_CH, C::VROW);
            const u32x4 w = __builtin_bit_cast(u32x4, treg[p]);
            u32x2* dstp = reinterpret_cast<u32x2*>(img + off);
            dstp[0] = (u32x2){w[0], w[1]};
            dstp[1] = (u32x2){w[2], w[3]};
        }
    };

    if (n_tiles > 0) {
    load_tile(0);
    store_tile(img_s);
    // every load so far (Q fragments, tile 0) has landed; saying so keeps "Q may be in flight" out of the loop header
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    load_tile(n_tiles > 1 ? 1 : 0);
    __syncthreads();
#ifdef KR_ATTN_STAMPS
    unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0, st5 = 0, st6 = 0;
    unsigned long long acc[6] = {0, 0, 0, 0, 0, 0};
#endif
    // P (bf16) and the V^T fragments of a tile; with ROT they live across the barrier (see below)
    bf16x8 vf[C::DT][4], pf[2][2];
    auto load_v = [&](const char* v_s, int dt) {
        const char* vrow = v_s + (dt * 32 + lq) * C::VROW;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int kb = f * 16 + 4 * lh;  // keys kb..kb+3 and kb+8..kb+11
            const u32x2 lo = *reinterpret_cast<const u32x2*>(vrow + kb * 2);
            const u32x2 hi = *reinterpret_cast<const u32x2*>(vrow + (kb + 8) * 2);
            vf[dt][f] = __builtin_bit_cast(bf16x8, (u32x4){lo[0], lo[1], hi[0], hi[1]});
        }
    };
    // ---- O^T += V^T P^T
    auto do_pv = [&](const char* v_s) {
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
            if (dt + VPRE < C::DT) load_v(v_s, dt + VPRE);
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int ss = 0; ss < 2; ++ss)
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[dt][sub * 2 + ss], pf[sub][ss], o[dt], 0, 0, 0);
        }
    };
    // ROT (8-wave workgroups with every V^T fragment prefetched): the two waves a SIMD holds come from the same workgroup and
    // march in step — both in QK^T (matrix pipe contended, vector ALU idle), then both in the softmax (the reverse).  The
    // second half of the waves therefore runs its PV one barrier LATE, from registers: tile t's P and V^T fragments are kept
    // across the barrier and multiplied at the top of the next iteration, so one wave's PV / QK^T MFMAs sit beside the other's
    // softmax.  Nothing else changes: PV(t) still completes before tile t + 1's rescale decision, in program order.
    // MEASURED (r2, same box, bit-identical outputs): 8 x 4900 tokens 1.196-1.200 ms per block against 1.135-1.152 without,
    // 19 276 tokens 2.21 against 2.11 — the late PV is 12 bare MFMAs that no longer hide this wave's exponentials, and that
    // costs more than the staggering gains.  OFF by default (KR_ATTN_ROT).
    constexpr bool ROT = (NW == 8) && (VPRE == C::DT) && KR_ATTN_ROT;
    const bool rot = ROT && __builtin_amdgcn_readfirstlane(wave) >= NW / 2;
    bool pend = false;
    for (int t = 0; t < n_tiles; ++t) {
        KR_STAMP(st0);
        const char* k_s = img_s + (t & 1) * IMG;
        const char* v_s = k_s + K_BYTES;
        if (ROT && rot && pend) {
            do_pv(nullptr);
            pend = false;
        }

        // ---- S^T = K Q^T.  All K fragments of the tile are requested before the first MFMA.
        // (r2: requesting one 32-key half at a time does not lower the register peak, and forcing 3 waves per SIMD
        // with __launch_bounds__ spills into the loop: 1.65 ms per ViT block against 1.42 on the same box.)
        // causal: a wave whose 32 queries all sit before this tile's first key has nothing to add (every score would be
        // masked: p = 0 exactly), it only takes part in the staging — waves 0 and 1 of a block skip its last tile
        // a wave without a valid query (the ragged last query block of a segment: 4900 = 19 x 256 + 36 leaves six of eight waves
        // empty) only takes part in the staging: the matrix pipe and the LDS ports go to the waves that have rows
        // (8-wave workgroups only: the same test in the 4-wave instantiation measured 5 % slower on full blocks)
        if ((!CAUSAL || t * 64 <= q_pos0 + wave * 32 + 31) && (NW == 4 || wave * 32 < n_q)) {
        bf16x8 kf[2][C::KS];
        if (HD == 128) {
            // k_lds_off(32 sub + lq, 2 ks + lh) = 8192 sub + [256 lq + ((lh ^ (lq & 15)) << 4)] ^ (ks << 5): ONE lane
            // constant, and the xor is taken after the image offset is added (bits 5-7 of IMG are 0), so that the
            // compiler cannot hoist eight per-k-step addresses out of the loop and hold them in registers it lacks
            static_assert(HD != 128 || (IMG & 0xe0) == 0, "image offset must leave the swizzle bits alone");
            const int base = (t & 1) * IMG + lq * 256 + ((lh ^ (lq & 15)) << 4);
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) {
                const char* kp = img_s + (base ^ (ks << 5));
                kf[0][ks] = *reinterpret_cast<const bf16x8*>(kp);
                kf[1][ks] = *reinterpret_cast<const bf16x8*>(kp + 32 * 256);
            }
        } else {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int ks = 0; ks < C::KS; ++ks)
                    kf[sub][ks] = *reinterpret_cast<const bf16x8*>(k_s + k_lds_off<HD>(sub * 32 + lq, 2 * ks + lh));
        }
        __builtin_amdgcn_sched_barrier(0);
        KR_STAMP(st1);
#ifdef KR_ATTN_PRIO
        __builtin_amdgcn_s_setprio(1);
#endif
        f32x16 s[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            if (PRESCALE) {
                s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[sub][0], qf[0], cinit, 0, 0, 0);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[sub][r] = 0.f;
                s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[sub][0], qf[0], s[sub], 0, 0, 0);
            }
#pragma unroll
            for (int ks = 1; ks < C::KS; ++ks)
                s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[sub][ks], qf[ks], s[sub], 0, 0, 0);
        }
#ifdef KR_ATTN_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        // V^T fragments of the first VPRE 32-row tiles of O^T are requested here, behind the QK^T MFMAs: their LDS
        // round trips run under the softmax instead of in front of each PV MFMA (the register file has the room since
        // the accumulators stopped being copied: 186 -> ~230 of the 256 two waves per SIMD allow)
#pragma unroll
        for (int dt = 0; dt < VPRE; ++dt) load_v(v_s, dt);
        // ---- mask, online softmax (lane = query; rows = keys).  The softmax is the VALU-bound part of this
        // kernel (PMC: vector ALU ~70 % busy, MFMA 22 %), so: masks only on tiles that need one (wave-uniform
        // test), the scale folded into one fma per score, P converted to bf16 pairwise, and a LAZY running max:
        // O and l are rescaled only when some query's max grew by more than 2^8 since its last rescale — the
        // numerator and the denominator keep using the same (stale) reference, so the quotient is unchanged and
        // bf16(P) keeps its relative precision at values up to 256.
        bool need_mask = t * 64 + 64 > kv_len_seg;
        if (CAUSAL) need_mask |= t * 64 + 63 > q_pos0 + wave * 32;
        if (need_mask) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * 64 + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const bool ok = key < kv_len_seg && (!CAUSAL || key <= qpos);
                    s[sub][r] = ok ? s[sub][r] : -INFINITY;
                }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[sub][r]);
#ifdef KR_ATTN_SHFL
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
#else
        {   // the other half-wave's maximum: one v_permlane32_swap (vector ALU) instead of an LDS round trip
            const unsigned mb = __builtin_bit_cast(unsigned, mx);
            const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
            mx = fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
        }
#endif
        if (!PRESCALE) mx *= scale_log2e;  // scale > 0: max commutes with it
#ifdef KR_ATTN_STAMPS
        asm volatile("" : "+v"(mx));
#endif
        KR_STAMP(st2);
        if (PRESCALE) {
            // the scores ARE s - m_run already: mx is this tile's maximum relative to the reference.  The first tile moves
            // the reference onto its own maximum (o and l are still zero: nothing to scale), later tiles move it only
            // when a query's maximum grew by more than 2^8 — both rare enough to pay 48 extra subtractions there.
            const bool first = t == 0;
            if (first || __any(mx > 8.0f)) {
                const float d = first ? (mx > -INFINITY ? mx : 0.f) : fmaxf(mx, 0.f);
                m_run += d;
#pragma unroll
                for (int r = 0; r < 16; ++r) cinit[r] = -m_run;
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[sub][r] -= d;
                if (!first) {
                    const float alpha = __builtin_amdgcn_exp2f(-d);
                    l_run *= alpha;
#pragma unroll
                    for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
                }
            }
        } else {
            const float m_new = fmaxf(m_run, mx);
            if (__any(m_new - m_run > 8.0f)) {  // first tile (m_run = -1e30), then rarely
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                m_run = m_new;
                l_run *= alpha;
#pragma unroll
                for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            }
        }
        float psum = 0.f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int h8 = 0; h8 < 2; ++h8) {
                f32x8 pv;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    pv[j] = PRESCALE ? __builtin_amdgcn_exp2f(s[sub][h8 * 8 + j])
                                     : __builtin_amdgcn_exp2f(__builtin_fmaf(s[sub][h8 * 8 + j], scale_log2e, -m_run));
                    if (!ONES) psum += pv[j];
                }
                pf[sub][h8] = __builtin_convertvector(pv, bf16x8);
            }
        l_run += psum;
        if (ROT && rot) pend = true;
        else do_pv(v_s);
        }
#ifdef KR_ATTN_STAMPS
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) asm volatile("" : "+v"(o[dt]));
#endif
        KR_STAMP(st3);
        // tile t+1 (in registers since the previous barrier) -> the other image; it was last read during tile
        // t-1, which every wave left at the previous barrier
        store_tile(img_s + ((t + 1) & 1) * IMG);
        KR_STAMP(st4);
        __syncthreads();
        KR_STAMP(st5);
        load_tile(t + 2 < n_tiles ? t + 2 : n_tiles - 1);
        KR_STAMP(st6);
#ifdef KR_ATTN_STAMPS
        acc[0] += st1 - st0; acc[1] += st2 - st1; acc[2] += st3 - st2; acc[3] += st4 - st3; acc[4] += st5 - st4; acc[5] += st6 - st5;
#endif
    }
    if (ROT && rot && pend) do_pv(nullptr);
#ifdef KR_ATTN_STAMPS
    if (blockIdx.x == gridDim.x / 2 && lane == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) kr_attn_dbg[wave * 8 + i] = acc[i];
        kr_attn_dbg[wave * 8 + 6] = (unsigned long long)n_tiles;
    }
#endif
    }  // n_tiles > 0

    // ---- normalise and store: lane = query, 4 consecutive d per register quad
    float l_tot;
    if (ONES) {  // O^T row HD: tile HD/32, row HD%32 = 16 -> register 8 of the lanes lh = 0
        static_assert(!ONES || HD % 32 == 16, "ones row register");
        l_tot = __shfl(o[HD / 32][8], lq, 64);
    } else {
        l_tot = l_run + __shfl_xor(l_run, 32, 64);
    }
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (!q_valid) return;
    kr_bf16* op = out + (q_row0 + ql) * ((int64_t)q_heads * HD) + (int64_t)head * HD;
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int d = dt * 32 + 8 * i + 4 * lh;
            if (d < HD) {
                bf16x4 ov;
#pragma unroll
                for (int j = 0; j < 4; ++j) ov[j] = f2bf(o[dt][4 * i + j] * inv);
                *reinterpret_cast<bf16x4*>(op + d) = ov;
            }
        }
}

// =====================================================================================
// 4 waves x 64 queries: ONE wave per SIMD with the whole register file (VERDICT r3 next #4b; the guide's "4-wave, one-wave-per-SIMD"
// attention structure, cdna_hip_programming.md)
// =====================================================================================
// A workgroup is still 256 queries of one head over the same two-image K / V^T ring, but its 4 waves own 64 queries each (two
// 32-query blocks): the K and V^T fragments a wave reads from LDS feed TWO MFMA chains (half the fragment reads per MFMA), and the
// overlap of softmax and matrix work has to come from inside the wave — QK^T of block 1 and PV of block 0 are independent of
// block 0's / block 1's exponentials, in one basic block for the scheduler to interleave — instead of from a second wave on the
// SIMD.  Same arithmetic per query as attn_varlen_kernel (PRESCALE, ONES row, lazy reference): bit-identical outputs.
template <int HD, bool CAUSAL>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) attn_varlen_q64_kernel(
    const kr_bf16* __restrict__ q, const kr_bf16* __restrict__ k, const kr_bf16* __restrict__ vt, kr_bf16* __restrict__ out,
    const int32_t* __restrict__ qblk, const int32_t* __restrict__ qblk_len, int64_t nq_total, int q_heads, int group,
    int64_t k_head_stride, int64_t vt_head_stride, float scale_log2e) {
    using C = AttnCfg<HD>;
    constexpr int NTHR = 256, QB = 2;
    constexpr int K_CH = 64 * C::KCH, V_CH = HD * 8, T_CH = K_CH + V_CH, PASSES = (T_CH + NTHR - 1) / NTHR;
    constexpr int K_BYTES = 64 * C::KROW, V_BYTES = C::DT * 32 * C::VROW, IMG = K_BYTES + V_BYTES;
    __shared__ __attribute__((aligned(16))) char img_s[2 * IMG];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, lh = lane >> 5;
    const int bi = blockIdx.x / q_heads, head = blockIdx.x - bi * q_heads, kvh = head / group;
    const int64_t q_row0 = qblk[4 * bi + 0];
    const int n_q = qblk[4 * bi + 1];
    const int64_t k_row0 = (int64_t)qblk[4 * bi + 2];
    const int64_t vt_blk0 = (int64_t)qblk[4 * bi + 3];
    const int kv_len_seg = qblk_len[2 * bi + 0];
    const int q_pos0 = qblk_len[2 * bi + 1];
    int kv_len = kv_len_seg;
    if (CAUSAL) kv_len = min(kv_len, q_pos0 + n_q);
    const int n_tiles = (kv_len + 63) >> 6;
    constexpr bool ONES = C::DT * 32 > HD;
    if (ONES) {
        for (int e = tid; e < (C::DT * 32 - HD) * C::VROW / 8; e += NTHR) {
            const unsigned v = e < 16 ? 0x3F803F80u : 0u;
            reinterpret_cast<u32x2*>(img_s + K_BYTES + HD * C::VROW)[e] = (u32x2){v, v};
            reinterpret_cast<u32x2*>(img_s + IMG + K_BYTES + HD * C::VROW)[e] = (u32x2){v, v};
        }
    }
    constexpr bool PRESCALE = (HD == 80) && KR_ATTN_PRESCALE;
    int ql[QB];
    bool q_valid[QB];
    bf16x8 qf[QB][C::KS];
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        ql[b] = wave * 64 + b * 32 + lq;
        q_valid[b] = ql[b] < n_q;
        if (!q_valid[b]) ql[b] = n_q - 1;
        const kr_bf16* qp = q + ((int64_t)head * nq_total + q_row0 + ql[b]) * HD + lh * 8;
#pragma unroll
        for (int s = 0; s < C::KS; ++s) qf[b][s] = ld8(qp + s * 16);
        if (PRESCALE) {
#pragma unroll
            for (int s = 0; s < C::KS; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) qf[b][s][j] = f2bf(bf2f(qf[b][s][j]) * scale_log2e);
        }
    }
    float m_run[QB], l_run[QB];
    f32x16 o[QB][C::DT], cinit[QB];   // cinit: -m_ref of the block's queries in every register (zero without PRESCALE)
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        m_run[b] = PRESCALE ? 0.f : -1e30f;
        l_run[b] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) cinit[b][r] = 0.f;
#pragma unroll
        for (int t = 0; t < C::DT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[b][t][r] = 0.f;
    }
    const kr_bf16* kbase = k + (int64_t)kvh * k_head_stride + k_row0 * HD;
    const kr_bf16* vbase = vt + (int64_t)kvh * vt_head_stride + vt_blk0 * (int64_t)(HD * 64);
    bf16x8 treg[PASSES];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            int idx = p * NTHR + tid;
            idx = idx < T_CH ? idx : T_CH - 1;
            const int key = idx / C::KCH, c = idx - key * C::KCH;
            int kg = t * 64 + key;
            kg = kg < kv_len_seg ? kg : kv_len_seg - 1;
            const kr_bf16* kp = kbase + (int64_t)kg * HD + c * 8;
            const kr_bf16* vp = vbase + (int64_t)t * (HD * 64) + (idx - K_CH) * 8;
            treg[p] = ld8(idx < K_CH ? kp : vp);
        }
    };
    auto store_tile = [&](char* img) {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            int idx = p * NTHR + tid;
            idx = idx < T_CH ? idx : T_CH - 1;
            const int key = idx / C::KCH, c = idx - key * C::KCH;
            const int off = idx < K_CH ? k_lds_off<HD>(key, c) : K_BYTES + vt_lds_piece<HD>(idx - K_CH, C::VROW);
            const u32x4 w = __builtin_bit_cast(u32x4, treg[p]);
            u32x2* dstp = reinterpret_cast<u32x2*>(img + off);
            dstp[0] = (u32x2){w[0], w[1]};
            dstp[1] = (u32x2){w[2], w[3]};
        }
    };
    if (n_tiles > 0) {
        load_tile(0);
        store_tile(img_s);
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        load_tile(n_tiles > 1 ? 1 : 0);
        __syncthreads();
        for (int t = 0; t < n_tiles; ++t) {
            const char* k_s = img_s + (t & 1) * IMG;
            const char* v_s = k_s + K_BYTES;
            if (!CAUSAL || t * 64 <= q_pos0 + wave * 64 + 63) {
                bf16x8 kf[2][C::KS];
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int ks = 0; ks < C::KS; ++ks)
                        kf[sub][ks] = *reinterpret_cast<const bf16x8*>(k_s + k_lds_off<HD>(sub * 32 + lq, 2 * ks + lh));
                f32x16 s[QB][2];
#pragma unroll
                for (int b = 0; b < QB; ++b)
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub) {
                        // the chain starts from the block's persistent -m_ref tuple (PRESCALE) / from zero: no per-tile register fill
                        s[b][sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[sub][0], qf[b][0], cinit[b], 0, 0, 0);
#pragma unroll
                        for (int ks = 1; ks < C::KS; ++ks)
                            s[b][sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[sub][ks], qf[b][ks], s[b][sub], 0, 0, 0);
                    }
                // V^T fragments: read once, used by both query blocks
                bf16x8 vf[C::DT][4];
#pragma unroll
                for (int dt = 0; dt < C::DT; ++dt) {
                    const char* vrow = v_s + (dt * 32 + lq) * C::VROW;
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        const int kb = f * 16 + 4 * lh;
                        const u32x2 lo = *reinterpret_cast<const u32x2*>(vrow + kb * 2);
                        const u32x2 hi = *reinterpret_cast<const u32x2*>(vrow + (kb + 8) * 2);
                        vf[dt][f] = __builtin_bit_cast(bf16x8, (u32x4){lo[0], lo[1], hi[0], hi[1]});
                    }
                }
                bf16x8 pf[QB][2][2];
#pragma unroll
                for (int b = 0; b < QB; ++b) {
                    const int qpos = q_pos0 + wave * 64 + b * 32 + lq;
                    bool need_mask = t * 64 + 64 > kv_len_seg;
                    if (CAUSAL) need_mask |= t * 64 + 63 > q_pos0 + wave * 64 + b * 32;
                    if (need_mask) {
#pragma unroll
                        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int key = t * 64 + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                                const bool ok = key < kv_len_seg && (!CAUSAL || key <= qpos);
                                s[b][sub][r] = ok ? s[b][sub][r] : -INFINITY;
                            }
                    }
                    float mx = -INFINITY;
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[b][sub][r]);
                    {
                        const unsigned mb = __builtin_bit_cast(unsigned, mx);
                        const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
                        mx = fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
                    }
                    if (!PRESCALE) mx *= scale_log2e;
                    if (PRESCALE) {
                        const bool first = t == 0;
                        if (first || __any(mx > 8.0f)) {
                            const float d = first ? (mx > -INFINITY ? mx : 0.f) : fmaxf(mx, 0.f);
                            m_run[b] += d;
#pragma unroll
                            for (int r = 0; r < 16; ++r) cinit[b][r] = -m_run[b];
#pragma unroll
                            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                                for (int r = 0; r < 16; ++r) s[b][sub][r] -= d;
                            if (!first) {
                                const float alpha = __builtin_amdgcn_exp2f(-d);
                                l_run[b] *= alpha;
#pragma unroll
                                for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
                                    for (int r = 0; r < 16; ++r) o[b][dt][r] *= alpha;
                            }
                        }
                    } else {
                        const float m_new = fmaxf(m_run[b], mx);
                        if (__any(m_new - m_run[b] > 8.0f)) {
                            const float alpha = __builtin_amdgcn_exp2f(m_run[b] - m_new);
                            m_run[b] = m_new;
                            l_run[b] *= alpha;
#pragma unroll
                            for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
                                for (int r = 0; r < 16; ++r) o[b][dt][r] *= alpha;
                        }
                    }
                    float psum = 0.f;
#pragma unroll
                    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                        for (int h8 = 0; h8 < 2; ++h8) {
                            f32x8 pv;
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                pv[j] = PRESCALE ? __builtin_amdgcn_exp2f(s[b][sub][h8 * 8 + j])
                                                 : __builtin_amdgcn_exp2f(__builtin_fmaf(s[b][sub][h8 * 8 + j], scale_log2e, -m_run[b]));
                                if (!ONES) psum += pv[j];
                            }
                            pf[b][sub][h8] = __builtin_convertvector(pv, bf16x8);
                        }
                    l_run[b] += psum;
                }
#pragma unroll
                for (int b = 0; b < QB; ++b)
#pragma unroll
                    for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
                        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                            for (int ss = 0; ss < 2; ++ss)
                                o[b][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[dt][sub * 2 + ss], pf[b][sub][ss], o[b][dt], 0, 0, 0);
            }
            store_tile(img_s + ((t + 1) & 1) * IMG);
            __syncthreads();
            load_tile(t + 2 < n_tiles ? t + 2 : n_tiles - 1);
        }
    }
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        float l_tot;
        if (ONES) {
            l_tot = __shfl(o[b][HD / 32][8], lq, 64);
        } else {
            l_tot = l_run[b] + __shfl_xor(l_run[b], 32, 64);
        }
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
        if (q_valid[b]) {
            kr_bf16* op = out + (q_row0 + ql[b]) * ((int64_t)q_heads * HD) + (int64_t)head * HD;
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int d = dt * 32 + 8 * i + 4 * lh;
                    if (d < HD) {
                        bf16x4 ov;
#pragma unroll
                        for (int j = 0; j < 4; ++j) ov[j] = f2bf(o[b][dt][4 * i + j] * inv);
                        *reinterpret_cast<bf16x4*>(op + d) = ov;
                    }
                }
        }
    }
}

#ifdef KR_ATTN_PIPE_EXPERIMENT
// =====================================================================================
// EXPERIMENT (r3, measured, NOT adopted; built only with -DKR_ATTN_PIPE_EXPERIMENT via tools/build_variant.py):
// the same attention, software-pipelined INSIDE each wave
// =====================================================================================
// attn_varlen_kernel runs a tile as a dependent chain per wave — K reads -> QK^T -> max -> exp -> PV — so a wave's own
// MFMAs never sit beside its own exponentials, and its MFMA chains are dependent back to back.  Here the unit is a
// 32-key SUB-tile j and one step of a wave's stream is
//     matrix pipe : PV(j) and QK^T(j + 2), alternating (adjacent MFMAs are independent; PV leads by two while the
//                   K fragments land)
//     vector ALU  : P(j + 1) = bf16(exp2(S(j + 1))) in the MFMA gaps (sched_group_barrier), then mask and maximum of the
//                   S(j + 2) just finished
//     rare branch : the lazy reference moves by d for S(j + 2): O (PV(j) is in it) and l are scaled by 2^-d, the one P
//                   already formed from the old reference whose PV is still to come (P(j + 1)) is scaled with them, d is
//                   taken from S(j + 2) and from the QK^T start tuple
// i.e. scores are produced two sub-tiles ahead of their PV.  Tiles are staged TWO ahead (three LDS images: during
// iteration t the waves read K of tile t + 1 and V^T of tiles t / t + 1 and write tile t + 2), one barrier per tile;
// Q / K / V^T / P layouts, the ones row, PRESCALE and the branch-free staging are attn_varlen_kernel's.  The ISA is what
// was asked for (hd 80: 211 registers, no spill; MFMAs alternating with 4-8 v_exp / v_cvt between them).
// MEASURED (profiles/r03_attn_pipe_ablation.txt; same box, same operands, outputs within the kernel tests' tolerance):
// 8 x 4900 tokens 1.071 ms against 1.068 for attn_varlen_kernel, 19 276 tokens 1.971 against 1.970 — NO gain, and two
// earlier forms (QK^T one sub-tile ahead; the decision taken beside the next step's leading MFMAs) the same.  The
// ablation switches below say why: without exponentials -5 %, without staging and barrier -15 %, without the K / V^T
// fragment reads -10 %, all three -32 % (0.72 ms = the 22 MFMAs per tile alone, 1.5 PFLOP/s of matrix work), and the
// parts ADD whatever their order in the stream, at 0.49 MFMA-busy and a clock that moves little (1.97 - 2.26 GHz over the
// variants: not a DVFS effect).  The loop is not bound by a dependency chain, by issue order or by the two waves of a SIMD
// marching in step: a tile's vector-ALU, LDS and staging work does not run beside its MFMAs on this SIMD in any order
// tried, so only doing less per tile would help: fragment reads shared by 64 queries per wave (one wave per SIMD on 512
// registers) and LDS-DMA staging are the two levers left, ~5 % and ~8 % of this kernel.
// the order of a step's MFMAs: entry i is PV number idx[i] (is_pv) or QK^T k-step idx[i]
template <int NPV, int KS>
struct PipeOrder {
    bool is_pv[NPV + KS] = {};
    int idx[NPV + KS] = {};
    constexpr PipeOrder() {
        int np = 0, nq = 0;
        for (int i = 0; i < NPV + KS; ++i) {
            const bool take_pv = (np < NPV) && (np < 2 || nq >= KS || ((i & 1) == 0));
            is_pv[i] = take_pv;
            idx[i] = take_pv ? np++ : nq++;
        }
    }
};
template <int HD, bool CAUSAL, int NW>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2))) attn_varlen_pipe_kernel(
    const kr_bf16* __restrict__ q, const kr_bf16* __restrict__ k, const kr_bf16* __restrict__ vt, kr_bf16* __restrict__ out,
    const int32_t* __restrict__ qblk, const int32_t* __restrict__ qblk_len, int64_t nq_total, int q_heads, int group,
    int64_t k_head_stride, int64_t vt_head_stride, float scale_log2e) {
    using C = AttnCfg<HD>;
    constexpr int NTHR = NW * 64;
    constexpr int K_CH = 64 * C::KCH, V_CH = HD * 8, T_CH = K_CH + V_CH, PASSES = (T_CH + NTHR - 1) / NTHR;
    constexpr int K_BYTES = 64 * C::KROW, V_BYTES = C::DT * 32 * C::VROW, IMG = K_BYTES + V_BYTES;
    __shared__ __attribute__((aligned(16))) char img_s[3 * IMG];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, lh = lane >> 5;
    const int bi = blockIdx.x / q_heads, head = blockIdx.x - bi * q_heads, kvh = head / group;
    const int64_t q_row0 = qblk[4 * bi + 0];
    const int n_q = qblk[4 * bi + 1];
    const int64_t k_row0 = (int64_t)qblk[4 * bi + 2];
    const int64_t vt_blk0 = (int64_t)qblk[4 * bi + 3];
    const int kv_len_seg = qblk_len[2 * bi + 0];
    const int q_pos0 = qblk_len[2 * bi + 1];
    int kv_len = kv_len_seg;
    if (CAUSAL) kv_len = min(kv_len, q_pos0 + n_q);
    const int n_tiles = (kv_len + 63) >> 6;

    constexpr bool ONES = C::DT * 32 > HD;
    if (ONES) {
        for (int e = tid; e < (C::DT * 32 - HD) * C::VROW / 8; e += NTHR) {
            const unsigned v = e < 16 ? 0x3F803F80u : 0u;
#pragma unroll
            for (int i = 0; i < 3; ++i) reinterpret_cast<u32x2*>(img_s + i * IMG + K_BYTES + HD * C::VROW)[e] = (u32x2){v, v};
        }
    }

    int ql = wave * 32 + lq;
    const bool q_valid = ql < n_q;
    if (!q_valid) ql = n_q - 1;
    const kr_bf16* qp = q + ((int64_t)head * nq_total + q_row0 + ql) * HD + lh * 8;
    bf16x8 qf[C::KS];
#pragma unroll
    for (int s = 0; s < C::KS; ++s) qf[s] = ld8(qp + s * 16);
    constexpr bool PRESCALE = (HD == 80) && KR_ATTN_PRESCALE;
    if (PRESCALE) {
#pragma unroll
        for (int s = 0; s < C::KS; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[s][j] = f2bf(bf2f(qf[s][j]) * scale_log2e);
    }
    f32x16 cinit;   // PRESCALE: -reference in every register (lane = query); else zero
#pragma unroll
    for (int r = 0; r < 16; ++r) cinit[r] = 0.f;
    const int qpos = q_pos0 + wave * 32 + lq;

    const kr_bf16* kbase = k + (int64_t)kvh * k_head_stride + k_row0 * HD;
    const kr_bf16* vbase = vt + (int64_t)kvh * vt_head_stride + vt_blk0 * (int64_t)(HD * 64);

    f32x16 o[C::DT];
#pragma unroll
    for (int t = 0; t < C::DT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;   // m_run: the non-PRESCALE reference

    bf16x8 treg[PASSES];
    auto load_tile = [&](int t) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            int idx = p * NTHR + tid;
            idx = idx < T_CH ? idx : T_CH - 1;
            const int key = idx / C::KCH, c = idx - key * C::KCH;
            int kg = t * 64 + key;
            kg = kg < kv_len_seg ? kg : kv_len_seg - 1;
            const kr_bf16* kp = kbase + (int64_t)kg * HD + c * 8;
            const kr_bf16* vp = vbase + (int64_t)t * (HD * 64) + (idx - K_CH) * 8;   // the block in MEMORY order (see vt_lds_piece)
            treg[p] = ld8(idx < K_CH ? kp : vp);
        }
    };
    auto store_tile = [&](char* img) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            int idx = p * NTHR + tid;
            idx = idx < T_CH ? idx : T_CH - 1;
            const int key = idx / C::KCH, c = idx - key * C::KCH;
            const int off = idx < K_CH ? k_lds_off<HD>(key, c) : K_BYTES + vt_lds_piece<HD>(idx - K_CH, C::VROW);
            const u32x4 w = __builtin_bit_cast(u32x4, treg[p]);
            u32x2* dstp = reinterpret_cast<u32x2*>(img + off);
            dstp[0] = (u32x2){w[0], w[1]};
            dstp[1] = (u32x2){w[2], w[3]};
        }
    };
    auto read_k = [&](const char* k_s, int sub, bf16x8 (&kf)[C::KS]) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks)
            kf[ks] = *reinterpret_cast<const bf16x8*>(k_s + k_lds_off<HD>(sub * 32 + lq, 2 * ks + lh));
    };
    auto read_v = [&](const char* v_s, int sub, bf16x8 (&vf)[C::DT][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
            const char* vrow = v_s + (dt * 32 + lq) * C::VROW;
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                const int kb = (sub * 2 + ss) * 16 + 4 * lh;  // keys kb..kb+3 and kb+8..kb+11
                const u32x2 lo = *reinterpret_cast<const u32x2*>(vrow + kb * 2);
                const u32x2 hi = *reinterpret_cast<const u32x2*>(vrow + (kb + 8) * 2);
                vf[dt][ss] = __builtin_bit_cast(bf16x8, (u32x4){lo[0], lo[1], hi[0], hi[1]});
            }
        }
    };
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    auto mask_scores = [&](f32x16& s, int t, int sub) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = t * 64 + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const bool ok = key < kv_len_seg && (!CAUSAL || key <= qpos);
            s[r] = ok ? s[r] : -INFINITY;
        }
    };
    auto max_scores = [&](const f32x16& s) __attribute__((always_inline)) {
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[r]);
        const unsigned mb = __builtin_bit_cast(unsigned, mx);
        const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
        return fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
    };
    auto exp_scores = [&](const f32x16& s, bf16x8 (&pf)[2]) __attribute__((always_inline)) {
        float psum = 0.f;
#pragma unroll
        for (int h8 = 0; h8 < 2; ++h8) {
            f32x8 pv8;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#ifdef KR_PIPE_NO_EXP
                pv8[j] = s[h8 * 8 + j];
#else
                pv8[j] = PRESCALE ? __builtin_amdgcn_exp2f(s[h8 * 8 + j])
                                  : __builtin_amdgcn_exp2f(__builtin_fmaf(s[h8 * 8 + j], scale_log2e, -m_run));
#endif
                if (!ONES) psum += pv8[j];
            }
            pf[h8] = __builtin_convertvector(pv8, bf16x8);
        }
        l_run += psum;
    };
    // One step j: PV(j) (P = pf_in, V^T fragments vf read one step earlier) and QK^T(j + 2) (sub-tile (tk, subk), into
    // s_acc) on the matrix pipe, alternating; beside them P(j + 1) = exp2(s_prev) into pf_out; then mask and maximum of the
    // finished s_acc and, rarely, the reference move for it: O (PV(j) is in it) and l are scaled by 2^-d, the one P already
    // formed from the old reference whose PV is still to come (pf_out) is scaled with them, and d is taken from s_acc and
    // from the QK^T start tuple.
    constexpr int NPV = 2 * C::DT, N_MFMA = NPV + C::KS;
    auto step = [&](auto MASKC, f32x16& s_prev, f32x16& s_acc, int tk, int subk, bf16x8 (&pf_out)[2], const bf16x8 (&pf_in)[2],
                    const char* k_s, const bf16x8 (&vf)[C::DT][2], const char* v_next, int sub_next,
                    bf16x8 (&vf_next)[C::DT][2]) __attribute__((always_inline)) {
        constexpr bool MASK = decltype(MASKC)::value;
        asm volatile("" : "+v"(s_prev));   // the step's vector work stays inside the step
        bf16x8 kf[C::KS];
#ifdef KR_PIPE_NO_KREAD
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) kf[ks] = qf[ks];
#else
        read_k(k_s, subk, kf);
#endif
#ifdef KR_PIPE_NO_VREAD
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) vf_next[dt][ss] = vf[dt][ss];
#else
        read_v(v_next, sub_next, vf_next);
#endif
        f32x16 acc = PRESCALE ? cinit : zero16;
        // MFMA i of the step (PipeOrder): PV leads by two (its operands are in registers, K fragments are landing), then
        // the chains alternate: adjacent MFMAs never depend on each other
        constexpr PipeOrder<NPV, C::KS> ORD{};
#pragma unroll
        for (int i = 0; i < N_MFMA; ++i) {
            const int n = ORD.idx[i];
            if (ORD.is_pv[i]) {
#ifndef KR_PIPE_NO_PV
                o[n % C::DT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[n % C::DT][n / C::DT], pf_in[n / C::DT], o[n % C::DT], 0, 0, 0);
#endif
            } else {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[n], qf[n], acc, 0, 0, 0);
            }
        }
        exp_scores(s_prev, pf_out);
        s_acc = acc;
        if (MASK) mask_scores(s_acc, tk, subk);
        float mx = max_scores(s_acc);
#pragma unroll
        for (int i = 0; i < N_MFMA; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, ONES ? 3 : 5, 0);
        }
        if (!PRESCALE) mx *= scale_log2e;
        const float grow = PRESCALE ? mx : fmaxf(m_run, mx) - m_run;
        if (__any(grow > 8.0f)) {
            const float d = fmaxf(grow, 0.f);
            const float alpha = __builtin_amdgcn_exp2f(-d);
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
#pragma unroll
            for (int h8 = 0; h8 < 2; ++h8)
#pragma unroll
                for (int j = 0; j < 8; ++j) pf_out[h8][j] = f2bf(bf2f(pf_out[h8][j]) * alpha);
            if (PRESCALE) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    cinit[r] -= d;
                    s_acc[r] -= d;
                }
            } else {
                m_run += d;
            }
        }
    };

    if (n_tiles > 0) {
        load_tile(0);
        store_tile(img_s);
        load_tile(n_tiles > 1 ? 1 : 0);
        store_tile(img_s + IMG);
        load_tile(n_tiles > 2 ? 2 : n_tiles - 1);
        __syncthreads();
        bf16x8 vfa[C::DT][2], vfb[C::DT][2], pfa[2], pfb[2];
        f32x16 s0, s1;
        // ---- prologue: S(0, 0) with the first reference = its own maximum, P(0, 0), S(0, 1) (not yet looked at), V^T(0, 0)
        {
            bf16x8 kf[C::KS];
            read_k(img_s, 0, kf);
            s0 = zero16;
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], s0, 0, 0, 0);
            mask_scores(s0, 0, 0);
            const float mx = max_scores(s0);
            if (PRESCALE) {
                const float d = mx > -INFINITY ? mx : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    cinit[r] = -d;
                    s0[r] -= d;
                }
            } else {
                m_run = mx > -INFINITY ? mx * scale_log2e : 0.f;
            }
            read_k(img_s, 1, kf);
            s1 = PRESCALE ? cinit : zero16;
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], s1, 0, 0, 0);
            exp_scores(s0, pfa);
            mask_scores(s1, 0, 1);
            float mx1 = max_scores(s1);   // the decision for (0, 1): O is still zero, P(0, 0) waits for its PV
            if (!PRESCALE) mx1 *= scale_log2e;
            const float grow = PRESCALE ? mx1 : fmaxf(m_run, mx1) - m_run;
            if (__any(grow > 8.0f)) {
                const float d = fmaxf(grow, 0.f);
                const float alpha = __builtin_amdgcn_exp2f(-d);
                l_run *= alpha;
#pragma unroll
                for (int h8 = 0; h8 < 2; ++h8)
#pragma unroll
                    for (int j = 0; j < 8; ++j) pfa[h8][j] = f2bf(bf2f(pfa[h8][j]) * alpha);
                if (PRESCALE) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        cinit[r] -= d;
                        s1[r] -= d;
                    }
                } else {
                    m_run += d;
                }
            }
            read_v(img_s + K_BYTES, 0, vfa);
        }
        auto tile_body = [&](auto MASKC, int t) __attribute__((always_inline)) {
            const int cur = t % 3, nx = (t + 1) % 3, nn = (t + 2) % 3;
            const char* v_s = img_s + cur * IMG + K_BYTES;
            const char* k_s = img_s + nx * IMG;
            // PV(t, 0) with QK^T(t + 1, 0); P of (t, 1); V^T(t, 1) for the next step
            step(MASKC, s1, s0, t + 1, 0, pfb, pfa, k_s, vfa, v_s, 1, vfb);
            // PV(t, 1) with QK^T(t + 1, 1); P of (t + 1, 0); V^T(t + 1, 0).  Past the last tile the image
            // holds that tile again: those scores are never used, and no tile-count branch sits in the loop
            step(MASKC, s0, s1, t + 1, 1, pfa, pfb, k_s, vfb, k_s + K_BYTES, 0, vfa);
            // tile t + 2 (in registers since the previous barrier) -> the third image, last read (V^T of tile t - 1)
            // before the previous barrier
#ifndef KR_PIPE_NO_STAGE
            store_tile(img_s + nn * IMG);
#endif
#ifndef KR_PIPE_NO_BARRIER
            __syncthreads();
#endif
#ifndef KR_PIPE_NO_STAGE
            load_tile(t + 3 < n_tiles ? t + 3 : n_tiles - 1);
#endif
        };
        // iteration t forms the scores of tile t + 1: plain while every key of it is visible to every query of the block
        int t_plain = kv_len_seg >> 6;
        if (CAUSAL) t_plain = min(t_plain, (q_pos0 + 1) >> 6);
        t_plain = min(t_plain - 1, n_tiles);
        int t = 0;
        for (; t < t_plain; ++t) tile_body(std::false_type{}, t);
        for (; t < n_tiles; ++t) tile_body(std::true_type{}, t);
    }

    float l_tot;
    if (ONES) {
        static_assert(!ONES || HD % 32 == 16, "ones row register");
        l_tot = __shfl(o[HD / 32][8], lq, 64);
    } else {
        l_tot = l_run + __shfl_xor(l_run, 32, 64);
    }
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    if (!q_valid) return;
    kr_bf16* op = out + (q_row0 + ql) * ((int64_t)q_heads * HD) + (int64_t)head * HD;
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int d = dt * 32 + 8 * i + 4 * lh;
            if (d < HD) {
                bf16x4 ov;
#pragma unroll
                for (int j = 0; j < 4; ++j) ov[j] = f2bf(o[dt][4 * i + j] * inv);
                *reinterpret_cast<bf16x4*>(op + d) = ov;
            }
        }
}
#endif  // KR_ATTN_PIPE_EXPERIMENT

// =====================================================================================
// decode: rotary + KV append for one new token per sequence
// =====================================================================================
__global__ void __launch_bounds__(256) decode_qkv_prep_kernel(const kr_bf16* __restrict__ qkv,
                                                              const float* __restrict__ inv_freq,
                                                              const int32_t* __restrict__ ctx_len,
                                                              const int32_t* __restrict__ rope_delta,
                                                              kr_bf16* __restrict__ q_out, kr_bf16* __restrict__ kcache,
                                                              kr_bf16* __restrict__ vtcache, int heads, int kv_heads,
                                                              int hd, int layer, int batch, int s_max) {
    const int b = blockIdx.x;
    const int pos = ctx_len[b];
    const float rp = (float)(pos + rope_delta[b]);
    const int half = hd >> 1;
    const int qkv_dim = (heads + 2 * kv_heads) * hd;
    const kr_bf16* row = qkv + (int64_t)b * qkv_dim;
    const int64_t kv_base = ((int64_t)layer * batch + b) * kv_heads;
    // rotary on q heads and k heads: one thread per (head, i < hd/2)
    for (int e = threadIdx.x; e < (heads + kv_heads) * half; e += blockDim.x) {
        const int h = e / half, i = e - h * half;
        const float ang = rp * inv_freq[i];
        // HF casts cos/sin to the activation dtype (TF:modeling_qwen2_vl.py:169)
        const float c = bfround(cosf(ang)), s = bfround(sinf(ang));
        const float x0 = bfbits2f(row[h * hd + i]), x1 = bfbits2f(row[h * hd + i + half]);
        const __bf16 y0 = f2bf(x0 * c - x1 * s), y1 = f2bf(x1 * c + x0 * s);
        kr_bf16* dst;
        if (h < heads) {
            dst = q_out + ((int64_t)b * heads + h) * hd;
        } else {
            dst = kcache + ((kv_base + (h - heads)) * s_max + pos) * hd;
        }
        dst[i] = __builtin_bit_cast(kr_bf16, y0);
        dst[i + half] = __builtin_bit_cast(kr_bf16, y1);
    }
    // V -> transposed cache column
    const kr_bf16* vrow = row + (heads + kv_heads) * hd;
    for (int e = threadIdx.x; e < kv_heads * hd; e += blockDim.x) {
        const int h = e / hd, d = e - h * hd;
        vtcache[((kv_base + h) * (s_max >> 6) + (pos >> 6)) * (hd * 64) + kr_vt_off(d, pos & 63, hd)] = vrow[e];
    }
}

// =====================================================================================
// decode attention, v_mfma_f32_16x16x32_bf16, K and V^T straight from HBM to registers
// =====================================================================================
// grid = (n_split, kv_heads, batch), 4 waves per block; wave `part` = split*4 + wave walks the
// 64-key blocks part, part + n_part, ...  Heads of the GQA group sit on the 16 MFMA columns.
//   S[key][g]   : A = K rows (keys permuted so that lane group fg ends up with keys 8fg..8fg+7)
//                 B = Q^T
//   O^T[d][g]  += A = V^T (16 B = 8 consecutive keys per lane), B = P^T (the S accumulators)
// Partials (m, l, O) go to the fp32 workspace; attn_decode_combine_kernel merges them.
template <int HD>
__global__ void __launch_bounds__(256) attn_decode_kernel(const kr_bf16* __restrict__ q, const kr_bf16* __restrict__ kcache,
                                                          const kr_bf16* __restrict__ vtcache,
                                                          const int32_t* __restrict__ ctx_len, float* __restrict__ ws,
                                                          int heads, int kv_heads, int layer, int batch, int s_max,
                                                          float scale_log2e) {
    constexpr int DT = HD / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int kvh = blockIdx.y, b = blockIdx.z;
    const int group = heads / kv_heads;
    const int n_part = gridDim.x * 4;
    const int part = blockIdx.x * 4 + wave;
    const int ctx = ctx_len[b] + 1;
    const int nb = (ctx + 63) >> 6;

    const int g = fr < group ? fr : 0;
    const kr_bf16* qp = q + ((int64_t)b * heads + kvh * group + g) * HD + fg * 32;
    bf16x8 qf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) qf[i] = ld8(qp + i * 8);

    const int64_t kv_base = ((int64_t)layer * batch + b) * kv_heads + kvh;
    const kr_bf16* kc = kcache + kv_base * s_max * HD;
    const kr_bf16* vc = vtcache + kv_base * (int64_t)(s_max >> 6) * (HD * 64);

    f32x4 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) o[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -1e30f, l_run = 0.f;

    for (int blk = part; blk < nb; blk += n_part) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int key0 = blk * 64 + hf * 32;
            if (key0 >= ctx) break;  // wave-uniform
            // K fragments: tile kt, row r holds key 8(r>>2) + 4kt + (r&3)
            bf16x8 kf[2][4];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const kr_bf16* kp = kc + (int64_t)(key0 + 8 * (fr >> 2) + 4 * kt + (fr & 3)) * HD + fg * 32;
#pragma unroll
                for (int i = 0; i < 4; ++i) kf[kt][i] = ld8_nt(kp + i * 8);
            }
            // V^T fragments: row d = dt*16 + fr, keys key0 + 8fg .. +7
            bf16x8 vf[DT];
            const kr_bf16* vp = vc + (int64_t)blk * (HD * 64) + hf * (HD * 32) + fg * 8;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) vf[dt] = ld8_nt(vp + (dt * 16 + fr) * 32);

            f32x4 s[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 4; ++i) s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][i], qf[i], s[kt], 0, 0, 0);
            }
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = key0 + 8 * fg + 4 * kt + r;
                    const float v = key < ctx ? s[kt][r] * scale_log2e : -INFINITY;
                    s[kt][r] = v;
                    mx = fmaxf(mx, v);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            bf16x8 pf;
            float psum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const __bf16 pb = f2bf(__builtin_amdgcn_exp2f(s[kt][r] - m_new));
                    psum += bf2f(pb);
                    pf[kt * 4 + r] = pb;
                }
            l_run = l_run * alpha + psum;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[dt], pf, o[dt], 0, 0, 0);
            }
        }
    }
    // l: sum the four key groups of each head column
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    if (fr < group) {
        float* w = ws + (((int64_t)b * heads + kvh * group + fr) * n_part + part) * (HD + 2);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<f32x4*>(w + dt * 16 + fg * 4) = o[dt];
        if (fg == 0) {
            w[HD] = m_run;
            w[HD + 1] = l_run;
        }
    }
}

// grid = batch*heads blocks of HD threads
__global__ void attn_decode_combine_kernel(const float* __restrict__ ws, kr_bf16* __restrict__ out, int n_part, int hd) {
    const int bh = blockIdx.x, d = threadIdx.x;
    const float* w = ws + (int64_t)bh * n_part * (hd + 2);
    float mx = -1e30f;
    for (int p = 0; p < n_part; ++p) mx = fmaxf(mx, w[p * (hd + 2) + hd]);
    float acc = 0.f, l = 0.f;
    for (int p = 0; p < n_part; ++p) {
        const float sc = __builtin_amdgcn_exp2f(w[p * (hd + 2) + hd] - mx);
        l += w[p * (hd + 2) + hd + 1] * sc;
        acc += w[p * (hd + 2) + d] * sc;
    }
    out[(int64_t)bh * hd + d] = __builtin_bit_cast(kr_bf16, f2bf(l > 0.f ? acc / l : 0.f));
}

// scattered K / V^T append (standalone operator; the engine uses the blocked prep kernels)
__global__ void __launch_bounds__(256) kv_append_kernel(const kr_bf16* __restrict__ k, const kr_bf16* __restrict__ v,
                                                        int64_t row_stride, const int32_t* __restrict__ tok_seq,
                                                        const int32_t* __restrict__ tok_pos, kr_bf16* __restrict__ kcache,
                                                        kr_bf16* __restrict__ vtcache, int64_t n, int kv_heads, int hd,
                                                        int layer, int batch, int s_max) {
    const int64_t total = n * kv_heads * hd;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % hd);
        const int64_t t = i / hd;
        const int h = (int)(t % kv_heads);
        const int64_t tok = t / kv_heads;
        const int sq = tok_seq[tok], pos = tok_pos[tok];
        const int64_t base = ((int64_t)layer * batch + sq) * kv_heads + h;
        kcache[(base * s_max + pos) * hd + d] = k[tok * row_stride + h * hd + d];
        vtcache[(base * (s_max >> 6) + (pos >> 6)) * (hd * 64) + kr_vt_off(d, pos & 63, hd)] = v[tok * row_stride + h * hd + d];
    }
}

}  // namespace

// =====================================================================================
// C-ABI
// =====================================================================================
static int launch_rope_inplace(kr_bf16* x, const float* cos, const float* sin, int64_t n, int heads, int hd,
                               int64_t row_stride, kr_stream s, const char* who) {
    KR_CHECK_ARG(x && cos && sin && n >= 0 && heads > 0 && hd > 0 && hd % 16 == 0 && (row_stride & 7) == 0 &&
                     row_stride >= (int64_t)heads * hd,
                 "%s: bad args", who);
    if (n == 0) return KR_OK;
    const int64_t total = n * heads * (hd >> 4);
    int grid = (int)((total + 255) / 256);
    if (grid > 8192) grid = 8192;
    rope_inplace_kernel<<<grid, 256, 0, kr_hs(s)>>>(x, cos, sin, n, heads, hd, row_stride);
    KR_CHECK_LAUNCH();
    return KR_OK;
}

extern "C" int kr_rope2d_vision(kr_bf16* x, const float* cos, const float* sin, int64_t n, int heads, int hd,
                                int64_t row_stride, kr_stream s) {
    return launch_rope_inplace(x, cos, sin, n, heads, hd, row_stride, s, "kr_rope2d_vision");
}

extern "C" int kr_mrope(kr_bf16* x, const float* cos, const float* sin, int64_t n, int heads, int hd,
                        int64_t row_stride, kr_stream s) {
    return launch_rope_inplace(x, cos, sin, n, heads, hd, row_stride, s, "kr_mrope");
}

extern "C" int kr_qkv_prep(const kr_bf16* qkv, int64_t ld_qkv, int q_off, int k_off, int v_off, const float* cos,
                           const float* sin, const int32_t* blk_tok0, const int32_t* blk_ntok,
                           const int64_t* blk_k_row0, const int64_t* blk_vt_blk, int n_blk, kr_bf16* q_out,
                           int64_t q_head_stride, kr_bf16* k_out, int64_t k_head_stride, kr_bf16* vt_out,
                           int64_t vt_head_stride, int q_heads, int kv_heads, int hd, kr_stream s) {
    KR_CHECK_ARG(qkv && cos && sin && blk_tok0 && blk_ntok && blk_k_row0 && blk_vt_blk && q_out && k_out && vt_out,
                 "kr_qkv_prep: null pointer");
    KR_CHECK_ARG(hd == 80 || hd == 128, "kr_qkv_prep: hd=%d (only 80, 128)", hd);
    KR_CHECK_ARG((ld_qkv & 7) == 0 && (q_off & 7) == 0 && (k_off & 7) == 0 && (v_off & 7) == 0, "kr_qkv_prep: alignment");
    if (n_blk == 0) return KR_OK;
    const int n_slots = q_heads + kv_heads;
    dim3 grid(n_blk, (n_slots + PREP_G - 1) / PREP_G);
    if (hd == 80)
        qkv_prep_kernel<80><<<grid, 256, 0, kr_hs(s)>>>(qkv, ld_qkv, q_off, k_off, v_off, cos, sin, blk_tok0, blk_ntok,
                                                        blk_k_row0, blk_vt_blk, q_out, q_head_stride, k_out,
                                                        k_head_stride, vt_out, vt_head_stride, q_heads, n_slots);
    else
        qkv_prep_kernel<128><<<grid, 256, 0, kr_hs(s)>>>(qkv, ld_qkv, q_off, k_off, v_off, cos, sin, blk_tok0, blk_ntok,
                                                         blk_k_row0, blk_vt_blk, q_out, q_head_stride, k_out,
                                                         k_head_stride, vt_out, vt_head_stride, q_heads, n_slots);
    KR_CHECK_LAUNCH();
    return KR_OK;
}

static int attn_varlen_impl(const kr_bf16* q, const kr_bf16* k, const kr_bf16* vt, kr_bf16* out, const int32_t* qblk,
                            const int32_t* qblk_len, int n_qblk, int64_t nq_total, int q_heads, int kv_heads, int hd,
                            int64_t k_head_stride, int64_t vt_head_stride, float scale, int causal, int q_block, kr_stream s) {
    KR_CHECK_ARG(q && k && vt && out && qblk && qblk_len, "kr_attn_varlen: null pointer");
    KR_CHECK_ARG(hd == 80 || hd == 128, "kr_attn_varlen: hd=%d (only 80, 128)", hd);
    KR_CHECK_ARG(q_heads > 0 && kv_heads > 0 && q_heads % kv_heads == 0, "kr_attn_varlen: heads");
    KR_CHECK_ARG(q_block == 128 || q_block == 256, "kr_attn_varlen: q_block=%d (128 or 256)", q_block);
    if (n_qblk == 0) return KR_OK;
    KR_CHECK_ARG((int64_t)n_qblk * q_heads < (1ll << 31), "kr_attn_varlen: %d query blocks x %d heads", n_qblk, q_heads);
    dim3 grid((unsigned)(n_qblk * q_heads));
    const float sl = scale * 1.4426950408889634f;
    const int group = q_heads / kv_heads;
#ifdef KR_ATTN_PIPE_EXPERIMENT
    const char* penv = getenv("KARANTA_ATTN_PIPE");   // experiment build: 0 = the product loop (same-library A/B)
    const int pipe = penv ? atoi(penv) : 1;
#define KR_LAUNCH_ATTN(HD_, C_, NW_)                                                                                       \
    do {                                                                                                                   \
        if (pipe)                                                                                                          \
            attn_varlen_pipe_kernel<HD_, C_, NW_><<<grid, NW_ * 64, 0, kr_hs(s)>>>(q, k, vt, out, qblk, qblk_len, nq_total,     \
                                                                                   q_heads, group, k_head_stride,          \
                                                                                   vt_head_stride, sl);                    \
        else                                                                                                               \
            attn_varlen_kernel<HD_, C_, NW_><<<grid, NW_ * 64, 0, kr_hs(s)>>>(q, k, vt, out, qblk, qblk_len, nq_total,          \
                                                                              q_heads, group, k_head_stride,               \
                                                                              vt_head_stride, sl);                         \
    } while (0)
#else
#define KR_LAUNCH_ATTN(HD_, C_, NW_)                                                                                       \
    attn_varlen_kernel<HD_, C_, NW_><<<grid, NW_ * 64, 0, kr_hs(s)>>>(q, k, vt, out, qblk, qblk_len, nq_total, q_heads, group, \
                                                                      k_head_stride, vt_head_stride, sl)
#endif
    static const int q64_default = [] { const char* e = getenv("KARANTA_ATTN_Q64"); return e ? atoi(e) : 0; }();
    const char* q64e = getenv("KARANTA_ATTN_Q64_NOW");   // tools/attn_microbench.py: same-process A/B
    const bool q64 = q_block == 256 && (q64e ? atoi(q64e) != 0 : q64_default != 0);
#define KR_LAUNCH_ATTN_Q(HD_, C_)                                                                                          \
    do {                                                                                                                   \
        if (q64 && HD_ == 80) /* hd 128: two query blocks' accumulators do not fit 512 registers (spills) */               \
            attn_varlen_q64_kernel<80, C_><<<grid, 256, 0, kr_hs(s)>>>(q, k, vt, out, qblk, qblk_len, nq_total, q_heads, group,  \
                                                                       k_head_stride, vt_head_stride, sl);                 \
        else if (q_block == 256) KR_LAUNCH_ATTN(HD_, C_, 8);                                                               \
        else KR_LAUNCH_ATTN(HD_, C_, 4);                                                                                   \
    } while (0)
    if (hd == 80) {
        if (causal) KR_LAUNCH_ATTN_Q(80, true); else KR_LAUNCH_ATTN_Q(80, false);
    } else {
        if (causal) KR_LAUNCH_ATTN_Q(128, true); else KR_LAUNCH_ATTN_Q(128, false);
    }
#undef KR_LAUNCH_ATTN_Q
#undef KR_LAUNCH_ATTN
    KR_CHECK_LAUNCH();
    return KR_OK;
}

extern "C" int kr_attn_varlen(const kr_bf16* q, const kr_bf16* k, const kr_bf16* vt, kr_bf16* out, const int32_t* qblk,
                              const int32_t* qblk_len, int n_qblk, int64_t nq_total, int q_heads, int kv_heads, int hd,
                              int64_t k_head_stride, int64_t vt_head_stride, float scale, int causal, kr_stream s) {
    return attn_varlen_impl(q, k, vt, out, qblk, qblk_len, n_qblk, nq_total, q_heads, kv_heads, hd, k_head_stride, vt_head_stride,
                            scale, causal, 128, s);
}

extern "C" int kr_attn_varlen_q(const kr_bf16* q, const kr_bf16* k, const kr_bf16* vt, kr_bf16* out, const int32_t* qblk,
                                const int32_t* qblk_len, int n_qblk, int64_t nq_total, int q_heads, int kv_heads, int hd,
                                int64_t k_head_stride, int64_t vt_head_stride, float scale, int causal, int q_block, kr_stream s) {
    return attn_varlen_impl(q, k, vt, out, qblk, qblk_len, n_qblk, nq_total, q_heads, kv_heads, hd, k_head_stride, vt_head_stride,
                            scale, causal, q_block, s);
}

extern "C" int kr_kv_append(const kr_bf16* k, const kr_bf16* v, int64_t row_stride, const int32_t* tok_seq,
                            const int32_t* tok_pos, kr_bf16* kcache, kr_bf16* vtcache, int64_t n, int kv_heads, int hd,
                            int layer, int batch, int s_max, kr_stream s) {
    KR_CHECK_ARG(k && v && tok_seq && tok_pos && kcache && vtcache && s_max % 64 == 0, "kr_kv_append: bad args");
    if (n == 0) return KR_OK;
    const int64_t total = n * kv_heads * hd;
    int grid = (int)((total + 255) / 256);
    if (grid > 8192) grid = 8192;
    kv_append_kernel<<<grid, 256, 0, kr_hs(s)>>>(k, v, row_stride, tok_seq, tok_pos, kcache, vtcache, n, kv_heads, hd,
                                                 layer, batch, s_max);
    KR_CHECK_LAUNCH();
    return KR_OK;
}

extern "C" int kr_decode_qkv_prep(const kr_bf16* qkv, const float* inv_freq, const int32_t* ctx_len,
                                  const int32_t* rope_delta, kr_bf16* q_out, kr_bf16* kcache, kr_bf16* vtcache, int batch,
                                  int heads, int kv_heads, int hd, int layer, int s_max, kr_stream s) {
    KR_CHECK_ARG(qkv && inv_freq && ctx_len && rope_delta && q_out && kcache && vtcache, "kr_decode_qkv_prep: null");
    KR_CHECK_ARG(batch > 0 && s_max % 64 == 0 && hd % 2 == 0, "kr_decode_qkv_prep: bad sizes");
    decode_qkv_prep_kernel<<<batch, 256, 0, kr_hs(s)>>>(qkv, inv_freq, ctx_len, rope_delta, q_out, kcache, vtcache, heads,
                                                        kv_heads, hd, layer, batch, s_max);
    KR_CHECK_LAUNCH();
    return KR_OK;
}

extern "C" int kr_attn_decode_gqa(const kr_bf16* q, const kr_bf16* kcache, const kr_bf16* vtcache, const int32_t* ctx_len,
                                  kr_bf16* out, float* workspace, int batch, int heads, int kv_heads, int hd, int layer,
                                  int s_max, int n_split, float scale, kr_stream s) {
    KR_CHECK_ARG(q && kcache && vtcache && ctx_len && out && workspace, "kr_attn_decode_gqa: null pointer");
    KR_CHECK_ARG(hd == 128, "kr_attn_decode_gqa: hd=%d (only 128)", hd);
    KR_CHECK_ARG(heads % kv_heads == 0 && heads / kv_heads <= 16, "kr_attn_decode_gqa: GQA group must be <= 16");
    KR_CHECK_ARG(batch > 0 && n_split > 0 && s_max % 64 == 0, "kr_attn_decode_gqa: bad sizes");
    dim3 grid(n_split, kv_heads, batch);
    attn_decode_kernel<128><<<grid, 256, 0, kr_hs(s)>>>(q, kcache, vtcache, ctx_len, workspace, heads, kv_heads, layer,
                                                        batch, s_max, scale * 1.4426950408889634f);
    KR_CHECK_LAUNCH();
    attn_decode_combine_kernel<<<batch * heads, hd, 0, kr_hs(s)>>>(workspace, out, n_split * 4, hd);
    KR_CHECK_LAUNCH();
    return KR_OK;
}
