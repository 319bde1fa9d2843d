// Decode-step kernels (one new token per live sequence): 5 launches per decoder layer + 2 per step.
//
//   dec_linear_kernel   y[M<=16, N] = epi(x W^T): the weight matrix is read from HBM exactly once, as
//                       one linear stream — weights are stored PACKED as [N/16][K/32][4][16][8]: one
//                       (16 x 32) block is an MFMA fragment set in lane order, so every wave-level load
//                       is 1 KiB of contiguous memory (8 full 128-byte lines); K is split over
//                       the 4 / 8 / 16 waves of a workgroup (narrow layers get more waves per workgroup
//                       instead of a cross-workgroup split: every fence / atomic hop between workgroups
//                       costs microseconds, measured, and a decode step is a chain of 142 launches).
//                       An optional cross-workgroup split-K with a deterministic in-launch reduction
//                       (slabs + arrival counter, agent-scope release/acquire) is kept for wide-K shapes.
//                       Fused prologues: Qwen2VLRMSNorm, or the merge of the attention split partials.
//                       Fused epilogues:
//                         PLAIN   (+bias, +residual, bf16 or fp32 out)        o_proj, down_proj
//                         SILU    silu(gate)*up                              gate/up projection
//                         ROPE_KV +bias, M-RoPE, q -> q buffer, k/v -> KV cache (V transposed)
//                         ARGMAX  per-workgroup (max, index) partials [+ fp32 logits]   lm_head
//   attn_decode2_kernel split-KV MFMA attention; the split partials are merged either by the consumer
//                       (the o_proj prologue, out == NULL: the engine's path) or in-launch by the
//                       last-arriving workgroup of each (sequence, kv head).
//   sample_greedy_kernel final argmax over the lm_head partials, token history, EOS bookkeeping,
//                       next-token embedding gather, context advance.  (The rotary table of every
//                       decode position is built once per request on the host.)
#include "kr_common.h"

namespace {

// -DKR_EXPERIMENTS (csrc/tools/build_variant.py): the measured-and-not-adopted decode experiments of rounds 1-2 — Infinity-Cache
// prefetch workgroups riding on the qkv launch, the fast-residual mode (per-head o_proj with float atomics: dec_oproj_heads_kernel,
// f32 x rows in the wide / narrow kernels), the in-launch split-KV merge of the attention kernel.  The shipped library holds the
// product path and the general dec_linear_kernel fallback only; the experiment entry points are declared in
// include/karanta_hip_experiments.h.
#ifdef KR_EXPERIMENTS
constexpr bool KR_EXP = true;
#else
constexpr bool KR_EXP = false;
#endif

constexpr int DEPI_PLAIN = 0, DEPI_SILU = 1, DEPI_ROPE_KV = 2, DEPI_ARGMAX = 3, DEPI_SILU8 = 4;

struct DecLinArgs {
    const kr_bf16* x; int64_t ldx;
    const kr_bf16* wp;                 // packed weights
    const kr_bf16* bias;
    const kr_bf16* norm_w; float norm_eps;
    const kr_bf16* residual; int64_t ldr;
    kr_bf16* out; float* out_f32; int64_t ldc;
    int M, N, K, ksplit;
    int groups;                        // work items (tile groups); a workgroup walks g, g + gridDim.x, ... (ksplit == 1)
    float* ws; int* counters;          // split-K slabs [groups][ksplit][NT][256] f32, arrival counters [groups]
    // x = merge of attention partials [M][heads][n_split][132] f32 (o[128], m, l, 2 pad) (xmode 3)
    const float* attn_ws; int attn_split;
    int xmode;                         // 0: x straight from global, 1: LDS stage, 2: LDS stage + RMSNorm, 3: attention merge
    // ROPE_KV
    const float* cs_table;             // [M][cs_stride][128]: cos[0..64) then sin[0..64) per decode position (bf16 values)
    const int32_t* prompt_len; int cs_stride;
    const int32_t* ctx_len;
    kr_bf16* q_out; kr_bf16* kcache; kr_bf16* vtcache;
    int heads, kv_heads, s_max;
    // ARGMAX
    float* amax_val; int32_t* amax_idx;  // [M][gridDim.x]
    // wide kernel: launch geometry as kernel arguments (gridDim / blockDim come from the dispatch packet through
    // a dependent global load, ~1 us on the critical path of a 10 us kernel)
    int wide_blocks, wide_waves;
    // narrow kernel: deferred split-K.  `part_in` [PKS][M][K] f32 partial sums of the previous down_proj that the
    // x prologue adds to x (the residual stream); workgroup 0 writes the sum to x_out (a different buffer than x).
    // PARTIAL epilogue: out_f32 = [ksplit][M][ldc] f32 slabs, no bias / residual.
    const float* part_in; kr_bf16* x_out; int64_t ldxo;
    // fp8 (e4m3fn) weights: one f32 scale per output row (interleaved like the rows for SILU8), applied to the
    // accumulators at the top of every epilogue; NULL for bf16 weights
    const float* w_scale;
    // f32 residual accumulator (fast-residual mode): the narrow NORM kernel's workgroup 0 also stores x_new as f32
    // (x_out_f32, ldxf) — the buffer the per-head o_proj adds into with float atomics; the wide kernel then reads its x
    // rows from that f32 buffer (template XF32) and workgroup 0 stores their bf16 rounding to wide_x_out.
    float* x_out_f32; int64_t ldxf;
    kr_bf16* wide_x_out; int64_t wide_ldxo;
    int x_is_f32;                      // wide kernel: `x` points to f32 rows (ldx in floats)
    // narrow kernel: workgroups blockIdx.x >= groups are PREFETCHERS — they touch [pf_ptr, pf_ptr + pf_bytes) with
    // plain loads (the range lands in the memory-side Infinity Cache) on CUs the launch would otherwise leave idle
    const char* pf_ptr; int64_t pf_bytes; int pf_blocks;
    int part_rows;                     // rows per slab of part_in (0: = M)
    int n_part;                        // slabs in part_in (1 or 2)
    // ONE-slab mode of the deferred split (kr_decode_slab_next): the PARTIAL epilogue ADDS its K range into slab 0 with
    // global_atomic_add_f32 (two addends onto zero: the sum does not depend on their order), and a narrow launch may
    // carry a zeroing job for the accumulator the NEXT down_proj will add into
    int part_atomic;
    float* zero_ptr; int zero_n16;
    // SILU8 output in the XP layout of 17..32-row batches (kr_common.h, kr_xp_byte_offset): the down_proj launch of such a
    // batch (kr_linear_decode32) reads its x fragments as whole cache lines
    int out_xp;
};

// One 64-wide K chunk of a 16-row weight tile in registers, and where its operands sit.
//  bf16 : 2 KiB per chunk, two 16-byte loads per lane; lane (r, g) holds k = 32h + 8g .. +7 of k-step h
//  fp8  : 1 KiB per chunk, ONE 16-byte load per lane; lane (r, g) holds k = 16g .. 16g+15, k-step h takes 16g + 8h .. +7
//         (x is read in the same order, a dot product does not care), converted to bf16 in registers
//         (v_cvt_scalef32_pk_bf16_fp8: every e4m3 value is exact in bf16; the per-row scale is applied in the epilogue)
template <bool W8> struct WChunk;
template <> struct WChunk<false> {
    static constexpr int BYTES = 2048;
    bf16x8 v[2];
    __device__ __forceinline__ void load(const char* p, int64_t c) {
        v[0] = ld8_nt(reinterpret_cast<const kr_bf16*>(p + c * BYTES));
        v[1] = ld8_nt(reinterpret_cast<const kr_bf16*>(p + c * BYTES + 1024));
    }
    __device__ __forceinline__ bf16x8 frag(int h) const { return v[h]; }
    static __device__ __forceinline__ int x_byte(int h, int fg) { return h * 64 + fg * 16; }  // inside the chunk's 128 B of x
};
template <> struct WChunk<true> {
    static constexpr int BYTES = 1024;
    u32x4 q;
    __device__ __forceinline__ void load(const char* p, int64_t c) {
        q = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + c * BYTES));
    }
    __device__ __forceinline__ bf16x8 frag(int h) const {
        // each conversion yields two bf16 packed in one register; they are moved as 32-bit words (element-wise
        // extraction of the builtin's 2 x bf16 result is mis-lowered by this compiler: both halves read the low one)
        u32x4 o;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int w = (int)q[2 * h + i];
            o[2 * i + 0] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, false));
            o[2 * i + 1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, true));
        }
        return __builtin_bit_cast(bf16x8, o);
    }
    static __device__ __forceinline__ int x_byte(int h, int fg) { return fg * 32 + h * 16; }
};

__device__ __forceinline__ void apply_w_scale(const float* w_scale, int n, f32x4& acc) {
    if (w_scale) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(w_scale + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] *= sc[j];
    }
}

__device__ __forceinline__ void better(float& bv, int& bi, float v, int i) {
    if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
}

template <int NT, int EPI, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) dec_linear_kernel(const DecLinArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // chunks (64 k) in flight per wave: 16 x 16-byte loads outstanding (8 with 16 waves: 128-VGPR budget)
    constexpr int U = (WAVES == 16 ? 4 : 8) / NT;
    constexpr int NTHR = WAVES * 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int ks = blockIdx.y;
    const int M = a.M, K = a.K;
    const int nchunks = K >> 6, ntiles = a.N >> 4;
    // K range of this workgroup, then of this wave (contiguous => one linear HBM stream per wave)
    const int cpb = (nchunks + a.ksplit - 1) / a.ksplit;
    const int cb0 = min(ks * cpb, nchunks), cb1 = min(cb0 + cpb, nchunks);
    const int nblk = cb1 - cb0;
    const int per = (nblk + WAVES - 1) / WAVES;
    const int c0 = min(cb0 + wave * per, cb1), c1 = min(c0 + per, cb1);

    // Work item g = NT weight tiles.  With ksplit == 1 the workgroup is persistent over
    // g = blockIdx.x, blockIdx.x + gridDim.x, ...: the x prologue (RMSNorm) runs once per workgroup,
    // and the next item's first weight chunks are put in flight before the current item's reduction.
    int tile[NT];
    const kr_bf16* wp[NT];
    bf16x8 wbuf[U][NT][2];
    auto set_item = [&](int g) {
        if (EPI == DEPI_ROPE_KV) {  // tiles t and t+4: the two rotary halves of 16 head channels
            tile[0] = (g >> 2) * 8 + (g & 3);
            tile[NT - 1] = tile[0] + 4;
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) tile[t] = g * NT + t;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int tt = tile[t] < ntiles ? tile[t] : ntiles - 1;
            wp[t] = a.wp + ((int64_t)tt * nchunks) * 1024 + lane * 8;  // block (tile, k/32) = 512 elements, lane-linear
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + u;
            if (c < c1) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    wbuf[u][t][0] = ld8_nt(wp[t] + (int64_t)c * 1024);
                    wbuf[u][t][1] = ld8_nt(wp[t] + (int64_t)c * 1024 + 512);
                }
            }
        }
    };
    int g = blockIdx.x;
    set_item(g);

    // ---- x slice -> LDS (RMS-normalised, or merged from the attention partials, on the way)
    const bool xlds = a.xmode != 0;
    const int xrow = nblk * 128 + 16;
    float* red = reinterpret_cast<float*>(smem + (xlds ? ((M * xrow + 127) & ~127) : 0));  // [WAVES][NT][64][4]
    if (a.xmode == 2) {
        const int kc = K >> 3;
        for (int b = wave; b < M; b += WAVES) {
            bf16x8 v[8];
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = lane + i * 64;
                if (c < kc) {
                    v[i] = ld8(a.x + (int64_t)b * a.ldx + c * 8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) ss += bf2f(v[i][j]) * bf2f(v[i][j]);
                }
            }
            ss = wave_sum(ss);
            const float rs = rsqrtf(ss / (float)K + a.norm_eps);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = lane + i * 64;
                if (c < kc && c >= cb0 * 8 && c < cb1 * 8) {
                    const bf16x8 nw = ld8(a.norm_w + c * 8);
                    bf16x8 o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(nw[j]) * bfround(bf2f(v[i][j]) * rs));
                    *reinterpret_cast<bf16x8*>(smem + b * xrow + (c - cb0 * 8) * 16) = o;
                }
            }
        }
        __syncthreads();
    } else if (a.xmode == 1) {
        const int cpr = nblk * 8;  // 16-byte chunks per row slice
        for (int i = tid; i < M * cpr; i += NTHR) {
            const int b = i / cpr, c = i - b * cpr;
            *reinterpret_cast<bf16x8*>(smem + b * xrow + c * 16) = ld8(a.x + (int64_t)b * a.ldx + (cb0 * 8 + c) * 8);
        }
        __syncthreads();
    } else if (a.xmode == 3) {
        // x[b][h*128 + d] = sum_p o_p[d] * 2^(m_p - m) / sum_p l_p * 2^(m_p - m): the split-KV merge of
        // the decode attention, done here instead of in a kernel of its own (K = heads * 128).
        const int ns = a.attn_split, nbh = M * (K >> 7);
        float* wts = red;  // [nbh][ns] scratch, consumed before `red` is used for the K reduction
        for (int i = tid; i < nbh * ns; i += NTHR) {
            const int bh = i / ns;
            const float* w = a.attn_ws + (int64_t)bh * ns * 132;
            float mm = -1e30f;
            for (int p = 0; p < ns; ++p) mm = fmaxf(mm, w[p * 132 + 128]);
            float ll = 0.f;
            for (int p = 0; p < ns; ++p) ll += w[p * 132 + 129] * __builtin_amdgcn_exp2f(w[p * 132 + 128] - mm);
            const int p = i - bh * ns;
            wts[i] = ll > 0.f ? __builtin_amdgcn_exp2f(w[p * 132 + 128] - mm) / ll : 0.f;
        }
        __syncthreads();
        const int cpr = K >> 3;
        for (int i = tid; i < M * cpr; i += NTHR) {
            const int b = i / cpr, c = i - b * cpr;  // chunk c of row b: head c>>4, d = (c&15)*8
            if (c < cb0 * 8 || c >= cb1 * 8) continue;
            const int bh = b * (K >> 7) + (c >> 4);
            const float* w = a.attn_ws + (int64_t)bh * ns * 132 + (c & 15) * 8;
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
            for (int p = 0; p < ns; ++p) {
                const float wp_ = wts[bh * ns + p];
                const f32x4 o0 = *reinterpret_cast<const f32x4*>(w + p * 132);
                const f32x4 o1 = *reinterpret_cast<const f32x4*>(w + p * 132 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a0[j] += o0[j] * wp_;
                    a1[j] += o1[j] * wp_;
                }
            }
            bf16x8 ov;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ov[j] = f2bf(a0[j]);
                ov[4 + j] = f2bf(a1[j]);
            }
            *reinterpret_cast<bf16x8*>(smem + b * xrow + (c - cb0 * 8) * 16) = ov;
        }
        __syncthreads();
    }

    // x fragment of k-step (c, h): row fr, k = 64c + 32h + 8fg .. +7
    const char* xl = smem + (fr < M ? fr : 0) * xrow + fg * 16;
    const kr_bf16* xg = a.x + (int64_t)(fr < M ? fr : 0) * a.ldx + fg * 8;
    // wave 0 of the workgroup: cross-wave sum (+ optional cross-workgroup split-K), then the epilogue
    auto finish_item = [&](const int g, const int (&tile)[NT], const float* red) {
    f32x4 sum[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        sum[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const f32x4 p = *reinterpret_cast<const f32x4*>(red + ((w * NT + t) * 64 + lane) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) sum[t][j] += p[j];
        }
    }
    // ---- split-K across workgroups: slab + arrival counter; the last arriver reduces in fixed order
    if (a.ksplit > 1) {
        float* slab = a.ws + ((int64_t)(g * a.ksplit + ks) * NT) * 256;
#pragma unroll
        for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4*>(slab + t * 256 + lane * 4) = sum[t];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int old = 0;
        if (lane == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            old = __hip_atomic_fetch_add(a.counters + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        old = __builtin_amdgcn_readfirstlane(old);
        if (old != a.ksplit - 1) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(a.counters + g, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int t = 0; t < NT; ++t) sum[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < a.ksplit; ++s) {
            const float* sl = a.ws + ((int64_t)(g * a.ksplit + s) * NT) * 256;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x4 p = *reinterpret_cast<const f32x4*>(sl + t * 256 + lane * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) sum[t][j] += p[j];
            }
        }
    }

    // ---- epilogues: lane = (row b = fr, 4 consecutive features 4*fg..4*fg+3 of each tile)
    const int b = fr;
    if (EPI == DEPI_ARGMAX) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = tile[t] * 16 + fg * 4;
            if (tile[t] < ntiles) {
#pragma unroll
                for (int j = 0; j < 4; ++j) better(bv, bi, sum[t][j], n + j);
                if (a.out_f32 && b < M) *reinterpret_cast<f32x4*>(a.out_f32 + (int64_t)b * a.ldc + n) = sum[t];
            }
        }
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            better(bv, bi, ov, oi);
        }
        if (fg == 0 && b < M) {
            a.amax_val[(int64_t)b * a.groups + g] = bv;
            a.amax_idx[(int64_t)b * a.groups + g] = bi;
        }
        return;
    }
    if (EPI == DEPI_SILU8) {
        // one 16-row tile = 8 features: gate rows in lane groups 0,1, their up rows in groups 2,3 (lane ^ 32)
        float u[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) u[j] = __shfl_xor(sum[0][j], 32, 64);
        if (b < M && fg < 2 && tile[0] < ntiles) {
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = f2bf(act_silu(sum[0][j]) * u[j]);
            *reinterpret_cast<bf16x4*>(a.out + (int64_t)b * a.ldc + tile[0] * 8 + fg * 4) = o;
        }
        return;
    }
    if (b >= M) return;
    if (EPI == DEPI_SILU) {
        if (tile[0] >= ntiles) return;
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = f2bf(act_silu(sum[0][j]) * sum[NT - 1][j]);
        *reinterpret_cast<bf16x4*>(a.out + (int64_t)b * a.ldc + g * 16 + fg * 4) = o;
        return;
    }
    if (EPI == DEPI_ROPE_KV) {
        const int hh = tile[0] >> 3;                   // global head index in [q heads | k heads | v heads]
        const int i0 = (tile[0] & 7) * 16 + fg * 4;    // channel in [0, 64)
        float lo[4], hi[4];
        {
            const bf16x4 b0 = *reinterpret_cast<const bf16x4*>(a.bias + tile[0] * 16 + fg * 4);
            const bf16x4 b1 = *reinterpret_cast<const bf16x4*>(a.bias + tile[NT - 1] * 16 + fg * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lo[j] = bfround(sum[0][j] + bf2f(b0[j]));       // the projection output is a bf16 tensor
                hi[j] = bfround(sum[NT - 1][j] + bf2f(b1[j]));
            }
        }
        const int pos = a.ctx_len[b];
        if (hh < a.heads + a.kv_heads) {
            const float* cs = a.cs_table + ((int64_t)b * a.cs_stride + (pos - a.prompt_len[b])) * 128;
            bf16x4 o0, o1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float c = cs[i0 + j], s = cs[64 + i0 + j];
                o0[j] = f2bf(lo[j] * c - hi[j] * s);
                o1[j] = f2bf(hi[j] * c + lo[j] * s);
            }
            kr_bf16* dst = hh < a.heads
                               ? a.q_out + ((int64_t)b * a.heads + hh) * 128
                               : a.kcache + (((int64_t)b * a.kv_heads + (hh - a.heads)) * a.s_max + pos) * 128;
            *reinterpret_cast<bf16x4*>(dst + i0) = o0;
            *reinterpret_cast<bf16x4*>(dst + 64 + i0) = o1;
        } else {
            const int kvh = hh - a.heads - a.kv_heads;
            kr_bf16* vt = a.vtcache + (((int64_t)b * a.kv_heads + kvh) * (a.s_max >> 6) + (pos >> 6)) * (128 * 64) + kr_vt_off(0, pos & 63, 128);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                vt[(i0 + j) * 32] = __builtin_bit_cast(kr_bf16, f2bf(lo[j]));
                vt[(64 + i0 + j) * 32] = __builtin_bit_cast(kr_bf16, f2bf(hi[j]));
            }
        }
        return;
    }
    // PLAIN
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (tile[t] >= ntiles) continue;
        const int n = tile[t] * 16 + fg * 4;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = sum[t][j];
        if (a.bias) {
            const bf16x4 bv = *reinterpret_cast<const bf16x4*>(a.bias + n);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += bf2f(bv[j]);
        }
        if (a.residual) {
            const bf16x4 rv = *reinterpret_cast<const bf16x4*>(a.residual + (int64_t)b * a.ldr + n);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += bf2f(rv[j]);
        }
        if (a.out_f32) {
            *reinterpret_cast<f32x4*>(a.out_f32 + (int64_t)b * a.ldc + n) = (f32x4){v[0], v[1], v[2], v[3]};
        } else {
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = f2bf(v[j]);
            *reinterpret_cast<bf16x4*>(a.out + (int64_t)b * a.ldc + n) = o;
        }
    }
    };
    float* const red0 = red;
    int rbuf = 0;
    for (;;) {
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int cc = c0; cc < c1; cc += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = cc + u;
            if (c < c1) {
                bf16x8 x0, x1;
                if (xlds) {
                    x0 = *reinterpret_cast<const bf16x8*>(xl + (c - cb0) * 128);
                    x1 = *reinterpret_cast<const bf16x8*>(xl + (c - cb0) * 128 + 64);
                } else {
                    x0 = ld8(xg + c * 64);
                    x1 = ld8(xg + c * 64 + 32);
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wbuf[u][t][0], x0, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wbuf[u][t][1], x1, acc[t], 0, 0, 0);
                }
                const int cn = c + U;
                if (cn < c1) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        wbuf[u][t][0] = ld8_nt(wp[t] + (int64_t)cn * 1024);
                        wbuf[u][t][1] = ld8_nt(wp[t] + (int64_t)cn * 1024 + 512);
                    }
                }
            }
        }
    }

    // ---- this item is done streaming: put the next item's first chunks in flight, then reduce
    const int g_cur = g;
    int tile_cur[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) tile_cur[t] = tile[t];
    const int g_next = g + gridDim.x;
    const bool more = a.ksplit == 1 && g_next < a.groups;
    if (more) set_item(g_next);
    // double buffered when persistent: wave 0 may still be reading the other half
    float* red = red0 + rbuf * (WAVES * NT * 256);
    if (a.xmode == 3 && g_cur == (int)blockIdx.x) __syncthreads();  // `red` doubled as the merge-weight scratch
#pragma unroll
    for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4*>(red + ((wave * NT + t) * 64 + lane) * 4) = acc[t];
    __syncthreads();
    if (wave == 0) finish_item(g_cur, tile_cur, red);
    if (!more) break;
    g = g_next;
    rbuf ^= 1;  // (only reached when the grid is smaller than the item count: the host sized `red` for two halves)
    }  // persistent loop
}

// =====================================================================================
// wide layers (gate/up, lm_head): one WAVE per 16-row tile over the full K
// =====================================================================================
// No cross-wave reduction, no barrier after the prologue, and a register ring of U 64-wide K-chunks that keeps
// streaming across tile boundaries (K/64 is a multiple of U, so ring slots stay static): each wave is an
// independent linear HBM stream (U = 12 chunks = 24 KB per wave in flight at K = 1536).  Loads return in issue
// order, hence the order of issue: x rows and the norm weight FIRST (the norm prologue then waits only for them),
// weights right behind; the RMSNorm / LDS staging of x runs while the weights are in flight.  Measured (M=8, gate/up of the 2B decoder,
// chain of launches over rotating copies): launch 1.7 us + body 1.1 + stores 0.6 + x staging 1.8 + weight
// stream 7.5 (= 7.4 TB/s) were purely additive with the weights-first order.
// Wave (block b, wave w) walks tiles b + gridDim.x * (w + W * i).
// MT = 16-row column tiles of the batch: 1 for M <= 16, 2 for M <= 32 (x of 32 rows must fit the LDS: K <= 2048).
// With MT = 2 every weight fragment feeds two MFMAs and the prologue holds 4 x rows per wave, so the ring is shorter.
template <int NCH, int MT> struct WideCfg { static constexpr int U = 8, RL = 8; };   // generic: K <= 4096, K % 512 == 0
template <> struct WideCfg<24, 1> { static constexpr int U = 12, RL = 3; };     // K = 1536 (Qwen2-VL-2B)
template <> struct WideCfg<32, 1> { static constexpr int U = 16, RL = 4; };     // K = 2048 (Qwen2.5-VL-3B)
template <> struct WideCfg<56, 1> { static constexpr int U = 8, RL = 7; };      // K = 3584 (Qwen2-VL-7B): 2 x rows + norm = 84 VGPRs
template <> struct WideCfg<24, 2> { static constexpr int U = 12, RL = 3; };
template <> struct WideCfg<32, 2> { static constexpr int U = 8, RL = 4; };

// ACTIVE = this wave owns at least one tile.  The two cases are separate instantiations selected by ONE
// wave-uniform branch at the top of the kernel (each contains the workgroup's single barrier): a branch around
// the weight loads inside a common body would make the compiler's s_waitcnt bookkeeping assume the shorter
// path, and the wait for x would become a wait for the weights.
// The fields every wave needs before its first load, as plain kernel arguments ahead of the struct: with
// -mllvm -amdgpu-kernarg-preload-count=16 the command processor delivers them in SGPRs at wave start, so the x / weight
// requests do not wait for a scalar load of the (cold) kernarg segment.
struct WideHot {
    const kr_bf16* x; const kr_bf16* wp; const kr_bf16* norm_w; int64_t ldx;
    int M, N, wide_blocks, wide_waves; float norm_eps;
};

template <int EPI, int NCH, bool ACTIVE, bool W8, int MT, bool XF32>
__device__ __forceinline__ void dec_wide_body(const WideHot& h, const DecLinArgs& a, char* smem) {
    using WC = WChunk<W8>;
    constexpr int U = WideCfg<NCH, MT>::U, RL = WideCfg<NCH, MT>::RL;
    constexpr int NR = 2 * MT;  // x rows a wave stages in the branch-free prologue (8 waves x NR rows = 16 MT)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), W = h.wide_waves, nblk = h.wide_blocks;
    const int fr = lane & 15, fg = lane >> 4;
    const int M = h.M, K = NCH ? NCH * 64 : a.K, nchunks = NCH ? NCH : (K >> 6), ntiles = h.N >> 4;
    const int stride = nblk * W;
    const int xrow = K * 2 + 16, kc = K >> 3;
    int t = blockIdx.x + nblk * wave;

    // ---- issue: x rows `wave` and `wave + W` and the norm weight first (branch-free when K is a template
    // constant: s_waitcnt bookkeeping does not survive divergent control flow) ...
    constexpr bool FULL = NCH != 0;  // RL * 64 == K / 8 exactly
    bf16x8 xv[NR][RL], nwv[RL];
    f32x4 xf[XF32 ? NR : 1][XF32 ? RL : 1][2];   // XF32: the rows arrive as f32 (residual accumulator), rounded below
    const bool has_norm = h.norm_w != nullptr;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int b = wave + r * W;
        if constexpr (XF32) {
            const float* xp = reinterpret_cast<const float*>(h.x) + (int64_t)(b < M ? b : 0) * h.ldx;
#pragma unroll
            for (int i = 0; i < RL; ++i) {
                const int c = lane + i * 64;
                if (FULL || c < kc) {
                    xf[r][i][0] = *reinterpret_cast<const f32x4*>(xp + c * 8);
                    xf[r][i][1] = *reinterpret_cast<const f32x4*>(xp + c * 8 + 4);
                }
            }
        } else {
            const kr_bf16* xp = h.x + (int64_t)(b < M ? b : 0) * h.ldx;
#pragma unroll
            for (int i = 0; i < RL; ++i) {
                const int c = lane + i * 64;
                if (FULL || c < kc) xv[r][i] = ld8(xp + c * 8);
            }
        }
    }
    {
        const kr_bf16* np = has_norm ? h.norm_w : h.x;  // always a valid address; unused without a norm
#pragma unroll
        for (int i = 0; i < RL; ++i) {
            const int c = lane + i * 64;
            if (FULL || c < kc) nwv[i] = ld8(np + c * 8);
        }
    }
    // ---- ... then the weight ring
    WC wbuf[U];
    const char* wbase = reinterpret_cast<const char*>(h.wp) + lane * 16;
    const char* wp = wbase + ((int64_t)(ACTIVE ? t : 0) * nchunks) * WC::BYTES;
    if (ACTIVE) {
#pragma unroll
        for (int u = 0; u < U; ++u) wbuf[u].load(wp, u);
    }
    __builtin_amdgcn_sched_barrier(0);  // everything above is issued before any of the norm arithmetic below
    if constexpr (XF32) {
        // the residual stream is a bf16 tensor: round the accumulated f32 rows once, here; workgroup 0 keeps the rounded
        // rows for the kernels that take the residual as bf16 (the next layer's qkv prologue, a PLAIN down_proj)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int b = wave + r * W;
#pragma unroll
            for (int i = 0; i < RL; ++i) {
                if (FULL || lane + i * 64 < kc) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) xv[r][i][j] = f2bf(xf[r][i][j >> 2][j & 3]);
                    if (blockIdx.x == 0 && a.wide_x_out && b < M)
                        *reinterpret_cast<bf16x8*>(a.wide_x_out + (int64_t)b * a.wide_ldxo + (lane + i * 64) * 8) = xv[r][i];
                }
            }
        }
    }
    // ---- x -> LDS, RMS-normalised when norm_w is given
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int b = wave + r * W;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < RL; ++i) {
            if (FULL || lane + i * 64 < kc) {
#pragma unroll
                for (int j = 0; j < 8; ++j) ss += bf2f(xv[r][i][j]) * bf2f(xv[r][i][j]);
            }
        }
        ss = wave_sum(ss);
        const float rs = rsqrtf(ss / (float)K + h.norm_eps);
        if (b < M) {
#pragma unroll
            for (int i = 0; i < RL; ++i) {
                const int c = lane + i * 64;
                if (FULL || c < kc) {
                    bf16x8 o = xv[r][i];
                    if (has_norm) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(nwv[i][j]) * bfround(bf2f(xv[r][i][j]) * rs));
                    }
                    *reinterpret_cast<bf16x8*>(smem + b * xrow + c * 16) = o;
                }
            }
        }
    }
    for (int b = wave + NR * W; b < M; b += W) {  // fewer waves than that: the remaining rows, the slow way
        bf16x8 v[RL];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < RL; ++i) {
            const int c = lane + i * 64;
            if (c < kc) {
                if constexpr (XF32) {
                    const float* xp = reinterpret_cast<const float*>(h.x) + (int64_t)b * h.ldx + c * 8;
                    const f32x4 t0 = *reinterpret_cast<const f32x4*>(xp), t1 = *reinterpret_cast<const f32x4*>(xp + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[i][j] = f2bf(t0[j]);
                        v[i][4 + j] = f2bf(t1[j]);
                    }
                    if (blockIdx.x == 0 && a.wide_x_out)
                        *reinterpret_cast<bf16x8*>(a.wide_x_out + (int64_t)b * a.wide_ldxo + c * 8) = v[i];
                } else {
                    v[i] = ld8(h.x + (int64_t)b * h.ldx + c * 8);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) ss += bf2f(v[i][j]) * bf2f(v[i][j]);
            }
        }
        ss = wave_sum(ss);
        const float rs = h.norm_w ? rsqrtf(ss / (float)K + h.norm_eps) : 1.f;
#pragma unroll
        for (int i = 0; i < RL; ++i) {
            const int c = lane + i * 64;
            if (c < kc) {
                bf16x8 o = v[i];
                if (h.norm_w) {
                    const bf16x8 nw = ld8(h.norm_w + c * 8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(nw[j]) * bfround(bf2f(v[i][j]) * rs));
                }
                *reinterpret_cast<bf16x8*>(smem + b * xrow + c * 16) = o;
            }
        }
    }
    __syncthreads();
    if (!ACTIVE) {
        if (EPI == DEPI_ARGMAX && fg == 0) {  // the sampler reads every wave's partial
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int b = fr + 16 * mt;
                if (b < M) {
                    a.amax_val[(int64_t)b * stride + t] = -INFINITY;
                    a.amax_idx[(int64_t)b * stride + t] = 0x7fffffff;
                }
            }
        }
        return;
    }
    const char* xl[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xl[mt] = smem + min(fr + 16 * mt, M - 1) * xrow;
    const int pidx = t;  // this wave's partial-argmax slot: blockIdx.x + blocks * wave
    float rbv[MT];
    int rbi[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        rbv[mt] = -INFINITY;
        rbi[mt] = 0x7fffffff;
    }

    for (; t < ntiles; t += stride) {
        const int tn = t + stride;
        const bool more = tn < ntiles;
        const char* wpn = wbase + ((int64_t)(more ? tn : t) * nchunks) * WC::BYTES;
        f32x4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int cc = 0; cc < nchunks; cc += U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = cc + u;
                // keep the scheduler from hoisting every chunk's LDS reads to the top (their registers would
                // not fit next to a whole-tile ring)
                if ((u & (MT == 1 ? 3 : 1)) == 0) __builtin_amdgcn_sched_barrier(0);
                const bf16x8 w0 = wbuf[u].frag(0), w1 = wbuf[u].frag(1);   // converted once, used by every column tile
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const bf16x8 x0 = *reinterpret_cast<const bf16x8*>(xl[mt] + c * 128 + WC::x_byte(0, fg));
                    const bf16x8 x1 = *reinterpret_cast<const bf16x8*>(xl[mt] + c * 128 + WC::x_byte(1, fg));
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, x0, acc[mt], 0, 0, 0);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, x1, acc[mt], 0, 0, 0);
                }
                const int cn = c + U;
                if (cn < nchunks) {
                    wbuf[u].load(wp, cn);
                } else if (more) {  // the ring runs on into the next tile
                    wbuf[u].load(wpn, cn - nchunks);
                }
            }
        }
        wp = wpn;
        // ---- epilogue of this wave's tile: lane = (row b, features 4*fg .. 4*fg+3), once per column tile
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int b = fr + 16 * mt;
            apply_w_scale(a.w_scale, t * 16 + fg * 4, acc[mt]);
            if (EPI == DEPI_SILU8) {
                float u4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) u4[j] = __shfl_xor(acc[mt][j], 32, 64);
                if (b < M && fg < 2) {
                    bf16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = f2bf(act_silu(acc[mt][j]) * u4[j]);
                    if (MT == 2 && a.out_xp)
                        *reinterpret_cast<bf16x4*>(reinterpret_cast<char*>(a.out) + kr_xp_byte_offset(b, t * 8 + fg * 4)) = o;
                    else
                        *reinterpret_cast<bf16x4*>(a.out + (int64_t)b * a.ldc + t * 8 + fg * 4) = o;
                }
            } else if (EPI == DEPI_ARGMAX) {  // running argmax over this wave's tiles, written once after the loop
                const int n = t * 16 + fg * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) better(rbv[mt], rbi[mt], acc[mt][j], n + j);
                if (a.out_f32 && b < M) *reinterpret_cast<f32x4*>(a.out_f32 + (int64_t)b * a.ldc + n) = acc[mt];
            } else if (b < M) {  // PLAIN
                const int n = t * 16 + fg * 4;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[mt][j];
                if (a.bias) {
                    const bf16x4 bv = *reinterpret_cast<const bf16x4*>(a.bias + n);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += bf2f(bv[j]);
                }
                if (a.residual) {
                    const bf16x4 rv = *reinterpret_cast<const bf16x4*>(a.residual + (int64_t)b * a.ldr + n);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += bf2f(rv[j]);
                }
                if (a.out_f32) {
                    *reinterpret_cast<f32x4*>(a.out_f32 + (int64_t)b * a.ldc + n) = (f32x4){v[0], v[1], v[2], v[3]};
                } else {
                    bf16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = f2bf(v[j]);
                    *reinterpret_cast<bf16x4*>(a.out + (int64_t)b * a.ldc + n) = o;
                }
            }
        }
    }
    if (EPI == DEPI_ARGMAX) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int b = fr + 16 * mt;
#pragma unroll
            for (int o = 16; o < 64; o <<= 1) {
                const float ov = __shfl_xor(rbv[mt], o, 64);
                const int oi = __shfl_xor(rbi[mt], o, 64);
                better(rbv[mt], rbi[mt], ov, oi);
            }
            if (fg == 0 && b < M) {
                a.amax_val[(int64_t)b * stride + pidx] = rbv[mt];
                a.amax_idx[(int64_t)b * stride + pidx] = rbi[mt];
            }
        }
    }
}

template <int EPI, int NCH, bool W8, int MT, bool XF32>
__global__ void __launch_bounds__(512) dec_wide_kernel(const kr_bf16* x, const kr_bf16* wp, const kr_bf16* norm_w, int64_t ldx, int M,
                                                       int N, int wide_blocks, int wide_waves, float norm_eps, const DecLinArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const WideHot h{x, wp, norm_w, ldx, M, N, wide_blocks, wide_waves, norm_eps};
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if ((int)blockIdx.x + wide_blocks * wave < (N >> 4)) dec_wide_body<EPI, NCH, true, W8, MT, XF32>(h, a, smem);
    else dec_wide_body<EPI, NCH, false, W8, MT, XF32>(h, a, smem);
}

template <int EPI, int NCH, bool W8, int MT, bool XF32>
int launch_wide_x(DecLinArgs& a, int blocks, int waves, kr_stream s) {
    const size_t lds = (size_t)a.M * (a.K * 2 + 16);
    KR_CHECK_ARG(lds <= 160 * 1024, "kr_linear_decode_wide: x (%d rows, K=%d) needs %zu bytes of LDS", a.M, a.K, lds);
    auto fn = &dec_wide_kernel<EPI, NCH, W8, MT, XF32>;
    static KrPerDeviceOnce attr;
    if (attr.need()) {
        KR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    fn<<<blocks, waves * 64, lds, kr_hs(s)>>>(a.x, a.wp, a.norm_w, a.ldx, a.M, a.N, a.wide_blocks, a.wide_waves, a.norm_eps, a);
    KR_CHECK_LAUNCH();
    return KR_OK;
}

// x as f32 (residual accumulator): instantiated for the epilogues that follow a per-head o_proj (SILU8: gate/up) and
// the last layer's final norm (ARGMAX: lm_head); DEPI_PLAIN keeps its bf16 input
template <int EPI, int NCH, bool W8, int MT>
int launch_wide_m(DecLinArgs& a, int blocks, int waves, kr_stream s) {
    if (a.x_is_f32) {
        if constexpr (KR_EXP && (EPI == DEPI_SILU8 || EPI == DEPI_ARGMAX)) {
            return launch_wide_x<EPI, NCH, W8, MT, true>(a, blocks, waves, s);
        } else {
            kr_set_error("kr_linear_decode_wide: f32 x rows (experiment builds: SILU8 and ARGMAX modes only)");
            return KR_ERR_ARG;
        }
    }
    return launch_wide_x<EPI, NCH, W8, MT, false>(a, blocks, waves, s);
}

// =====================================================================================
// wide layers at 17..32 rows when 32 x rows do not fit the LDS (K = 3584: the 7B width): K IN TWO HALVES
// =====================================================================================
// 32 rows x 3584 bf16 = 229 KB against 160 KB of LDS.  The workgroup stages the rows' FIRST half of K (32 x 1792 bf16 =
// 115 KB), every wave streams that half of each of its weight tiles into per-tile accumulators, then (one barrier) the
// second half is staged over the first and the tiles' second halves follow.  Weights are read once; per wave the
// stream is [tile 0, chunks 0..27] [tile 1, chunks 0..27] ... [tile 0, chunks 28..55] ...: 56 KB contiguous pieces.
// The RMS statistic needs whole rows: the prologue loads both halves of the wave's 4 rows (the oldest loads, ahead of
// the weight ring), keeps the first half in registers for staging and re-reads the second half (L2) when its turn comes.
// Ring depth 7 (28 chunks per half = 4 x 7: slots stay static).  TMAX = tiles a wave may own (lm_head of the 7B model
// on 256 x 8 waves: 9504 tiles -> 5).
template <int EPI, bool W8, bool KEEP>
__global__ void __launch_bounds__(512) dec_wide_kh_kernel(const kr_bf16* x, const kr_bf16* wpk, const kr_bf16* norm_w, int64_t ldx, int M,
                                                          int N, int wide_blocks, int wide_waves, float norm_eps, const DecLinArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using WC = WChunk<W8>;
    // ring depth 7 (r4: a 14-deep ring — 28 KB per wave in flight across the restaging barrier — measured 74.9 us against 49.8)
    constexpr int NCH = 56, CH = 28, U = 7, KHALF = CH * 64, HC = KHALF / 8;   // 224 16-byte pieces per half row
    // KEEP (launches with at most two tiles per wave: the 7B decoder's gate/up, 2368 tiles over 256 x 8 waves): the second half
    // of the wave's rows is requested again right after the prologue and waits in registers; up to five tiles per wave
    // (lm_head) the accumulators take those registers and the second half is re-read from L2 when its turn comes
    constexpr int NR = 4, RLH = 4, TMAX = KEEP ? 2 : 5, MT = 2;
    constexpr int XROW = KHALF * 2 + 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), W = wide_waves, nblk = wide_blocks;
    const int fr = lane & 15, fg = lane >> 4;
    const int ntiles = N >> 4, stride = nblk * W;
    const bool has_norm = norm_w != nullptr;
    const int t0 = blockIdx.x + nblk * wave;

    // ---- rows wave, wave + W, ...: first half kept, second half only for the statistic (it is requested again later)
    bf16x8 x0[NR][RLH], x1[NR][RLH], nw0[RLH];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int b = wave + r * W;
        const kr_bf16* xp = x + (int64_t)(b < M ? b : 0) * ldx;
#pragma unroll
        for (int i = 0; i < RLH; ++i) {
            const int c = min(lane + i * 64, HC - 1);
            x0[r][i] = ld8(xp + c * 8);
            x1[r][i] = ld8(xp + KHALF + c * 8);
        }
    }
    {
        const kr_bf16* np = has_norm ? norm_w : x;
#pragma unroll
        for (int i = 0; i < RLH; ++i) nw0[i] = ld8(np + min(lane + i * 64, HC - 1) * 8);
    }
    // ---- weight ring, first half of the first tile
    WC wbuf[U];
    const char* wbase = reinterpret_cast<const char*>(wpk) + lane * 16;
    auto tile_ptr = [&](int i, int kh) {   // half kh of this wave's i-th tile (clamped: an absent tile re-reads the last one)
        const int t = min(t0 + stride * i, ntiles - 1);
        return wbase + ((int64_t)t * NCH + kh * CH) * WC::BYTES;
    };
    const int my_tiles = t0 < ntiles ? (ntiles - 1 - t0) / stride + 1 : 0;   // <= TMAX by the host's geometry check
    if (my_tiles > 0) {
#pragma unroll
        for (int u = 0; u < U; ++u) wbuf[u].load(tile_ptr(0, 0), u);
    }
    __builtin_amdgcn_sched_barrier(0);
    float rs[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < RLH; ++i) {
            if (lane + i * 64 < HC) {
#pragma unroll
                for (int j = 0; j < 8; ++j) ss += bf2f(x0[r][i][j]) * bf2f(x0[r][i][j]) + bf2f(x1[r][i][j]) * bf2f(x1[r][i][j]);
            }
        }
        ss = wave_sum(ss);
        rs[r] = rsqrtf(ss / (float)(2 * KHALF) + norm_eps);
    }
    auto stage = [&](const bf16x8 (&xr)[NR][RLH], const bf16x8 (&nw)[RLH]) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int b = wave + r * W;
            if (b < M) {
#pragma unroll
                for (int i = 0; i < RLH; ++i) {
                    const int c = lane + i * 64;
                    if (c < HC) {
                        bf16x8 o = xr[r][i];
                        if (has_norm) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(nw[i][j]) * bfround(bf2f(xr[r][i][j]) * rs[r]));
                        }
                        *reinterpret_cast<bf16x8*>(smem + b * XROW + c * 16) = o;
                    }
                }
            }
        }
    };
    auto load_second_half = [&]() {
#pragma unroll
        for (int i = 0; i < RLH; ++i) nw0[i] = ld8((has_norm ? norm_w : x) + (has_norm ? KHALF : 0) + min(lane + i * 64, HC - 1) * 8);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int b = wave + r * W;
            const kr_bf16* xp = x + (int64_t)(b < M ? b : 0) * ldx + KHALF;
#pragma unroll
            for (int i = 0; i < RLH; ++i) x1[r][i] = ld8(xp + min(lane + i * 64, HC - 1) * 8);
        }
    };
    stage(x0, nw0);
    __syncthreads();
    if constexpr (KEEP) {   // the second half again, requested NOW (behind the initial ring, ahead of every refill): in registers long
        //                     before the restaging barrier, where a re-read queues behind the whole ring in flight (loads return in
        //                     issue order) — gate/up at the 7B width, 32 rows: 49.9 us with the re-read at the barrier
        __builtin_amdgcn_sched_barrier(0);
        load_second_half();
        __builtin_amdgcn_sched_barrier(0);
    }

    f32x4 acc[TMAX][MT];
#pragma unroll
    for (int i = 0; i < TMAX; ++i)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[i][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const char* xl[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xl[mt] = smem + min(fr + 16 * mt, M - 1) * XROW;

#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
        if (kh == 1) {
            __syncthreads();                 // every wave has read the first half
            // second half of the rows, normalised with the statistic of the whole row: re-read here (L2), or (KEEP) already back
            if constexpr (!KEEP) load_second_half();
            stage(x1, nw0);
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < TMAX; ++i) {
            if (i < my_tiles) {
                const char* wp = tile_ptr(i, kh);
                // what the ring runs on into: this wave's next tile in this half, else its first tile's second half
                const bool next_tile = i + 1 < my_tiles;
                const char* wpn = next_tile ? tile_ptr(i + 1, kh) : tile_ptr(0, 1);
                const bool more = next_tile || kh == 0;
                for (int cc = 0; cc < CH; cc += U) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int c = cc + u;
                        if ((u & 1) == 0) __builtin_amdgcn_sched_barrier(0);
                        const bf16x8 w0 = wbuf[u].frag(0), w1 = wbuf[u].frag(1);
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            const bf16x8 xa = *reinterpret_cast<const bf16x8*>(xl[mt] + c * 128 + WC::x_byte(0, fg));
                            const bf16x8 xb = *reinterpret_cast<const bf16x8*>(xl[mt] + c * 128 + WC::x_byte(1, fg));
                            acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, xa, acc[i][mt], 0, 0, 0);
                            acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, xb, acc[i][mt], 0, 0, 0);
                        }
                        const int cn = c + U;
                        if (cn < CH) {
                            wbuf[u].load(wp, cn);
                        } else if (more) {
                            wbuf[u].load(wpn, cn - CH);
                        }
                    }
                }
            }
        }
    }

    // ---- epilogues, tile by tile (the same arithmetic as dec_wide_body)
    float rbv[MT];
    int rbi[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        rbv[mt] = -INFINITY;
        rbi[mt] = 0x7fffffff;
    }
#pragma unroll
    for (int i = 0; i < TMAX; ++i) {
        if (i >= my_tiles) break;
        const int t = t0 + stride * i;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int b = fr + 16 * mt;
            apply_w_scale(a.w_scale, t * 16 + fg * 4, acc[i][mt]);
            if (EPI == DEPI_SILU8) {
                float u4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) u4[j] = __shfl_xor(acc[i][mt][j], 32, 64);
                if (b < M && fg < 2) {
                    bf16x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = f2bf(act_silu(acc[i][mt][j]) * u4[j]);
                    if (a.out_xp)
                        *reinterpret_cast<bf16x4*>(reinterpret_cast<char*>(a.out) + kr_xp_byte_offset(b, t * 8 + fg * 4)) = o;
                    else
                        *reinterpret_cast<bf16x4*>(a.out + (int64_t)b * a.ldc + t * 8 + fg * 4) = o;
                }
            } else {   // ARGMAX
                const int n = t * 16 + fg * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) better(rbv[mt], rbi[mt], acc[i][mt][j], n + j);
                if (a.out_f32 && b < M) *reinterpret_cast<f32x4*>(a.out_f32 + (int64_t)b * a.ldc + n) = acc[i][mt];
            }
        }
    }
    if (EPI == DEPI_ARGMAX) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int b = fr + 16 * mt;
#pragma unroll
            for (int o = 16; o < 64; o <<= 1) {
                const float ov = __shfl_xor(rbv[mt], o, 64);
                const int oi = __shfl_xor(rbi[mt], o, 64);
                better(rbv[mt], rbi[mt], ov, oi);
            }
            if (fg == 0 && b < M) {   // a wave without tiles leaves (-inf, INT_MAX): the sampler reads every slot
                a.amax_val[(int64_t)b * stride + t0] = rbv[mt];
                a.amax_idx[(int64_t)b * stride + t0] = rbi[mt];
            }
        }
    }
}

template <int EPI, bool W8>
int launch_wide_kh(DecLinArgs& a, int blocks, int waves, kr_stream s) {
    if constexpr (EPI == DEPI_SILU8 || EPI == DEPI_ARGMAX) {
        KR_CHECK_ARG(waves == 8, "kr_linear_decode_wide: 17..32 rows at K = 3584 run 8 waves per workgroup (got %d)", waves);
        const int tiles_per_wave = ((a.N >> 4) + blocks * waves - 1) / (blocks * waves);
        KR_CHECK_ARG(tiles_per_wave <= 5, "kr_linear_decode_wide: %d tiles over %d x %d waves is more than 5 tiles per wave", a.N >> 4,
                     blocks, waves);
        KR_CHECK_ARG(!a.x_is_f32 && !a.bias && !a.residual, "kr_linear_decode_wide: K-halves take bf16 x, no bias / residual");
        const size_t lds = (size_t)32 * (28 * 64 * 2 + 16);
        auto fn = tiles_per_wave <= 2 ? &dec_wide_kh_kernel<EPI, W8, true> : &dec_wide_kh_kernel<EPI, W8, false>;
        static KrPerDeviceOnce attr;
        if (attr.need()) {
            KR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_wide_kh_kernel<EPI, W8, true>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            KR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dec_wide_kh_kernel<EPI, W8, false>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
        fn<<<blocks, 512, lds, kr_hs(s)>>>(a.x, a.wp, a.norm_w, a.ldx, a.M, a.N, a.wide_blocks, a.wide_waves, a.norm_eps, a);
        KR_CHECK_LAUNCH();
        return KR_OK;
    } else {
        kr_set_error("kr_linear_decode_wide: M=%d > 16 at K = 3584 is built for the SILU8 and ARGMAX modes", a.M);
        return KR_ERR_ARG;
    }
}

template <int EPI, int NCH, bool W8>
int launch_wide_w(DecLinArgs& a, int blocks, int waves, kr_stream s) {
    if (a.M > 16) {  // two column tiles: only instantiated where 32 rows of x fit the LDS
        if constexpr (NCH == 56) {
            return launch_wide_kh<EPI, W8>(a, blocks, waves, s);      // 32 x rows do not fit the LDS: K in two halves
        } else {   // (the generic K instantiation checks the LDS size itself: 32 rows fit up to K = 2048)
            return launch_wide_m<EPI, NCH, W8, 2>(a, blocks, waves, s);
        }
    }
    return launch_wide_m<EPI, NCH, W8, 1>(a, blocks, waves, s);
}

template <int EPI, int NCH>
int launch_wide_n(DecLinArgs& a, int blocks, int waves, kr_stream s) {
    return a.w_scale ? launch_wide_w<EPI, NCH, true>(a, blocks, waves, s) : launch_wide_w<EPI, NCH, false>(a, blocks, waves, s);
}

template <int EPI>
int launch_wide(DecLinArgs& a, int blocks, int waves, kr_stream s) {
    switch (a.K >> 6) {
        case 24: return launch_wide_n<EPI, 24>(a, blocks, waves, s);
        case 32: return launch_wide_n<EPI, 32>(a, blocks, waves, s);
        case 56: return launch_wide_n<EPI, 56>(a, blocks, waves, s);
        default: return launch_wide_n<EPI, 0>(a, blocks, waves, s);
    }
}

// =====================================================================================
// narrow layers (qkv, o_proj, down_proj): one workgroup per tile (pair), K split over its waves
// =====================================================================================
// Same issue discipline as the wide kernel (x and the norm weight before the weights, branch-free, geometry
// from kernel arguments), plus a split of K over blockIdx.y whose reduction is DEFERRED to the consumer: a
// cross-workgroup reduction inside the launch costs a release/acquire pair (5-6 us measured, more than the
// launch it saves), while two f32 slabs added in the next kernel's x prologue cost a few hundred bytes per lane.
//   down_proj   : EPI PARTIAL, ksplit 2 -> 192 workgroups instead of 96, slabs [2][M][d] f32
//   next qkv    : NORM prologue with PKS = 2: x_new = bf16(x + slab0 + slab1) (workgroup 0 stores it to x_out,
//                 the other residual buffer), RMSNorm of x_new, QKV + bias + M-RoPE + KV append as before
template <int NCH> struct NarrowCfg { static constexpr int RL = 8; };
template <> struct NarrowCfg<24> { static constexpr int RL = 3; };
template <> struct NarrowCfg<32> { static constexpr int RL = 4; };
template <> struct NarrowCfg<56> { static constexpr int RL = 7; };
constexpr int DEPI_PARTIAL = 16;  // internal: PLAIN with deferred split-K slabs

// U = ring depth in 64-wide K chunks: the host picks the smallest instantiated U that covers a wave's share of K
// (then every chunk is requested up front and at most one request per wave is redundant), else the deepest ring.
template <int NT, int EPI, int WAVES, int NCH, int PKS, bool NORM, int U, bool W8, int MT>
__global__ void __launch_bounds__(WAVES * 64) dec_narrow_kernel(const kr_bf16* hx, const kr_bf16* hwp, const kr_bf16* hnorm_w, const float* hpart_in,
                                                                int64_t hldx, int hM, int hN, int hK, int hcpb, float hnorm_eps,
                                                                int hprows, const DecLinArgs a) {
    // hprows: rows per slab of part_in ([PKS][hprows][K]); = M unless the launch covers a row range of a larger batch
    // hot fields as leading scalars: preloaded into SGPRs at wave start (see WideHot)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using WC = WChunk<W8>;
    constexpr int RL = NarrowCfg<NCH>::RL;
    constexpr bool FULL = NCH != 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int g = blockIdx.x, ks = blockIdx.y;
    if (KR_EXP && a.pf_blocks > 0 && (int)blockIdx.x >= (int)gridDim.x - a.pf_blocks) {
        // prefetcher workgroup: 16 loads of 16 bytes in flight per lane, results discarded (kept alive by the asm)
        const int pb = blockIdx.x - ((int)gridDim.x - a.pf_blocks);
        const int64_t n16 = a.pf_bytes >> 4, stride = (int64_t)a.pf_blocks * (WAVES * 64);
        const u32x4* p = reinterpret_cast<const u32x4*>(a.pf_ptr);
        u32x4 acc = {0u, 0u, 0u, 0u};
        int64_t i = (int64_t)pb * (WAVES * 64) + threadIdx.x;
        for (; i + 15 * stride < n16; i += 16 * stride) {
            u32x4 v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = p[i + k * stride];
#pragma unroll
            for (int k = 0; k < 16; ++k) acc ^= v[k];
        }
        for (; i < n16; i += stride) acc ^= p[i];
        asm volatile("" ::"v"(acc[0] ^ acc[1] ^ acc[2] ^ acc[3]));
        return;
    }
    const int M = hM, K = hK;
    const int nchunks = NCH ? NCH : (K >> 6), ntiles = hN >> 4, kc = K >> 3;
    const int cpb = hcpb;   // chunks per K split, computed on the host (a runtime integer division is ~30 instructions here)
    const int cb0 = min(ks * cpb, nchunks), cb1 = min(cb0 + cpb, nchunks);
    const int nblk = cb1 - cb0;
    const int c0 = cb0 + (wave * nblk) / WAVES, c1 = cb0 + ((wave + 1) * nblk) / WAVES;  // even shares, contiguous
    const int xrow = nblk * 128 + 16;
    // [WAVES][NT][64][4]; with two batch column tiles (MT = 2) it is used twice, one tile after the other
    float* red = reinterpret_cast<float*>(smem + (NORM ? ((M * xrow + 127) & ~127) : 0));

    int tile[NT];
    if (EPI == DEPI_ROPE_KV) {
        tile[0] = (g >> 2) * 8 + (g & 3);
        tile[NT - 1] = tile[0] + 4;
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) tile[t] = g * NT + t;
    }
    const char* wp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
        wp[t] = reinterpret_cast<const char*>(hwp) + ((int64_t)min(tile[t], ntiles - 1) * nchunks) * WC::BYTES + lane * 16;
    int rb[MT];  // batch rows of this lane's accumulator columns (one per column tile)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) rb[mt] = min(fr + 16 * mt, M - 1);

    // ---- 1. the oldest loads of every wave: what the prologue waits for (addresses from preloaded arguments only)
    bf16x8 xv[NORM ? RL : 1], nwv[NORM ? RL : 1];
    f32x4 pv[PKS ? PKS : 1][NORM ? RL : 1][2];
    bf16x8 xf[NORM ? 1 : U][MT][2];
    const char* xg[MT];   // x fragments: + c*128 + WC::x_byte(h, fg)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xg[mt] = reinterpret_cast<const char*>(hx + (int64_t)rb[mt] * hldx);
    if (NORM) {
        const int b = wave < M ? wave : 0;
#pragma unroll
        for (int i = 0; i < RL; ++i) {
            const int c = lane + i * 64;
            if (FULL || c < kc) {
                xv[i] = ld8(hx + (int64_t)b * hldx + c * 8);
#pragma unroll
                for (int k = 0; k < PKS; ++k) {
                    const float* pp = hpart_in + ((int64_t)k * hprows + b) * K + c * 8;
                    pv[k][i][0] = *reinterpret_cast<const f32x4*>(pp);
                    pv[k][i][1] = *reinterpret_cast<const f32x4*>(pp + 4);
                }
                nwv[i] = ld8(hnorm_w + c * 8);
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = min(max(min(c0 + u, c1 - 1), cb0), nchunks - 1);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                xf[u][mt][0] = *reinterpret_cast<const bf16x8*>(xg[mt] + c * 128 + WC::x_byte(0, fg));
                xf[u][mt][1] = *reinterpret_cast<const bf16x8*>(xg[mt] + c * 128 + WC::x_byte(1, fg));
            }
        }
    }
    // decode position / prompt length of the rotary epilogue: their pointers come from the argument STRUCT (a scalar
    // load), so they are requested after the x rows, which need preloaded arguments only
    int pos[MT], plen[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        pos[mt] = plen[mt] = 0;
        if (EPI == DEPI_ROPE_KV) {
            pos[mt] = a.ctx_len[rb[mt]];
            plen[mt] = a.prompt_len[rb[mt]];
        }
    }
    // ---- 2. the weight ring (clamped, not branched: short waves re-request their last chunk)
    WC wbuf[U][NT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int c = min(max(min(c0 + u, c1 - 1), cb0), nchunks - 1);
#pragma unroll
        for (int t = 0; t < NT; ++t) wbuf[u][t].load(wp[t], c);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (a.zero_ptr) {   // zero the other split-K accumulator (behind this launch's own requests: plain 16-byte stores)
        // over the WORKING workgroups only: prefetch workgroups (experiment builds) have returned above
        const int zgx = (int)gridDim.x - (KR_EXP ? a.pf_blocks : 0);
        const int zstep = zgx * (int)gridDim.y * (WAVES * 64);
        for (int zi = ((int)blockIdx.y * zgx + (int)blockIdx.x) * (WAVES * 64) + tid; zi < a.zero_n16; zi += zstep)
            reinterpret_cast<f32x4*>(a.zero_ptr)[zi] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    // ---- 3. x (+ deferred partial sums) -> RMSNorm -> LDS
    if constexpr (NORM) {
        auto norm_row = [&](const int b, bf16x8 (&xr)[RL], const bf16x8 (&nw)[RL]) {
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < RL; ++i) {
                if (FULL || lane + i * 64 < kc) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) ss += bf2f(xr[i][j]) * bf2f(xr[i][j]);
                }
            }
            ss = wave_sum(ss);
            const float rs = rsqrtf(ss / (float)K + hnorm_eps);
#pragma unroll
            for (int i = 0; i < RL; ++i) {
                const int c = lane + i * 64;
                if ((FULL || c < kc) && c >= cb0 * 8 && c < cb1 * 8) {
                    bf16x8 o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(nw[i][j]) * bfround(bf2f(xr[i][j]) * rs));
                    *reinterpret_cast<bf16x8*>(smem + b * xrow + (c - cb0 * 8) * 16) = o;
                }
            }
        };
        if (PKS) {
#pragma unroll
            for (int i = 0; i < RL; ++i) {
                if (FULL || lane + i * 64 < kc) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        float v = bf2f(xv[i][j]);
#pragma unroll
                        for (int k = 0; k < PKS; ++k) v += pv[k][i][j >> 2][j & 3];
                        xv[i][j] = f2bf(v);
                    }
                    if (g == 0 && ks == 0 && wave < M)
                        *reinterpret_cast<bf16x8*>(a.x_out + (int64_t)wave * a.ldxo + (lane + i * 64) * 8) = xv[i];
                }
            }
        }
        if (KR_EXP && a.x_out_f32 && g == 0 && ks == 0 && wave < M) {   // fast-residual mode: x_new also as the f32 accumulator's start value
#pragma unroll
            for (int i = 0; i < RL; ++i) {
                if (FULL || lane + i * 64 < kc) {
                    float* xo = a.x_out_f32 + (int64_t)wave * a.ldxf + (lane + i * 64) * 8;
                    *reinterpret_cast<f32x4*>(xo) = (f32x4){bf2f(xv[i][0]), bf2f(xv[i][1]), bf2f(xv[i][2]), bf2f(xv[i][3])};
                    *reinterpret_cast<f32x4*>(xo + 4) = (f32x4){bf2f(xv[i][4]), bf2f(xv[i][5]), bf2f(xv[i][6]), bf2f(xv[i][7])};
                }
            }
        }
        if (wave < M) norm_row(wave, xv, nwv);
        for (int b = wave + WAVES; b < M; b += WAVES) {  // more rows than waves: the slow way (loads behind the weights)
            bf16x8 v[RL];
#pragma unroll
            for (int i = 0; i < RL; ++i) {
                const int c = lane + i * 64;
                if (FULL || c < kc) {
                    v[i] = ld8(hx + (int64_t)b * hldx + c * 8);
                    if (PKS) {
                        float f[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) f[j] = bf2f(v[i][j]);
#pragma unroll
                        for (int k = 0; k < PKS; ++k) {
                            const float* pp = hpart_in + ((int64_t)k * hprows + b) * K + c * 8;
                            const f32x4 p0 = *reinterpret_cast<const f32x4*>(pp), p1 = *reinterpret_cast<const f32x4*>(pp + 4);
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                f[j] += p0[j];
                                f[4 + j] += p1[j];
                            }
                        }
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[i][j] = f2bf(f[j]);
                        if (g == 0 && ks == 0) *reinterpret_cast<bf16x8*>(a.x_out + (int64_t)b * a.ldxo + c * 8) = v[i];
                    }
                    if (KR_EXP && a.x_out_f32 && g == 0 && ks == 0) {
                        float* xo = a.x_out_f32 + (int64_t)b * a.ldxf + c * 8;
                        *reinterpret_cast<f32x4*>(xo) = (f32x4){bf2f(v[i][0]), bf2f(v[i][1]), bf2f(v[i][2]), bf2f(v[i][3])};
                        *reinterpret_cast<f32x4*>(xo + 4) = (f32x4){bf2f(v[i][4]), bf2f(v[i][5]), bf2f(v[i][6]), bf2f(v[i][7])};
                    }
                }
            }
            norm_row(b, v, nwv);
        }
        __syncthreads();
    }
    // ---- 4. epilogue operands of the rotary modes: in flight during the K loop
    float csv[MT][8];
    bf16x4 bias0 = {}, bias1 = {};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 8; ++j) csv[mt][j] = 0.f;
    if (EPI == DEPI_ROPE_KV) {
        const int i0 = (tile[0] & 7) * 16 + fg * 4;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float* cs = a.cs_table + ((int64_t)rb[mt] * a.cs_stride + (pos[mt] - plen[mt])) * 128;
            const f32x4 cv = *reinterpret_cast<const f32x4*>(cs + i0), sv = *reinterpret_cast<const f32x4*>(cs + 64 + i0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                csv[mt][j] = cv[j];
                csv[mt][4 + j] = sv[j];
            }
        }
        bias0 = *reinterpret_cast<const bf16x4*>(a.bias + tile[0] * 16 + fg * 4);
        bias1 = *reinterpret_cast<const bf16x4*>(a.bias + tile[NT - 1] * 16 + fg * 4);
    }

    // bias / residual of the PLAIN epilogue: requested here, behind the weight ring, instead of after the cross-wave
    // reduction (where the residual was one more dependent memory round trip at the end of every o_proj launch).
    // Unconditional loads (a valid dummy address when the operand is absent): no branch, no early wait.
    bf16x4 res_pre[MT][NT], bias_pre[NT];
    if (EPI == DEPI_PLAIN) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = min(tile[t], ntiles - 1) * 16 + fg * 4;
            bias_pre[t] = *reinterpret_cast<const bf16x4*>(a.bias ? a.bias + n : hx);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                res_pre[mt][t] = *reinterpret_cast<const bf16x4*>(a.residual ? a.residual + (int64_t)rb[mt] * a.ldr + n : hx);
        }
    }

    // ---- 5. K loop of this wave
    const char* xl[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xl[mt] = smem + rb[mt] * xrow;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int cc = c0; cc < c1; cc += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = cc + u;
            if (c < c1) {
                bf16x8 x0[MT], x1[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if (NORM) {
                        x0[mt] = *reinterpret_cast<const bf16x8*>(xl[mt] + (c - cb0) * 128 + WC::x_byte(0, fg));
                        x1[mt] = *reinterpret_cast<const bf16x8*>(xl[mt] + (c - cb0) * 128 + WC::x_byte(1, fg));
                    } else {
                        x0[mt] = xf[u][mt][0];
                        x1[mt] = xf[u][mt][1];
                    }
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const bf16x8 w0 = wbuf[u][t].frag(0), w1 = wbuf[u][t].frag(1);   // converted once per column tile pair
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, x0[mt], acc[mt][t], 0, 0, 0);
                        acc[mt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, x1[mt], acc[mt][t], 0, 0, 0);
                    }
                }
                const int cn = c + U;
                if (cn < c1) {
                    if (!NORM) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            xf[u][mt][0] = *reinterpret_cast<const bf16x8*>(xg[mt] + cn * 128 + WC::x_byte(0, fg));
                            xf[u][mt][1] = *reinterpret_cast<const bf16x8*>(xg[mt] + cn * 128 + WC::x_byte(1, fg));
                        }
                    }
#pragma unroll
                    for (int t = 0; t < NT; ++t) wbuf[u][t].load(wp[t], cn);
                }
            }
        }
    }
    // ---- 6. cross-wave sum, epilogue by wave 0 (one batch column tile after the other through the same buffer)
    auto epilogue = [&](const int mt, f32x4 (&sum)[NT]) {
        const int b = fr + 16 * mt;
        if (b >= M) return;
#pragma unroll
        for (int t = 0; t < NT; ++t) apply_w_scale(a.w_scale, min(tile[t], ntiles - 1) * 16 + fg * 4, sum[t]);
        if (EPI == DEPI_ROPE_KV) {
            const int hh = tile[0] >> 3;                   // global head index in [q heads | k heads | v heads]
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
                    o0[j] = f2bf(lo[j] * csv[mt][j] - hi[j] * csv[mt][4 + j]);
                    o1[j] = f2bf(hi[j] * csv[mt][j] + lo[j] * csv[mt][4 + j]);
                }
                kr_bf16* dst = hh < a.heads
                                   ? a.q_out + ((int64_t)b * a.heads + hh) * 128
                                   : a.kcache + (((int64_t)b * a.kv_heads + (hh - a.heads)) * a.s_max + pos[mt]) * 128;
                *reinterpret_cast<bf16x4*>(dst + i0) = o0;
                *reinterpret_cast<bf16x4*>(dst + 64 + i0) = o1;
            } else {
                const int kvh = hh - a.heads - a.kv_heads;
                kr_bf16* vt = a.vtcache + (((int64_t)b * a.kv_heads + kvh) * (a.s_max >> 6) + (pos[mt] >> 6)) * (128 * 64) + kr_vt_off(0, pos[mt] & 63, 128);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    vt[(i0 + j) * 32] = __builtin_bit_cast(kr_bf16, f2bf(lo[j]));
                    vt[(64 + i0 + j) * 32] = __builtin_bit_cast(kr_bf16, f2bf(hi[j]));
                }
            }
            return;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (tile[t] >= ntiles) continue;
            const int n = tile[t] * 16 + fg * 4;
            if (EPI == DEPI_PARTIAL) {
                if (a.part_atomic) {
                    float* acc = a.out_f32 + (int64_t)b * a.ldc + n;
#pragma unroll
                    for (int j = 0; j < 4; ++j) atomicAdd(acc + j, sum[t][j]);
                } else {
                    *reinterpret_cast<f32x4*>(a.out_f32 + ((int64_t)ks * M + b) * a.ldc + n) = sum[t];
                }
                continue;
            }
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = sum[t][j];
            if (a.bias) {
                const bf16x4 bv = EPI == DEPI_PLAIN ? bias_pre[t] : *reinterpret_cast<const bf16x4*>(a.bias + n);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += bf2f(bv[j]);
            }
            if (a.residual) {
                const bf16x4 rv = EPI == DEPI_PLAIN ? res_pre[mt][t]
                                                    : *reinterpret_cast<const bf16x4*>(a.residual + (int64_t)b * a.ldr + n);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += bf2f(rv[j]);
            }
            if (a.out_f32) {
                *reinterpret_cast<f32x4*>(a.out_f32 + (int64_t)b * a.ldc + n) = (f32x4){v[0], v[1], v[2], v[3]};
            } else {
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = f2bf(v[j]);
                *reinterpret_cast<bf16x4*>(a.out + (int64_t)b * a.ldc + n) = o;
            }
        }
    };
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (mt > 0) __syncthreads();   // wave 0 has read the previous column tile's partial sums
#pragma unroll
        for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4*>(red + ((wave * NT + t) * 64 + lane) * 4) = acc[mt][t];
        __syncthreads();
        if (wave == 0) {
            f32x4 sum[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                // THE fold of this K range, for every batch size (kr_decode32.hip reproduces it): the waves' sums in wave order
                // inside each HALF of the waves, then the two halves added — so that a 17..32-row launch may give the two halves
                // to two workgroups (each folds its own, both add into the slab: two addends, order-free) and still produce these
                // bits (r4; rounds 1-3: one left fold over all waves)
                f32x4 half[2];
#pragma unroll
                for (int hfi = 0; hfi < 2; ++hfi) {
                    half[hfi] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int w = hfi * (WAVES / 2); w < (hfi + 1) * (WAVES / 2); ++w) {
                        const f32x4 p = *reinterpret_cast<const f32x4*>(red + ((w * NT + t) * 64 + lane) * 4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) half[hfi][j] += p[j];
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) sum[t][j] = half[0][j] + half[1][j];
            }
            epilogue(mt, sum);
        }
    }
}

template <int NT, int EPI, int WAVES, int NCH, int PKS, bool NORM, int U, bool W8, int MT>
int launch_narrow_m(DecLinArgs& a, int groups, kr_stream s) {
    const int nchunks = a.K >> 6, cpb = (nchunks + a.ksplit - 1) / a.ksplit;
    const size_t xbytes = NORM ? (((size_t)a.M * (cpb * 128 + 16) + 127) & ~(size_t)127) : 0;
    const size_t lds = xbytes + (size_t)WAVES * NT * 256 * 4;
    KR_CHECK_ARG(lds <= 160 * 1024, "kr_linear_decode_narrow: M=%d K=%d needs %zu bytes of LDS", a.M, a.K, lds);
    auto fn = &dec_narrow_kernel<NT, EPI, WAVES, NCH, PKS, NORM, U, W8, MT>;
    static KrPerDeviceOnce attr;
    if (attr.need()) {
        KR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    KR_CHECK_ARG(a.pf_blocks == 0 || a.ksplit == 1, "kr_linear_decode_narrow: prefetch workgroups need ksplit 1");
    fn<<<dim3(groups + a.pf_blocks, a.ksplit), WAVES * 64, lds, kr_hs(s)>>>(a.x, a.wp, a.norm_w, a.part_in, a.ldx, a.M, a.N, a.K, cpb,
                                                                            a.norm_eps, a.part_rows > 0 ? a.part_rows : a.M, a);
    KR_CHECK_LAUNCH();
    return KR_OK;
}

// M > 16: two batch column tiles.  Only on 8-wave workgroups (the x fragments of the direct path double) and not
// with the K = 3584 row registers; the engine sizes its batches accordingly.
template <int NT, int EPI, int WAVES, int NCH, int PKS, bool NORM, int U, bool W8>
int launch_narrow_w(DecLinArgs& a, int groups, kr_stream s) {
    if (a.M > 16) {
        if constexpr (WAVES == 8 && NCH != 56) {
            return launch_narrow_m<NT, EPI, WAVES, NCH, PKS, NORM, U, W8, 2>(a, groups, s);
        } else {
            kr_set_error("kr_linear_decode_narrow: M=%d > 16 needs 8-wave workgroups and K != 3584", a.M);
            return KR_ERR_ARG;
        }
    }
    return launch_narrow_m<NT, EPI, WAVES, NCH, PKS, NORM, U, W8, 1>(a, groups, s);
}

template <int NT, int EPI, int WAVES, int NCH, int PKS, bool NORM, int U>
int launch_narrow_u(DecLinArgs& a, int groups, kr_stream s) {
    return a.w_scale ? launch_narrow_w<NT, EPI, WAVES, NCH, PKS, NORM, U, true>(a, groups, s)
                     : launch_narrow_w<NT, EPI, WAVES, NCH, PKS, NORM, U, false>(a, groups, s);
}

// chunks of K the busiest wave of a workgroup owns
inline int narrow_share(const DecLinArgs& a, int waves) {
    const int nchunks = a.K >> 6, cpb = (nchunks + a.ksplit - 1) / a.ksplit;
    return (cpb + waves - 1) / waves;
}

// x fragments straight from global memory (o_proj, down_proj): rings of 3 / 5 / 8 chunks (x + weights = 16 U VGPRs)
template <int EPI, int WAVES>
int launch_narrow_direct(DecLinArgs& a, int groups, kr_stream s) {
    const int share = narrow_share(a, WAVES);
    if constexpr (EPI == DEPI_PARTIAL) {
        // TWO weight tiles per workgroup share one ring of x fragments where there are tiles enough to keep every CU
        // loading (>= 192: down_proj at 7B widths, 224 tiles -> 112 x 2 workgroups).  A direct-path wave fetches its x
        // fragment (1-2 KB per K chunk, from L2) with every weight chunk (2 KB): with one tile a third of its loads
        // move activations, not weights.  Same-process A/B on the 7B decode step: 3.122 -> 2.986 ms (16 waves), 2.950
        // (8 waves, 5-deep ring) = 62.9 % of the step roofline; at 2B widths (96 tiles -> 48 x 2 workgroups) it starves
        // the chip: 1.171 -> 1.240 ms.  KARANTA_NARROW_NT2 = 0 / 1 forces either form.  (o_proj — no K split, so half as
        // many workgroups — loses with two tiles at both widths: 7B 2.98 -> 3.03 ms, 2B 1.17 -> 1.23.)
        const char* e = getenv("KARANTA_NARROW_NT2");
        const bool nt2 = e ? atoi(e) != 0 : groups >= 192;
        if (nt2 && (groups & 1) == 0 && (a.M <= 16 || WAVES == 8)) {   // two batch column tiles: 8-wave workgroups only
            if constexpr (WAVES == 16) return launch_narrow_u<2, EPI, WAVES, 0, 0, false, 3>(a, groups / 2, s);
            else return launch_narrow_u<2, EPI, WAVES, 0, 0, false, 5>(a, groups / 2, s);
        }
    }
    if (share <= 3) return launch_narrow_u<1, EPI, WAVES, 0, 0, false, 3>(a, groups, s);
    if constexpr (WAVES == 16) {  // 128-VGPR budget
        return launch_narrow_u<1, EPI, WAVES, 0, 0, false, 5>(a, groups, s);
    } else {
        if (share <= 5) return launch_narrow_u<1, EPI, WAVES, 0, 0, false, 5>(a, groups, s);
        return launch_narrow_u<1, EPI, WAVES, 0, 0, false, 8>(a, groups, s);
    }
}

// NORM kernels are specialised on K (x row registers); PKS = 2 only where the registers allow it
template <int NT, int EPI, int NCH, int PKS>
int launch_narrow_norm_u(DecLinArgs& a, int groups, kr_stream s) {
    if (narrow_share(a, 8) <= 3) return launch_narrow_u<NT, EPI, 8, NCH, PKS, true, 3>(a, groups, s);
    return launch_narrow_u<NT, EPI, 8, NCH, PKS, true, 8 / NT>(a, groups, s);
}

template <int NT, int EPI>
int launch_narrow_norm(DecLinArgs& a, int groups, kr_stream s) {
    const int nch = a.K >> 6;
    if (a.part_in && a.n_part == 1) {
        if (nch == 24) return launch_narrow_norm_u<NT, EPI, 24, 1>(a, groups, s);
        if (nch == 32) return launch_narrow_norm_u<NT, EPI, 32, 1>(a, groups, s);
        if (nch == 56) return launch_narrow_norm_u<NT, EPI, 56, 1>(a, groups, s);
        kr_set_error("kr_linear_decode_narrow: deferred partial sums need K = 1536, 2048 or 3584 (K=%d)", a.K);
        return KR_ERR_ARG;
    }
    if (a.part_in) {
        if (nch == 24) return launch_narrow_norm_u<NT, EPI, 24, 2>(a, groups, s);
        if (nch == 32) return launch_narrow_norm_u<NT, EPI, 32, 2>(a, groups, s);
        if (nch == 56) return launch_narrow_norm_u<NT, EPI, 56, 2>(a, groups, s);
        kr_set_error("kr_linear_decode_narrow: deferred partial sums need K = 1536, 2048 or 3584 (K=%d)", a.K);
        return KR_ERR_ARG;
    }
    if (nch == 24) return launch_narrow_norm_u<NT, EPI, 24, 0>(a, groups, s);
    if (nch == 32) return launch_narrow_norm_u<NT, EPI, 32, 0>(a, groups, s);
    if (nch == 56) return launch_narrow_norm_u<NT, EPI, 56, 0>(a, groups, s);
    return launch_narrow_norm_u<NT, EPI, 0, 0>(a, groups, s);
}

#ifdef KR_EXPERIMENTS
// =====================================================================================
// o_proj split over the attention heads, merge of the split-KV partials in its prologue, float-atomic epilogue
// =====================================================================================
// The launch that replaces [attn_merge_kernel -> o_proj] in the fast-residual mode.  The merge of the attention's
// split-KV partials is an all-to-all only because o_proj's K dimension spans all heads.  Split K BY HEAD instead:
// workgroup (tg, h) merges head h alone (M x n_split records of 528 bytes: 34 KB at M = 8 — not the 400 KB every
// workgroup of a merging o_proj prologue had to re-read), multiplies the merged [M, 128] slice by W_o[rows of tile
// group tg, columns of head h] (2 K-chunks per 16-row tile, straight to VGPRs, no cross-wave reduction) and ADDS its
// [M, rows] product to the f32 residual accumulator with global_atomic_add_f32 — `heads` adders per element, every
// wave-instruction 256 contiguous bytes of one row (the shape the microarch guide measures at the full atomic rate;
// 590 KB of adds per launch for the 2B model against 1.3 TB/s).  The accumulator was set to x_new (f32) by the qkv
// launch; the gate/up launch reads it back and rounds it to bf16 once.  Float atomics add in arrival order, so the
// low bits of the f32 sums vary from run to run: this is the engine's FAST mode; the deterministic slab path stays the
// parity mode.
template <bool W8, int TI, int NS>
__global__ void __launch_bounds__(512) dec_oproj_heads_kernel(const float* __restrict__ ws, const char* __restrict__ wp,
                                                              const float* __restrict__ w_scale, float* __restrict__ x_acc,
                                                              int64_t ld_acc, int M, int heads, int n_split, int tw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using WC = WChunk<W8>;
    constexpr int HD = 128, REC = HD + 4, XROW = HD * 2 + 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int tg = blockIdx.x, h = blockIdx.y;
    const int nchunks = heads * 2;                        // K = heads * 128
    float* outs = reinterpret_cast<float*>(smem + ((32 * XROW + 127) & ~127));   // [M][tw * 16] f32

    // ---- 1. the partial records of head h (oldest loads: the merge waits for them)
    const int ns = NS ? NS : n_split;
    constexpr int MAXS = NS ? NS : 16;
    const int b0 = tid >> 5, d4 = (tid & 31) << 2;        // thread -> (sequence b0 [+16], channels d4 .. d4+3)
    f32x4 ro[2][MAXS], rm[2][MAXS];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int b = b0 + 16 * mt;
        if (mt == 0 || M > 16) {
            const float* rec = ws + ((int64_t)(b < M ? b : 0) * heads + h) * ns * REC;
#pragma unroll
            for (int p = 0; p < MAXS; ++p) {
                if (p < ns) {
                    ro[mt][p] = *reinterpret_cast<const f32x4*>(rec + p * REC + d4);
                    rm[mt][p] = *reinterpret_cast<const f32x4*>(rec + p * REC + HD);
                }
            }
        }
    }
    // ---- 2. the weight fragments of this wave's tiles: tile tg * tw + wave + 8 i, chunks 2h and 2h + 1
    WC wb[TI][2];
    int tile[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int tl = wave + 8 * i;
        tile[i] = tg * tw + (tl < tw ? tl : tw - 1);
        const char* p = wp + ((int64_t)tile[i] * nchunks) * WC::BYTES + lane * 16;
        wb[i][0].load(p, 2 * h);
        wb[i][1].load(p, 2 * h + 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- 3. merge -> bf16 -> LDS x[b][128]
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int b = b0 + 16 * mt;
        if ((mt == 0 || M > 16) && b < M) {
            float mm = -1e30f;
#pragma unroll
            for (int p = 0; p < MAXS; ++p)
                if (p < ns) mm = fmaxf(mm, rm[mt][p][0]);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            float ll = 0.f;
#pragma unroll
            for (int p = 0; p < MAXS; ++p) {
                if (p < ns) {
                    const float sc = __builtin_amdgcn_exp2f(rm[mt][p][0] - mm);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] += ro[mt][p][j] * sc;
                    ll += rm[mt][p][1] * sc;
                }
            }
            const float inv = ll > 0.f ? 1.0f / ll : 0.f;   // same arithmetic as attn_merge_kernel / the in-launch merge
            bf16x4 ov;
#pragma unroll
            for (int j = 0; j < 4; ++j) ov[j] = f2bf(acc[j] * inv);
            *reinterpret_cast<bf16x4*>(smem + b * XROW + d4 * 2) = ov;
        }
    }
    __syncthreads();
    // ---- 4. [16 x 128] x [128 x M] per tile and batch column tile
    const int cols = tw * 16;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        if (mt == 1 && M <= 16) break;
        const char* xl = smem + min(fr + 16 * mt, M - 1) * XROW;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const bf16x8 x0 = *reinterpret_cast<const bf16x8*>(xl + c * 128 + WC::x_byte(0, fg));
                const bf16x8 x1 = *reinterpret_cast<const bf16x8*>(xl + c * 128 + WC::x_byte(1, fg));
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[i][c].frag(0), x0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[i][c].frag(1), x1, acc, 0, 0, 0);
            }
            apply_w_scale(w_scale, tile[i] * 16 + fg * 4, acc);
            const int tl = wave + 8 * i, b = fr + 16 * mt;
            if (tl < tw && b < M) *reinterpret_cast<f32x4*>(outs + b * cols + tl * 16 + fg * 4) = acc;
        }
    }
    __syncthreads();
    // ---- 5. add to the residual accumulator: consecutive lanes = consecutive floats of one row
    float* dst = x_acc + (int64_t)tg * cols;
    for (int e = tid; e < M * cols; e += 512) {
        const int b = e / cols, c = e - b * cols;
        atomicAdd(dst + (int64_t)b * ld_acc + c, outs[e]);
    }
}

template <bool W8, int TI>
int launch_oproj_heads(const float* ws, const void* wp, const float* w_scale, float* x_acc, int64_t ld_acc, int M, int N, int heads,
                       int n_split, int tw, kr_stream s) {
    const size_t lds = ((32 * (128 * 2 + 16) + 127) & ~127) + (size_t)M * tw * 16 * 4;
    const dim3 grid((N / 16) / tw, heads);
    if (n_split == 8)
        dec_oproj_heads_kernel<W8, TI, 8><<<grid, 512, lds, kr_hs(s)>>>(ws, reinterpret_cast<const char*>(wp), w_scale, x_acc, ld_acc, M,
                                                                        heads, n_split, tw);
    else if (n_split == 16)
        dec_oproj_heads_kernel<W8, TI, 16><<<grid, 512, lds, kr_hs(s)>>>(ws, reinterpret_cast<const char*>(wp), w_scale, x_acc, ld_acc, M,
                                                                         heads, n_split, tw);
    else
        dec_oproj_heads_kernel<W8, TI, 0><<<grid, 512, lds, kr_hs(s)>>>(ws, reinterpret_cast<const char*>(wp), w_scale, x_acc, ld_acc, M,
                                                                        heads, n_split, tw);
    KR_CHECK_LAUNCH();
    return KR_OK;
}
#endif  // KR_EXPERIMENTS

// =====================================================================================
// decode attention with in-launch merge
// =====================================================================================
// grid = (n_split, kv_heads, batch); WAVES waves; wave `part` = split*WAVES + wave walks 32-key units
// part, part + WAVES*n_split, ...   Layouts as in kr_attention.hip (K rows, V^T 64-key blocks).
#ifndef KR_ATTN_DEC_LD        // -DKR_ATTN_DEC_LD=ld8: default-policy K / V^T loads (A/B builds, csrc/tools/build_variant.py)
#define KR_ATTN_DEC_LD ld8_nt
#endif
template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64) attn_decode2_kernel(const kr_bf16* __restrict__ q, const kr_bf16* __restrict__ kcache,
                                                                  const kr_bf16* __restrict__ vtcache,
                                                                  const int32_t* __restrict__ ctx_len,
                                                                  const int32_t* __restrict__ finished, int heads, int kv_heads,
                                                                  int group, int n_split, int s_max, float scale_log2e,
                                                                  kr_bf16* __restrict__ out, float* __restrict__ ws,
                                                                  int* __restrict__ counters, int ws_bytes) {
    // argument order: everything the first loads need sits in the 16 preloaded dwords (kernarg preload), so the
    // scalar load of ctx_len[b] leaves at once instead of behind a load of the argument tail
    constexpr int HD = 128, DT = HD / 16, REC = HD + 4;
    __shared__ __attribute__((aligned(16))) float o_s[WAVES][16][HD];
    __shared__ float m_s[WAVES][16], l_s[WAVES][16];
    __shared__ int last_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    // group (= heads / kv_heads) and n_split (= gridDim.x) are arguments: a runtime division and a read of the dispatch
    // packet would both sit in front of the first loads
    const int split = blockIdx.x, kvh = blockIdx.y, b = blockIdx.z;
    const int n_part = n_split * WAVES, part = split * WAVES + wave;
    const int ctx = ctx_len[b] + 1;
    // a sequence that has finished (EOS flag set by the sampling launch, or retired by the host): nothing downstream reads its rows
    // any more — its workgroups leave without touching its cache (a server's idle slots: ~18 % of the rows in the corpus run)
    if (finished != nullptr && finished[b] != 0) return;

    const int g = fr < group ? fr : 0;
    // MFMA k-step i pairs K[key][32i + 8fg + j] with Q[g][32i + 8fg + j]: per load instruction the
    // four lane groups cover 64 contiguous bytes of each of 16 key rows
    const kr_bf16* qp = q + ((int64_t)b * heads + kvh * group + g) * HD + fg * 8;
    bf16x8 qf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) qf[i] = ld8(qp + i * 32);
    const int64_t kv_base = (int64_t)b * kv_heads + kvh;
    const kr_bf16* kc = kcache + kv_base * s_max * HD;
    const kr_bf16* vc = vtcache + kv_base * (int64_t)(s_max >> 6) * (HD * 64);

    f32x4 o[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) o[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m_run = -1e30f, l_run = 0.f;
    // unit = 32 keys (half a V^T block); wave `part` takes units part, part + n_part, ...: at the contexts of a page
    // (1.4k .. 2.4k keys = 44 .. 76 units) 64 parts leave one unit — one memory round trip — per wave; a wave with
    // more requests unit u + n_part before it computes unit u
    const int nu = (ctx + 31) >> 5;
    bf16x8 kf[2][4], vf[DT], kf2[2][4], vf2[DT];
    auto load_unit = [&](int u, bf16x8 (&kk)[2][4], bf16x8 (&vv)[DT]) {
        const int key0 = u * 32;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const kr_bf16* kp = kc + (int64_t)(key0 + 8 * (fr >> 2) + 4 * kt + (fr & 3)) * HD + fg * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) kk[kt][i] = KR_ATTN_DEC_LD(kp + i * 32);
        }
        // the unit's half of its V^T block is contiguous ([2][HD][32]: kr_common.h): 16 channel rows x 64 B per instruction = 1 KiB of
        // whole lines (rounds 1-3: [HD][64], half of every line — 4.0 against 6.5 TB/s for this shape, profiles/r04_halfline_read.txt)
        const kr_bf16* vp = vc + (int64_t)(u >> 1) * (HD * 64) + (u & 1) * (HD * 32) + fg * 8;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) vv[dt] = KR_ATTN_DEC_LD(vp + (dt * 16 + fr) * 32);
    };
    int u = part;
    if (u < nu) load_unit(u, kf, vf);
    while (u < nu) {
        const int un = u + n_part;
        if (un < nu) load_unit(un, kf2, vf2);
        {
            const int key0 = u * 32;
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
        if (un < nu) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int i = 0; i < 4; ++i) kf[kt][i] = kf2[kt][i];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) vf[dt] = vf2[dt];
        }
        u = un;
    }
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    // ---- merge the waves through LDS
    if (fr < group) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<f32x4*>(&o_s[wave][fr][dt * 16 + fg * 4]) = o[dt];
        if (fg == 0) {
            m_s[wave][fr] = m_run;
            l_s[wave][fr] = l_run;
        }
    }
    __syncthreads();
    // element t < group * 32 = 4 consecutive channels d4 .. d4+3 of head gg: one 16-byte piece of the record
    // [o[128], m, l, 0, 0] (REC floats) of this (sequence, head, split); a thread owns elements tid, tid + NTHR, ...
    constexpr int NTHR = WAVES * 64, IT = (16 * 32 + NTHR - 1) / NTHR;
    const int bh0 = b * heads + kvh * group;
    const int nq = group * (HD / 4);
    f32x4 acc4[IT];
    float mm[IT], ll[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int t = tid + it * NTHR, gg = t >> 5, d4 = (t & 31) << 2;
        acc4[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mm[it] = -1e30f;
        ll[it] = 0.f;
        if (t < nq) {
#pragma unroll
            for (int w = 0; w < WAVES; ++w) mm[it] = fmaxf(mm[it], m_s[w][gg]);
#pragma unroll
            for (int w = 0; w < WAVES; ++w) {
                const float sc = __builtin_amdgcn_exp2f(m_s[w][gg] - mm[it]);
                const f32x4 ow = *reinterpret_cast<const f32x4*>(&o_s[w][gg][d4]);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc4[it][j] += ow[j] * sc;
                ll[it] += l_s[w][gg] * sc;
            }
        }
    }
    auto store_out = [&](int t, const f32x4& a, float l) {
        const float inv = l > 0.f ? 1.0f / l : 0.f;
        bf16x4 ov;
#pragma unroll
        for (int j = 0; j < 4; ++j) ov[j] = f2bf(a[j] * inv);
        *reinterpret_cast<bf16x4*>(out + (int64_t)(bh0 + (t >> 5)) * HD + ((t & 31) << 2)) = ov;
    };
    auto rec_of = [&](int t) { return ((int64_t)(bh0 + (t >> 5)) * n_split) * REC; };   // first record of element t's head (floats)
    if (n_split == 1 && out) {
#pragma unroll
        for (int it = 0; it < IT; ++it)
            if (tid + it * NTHR < nq) store_out(tid + it * NTHR, acc4[it], ll[it]);
        return;
    }
    if (!KR_EXP || !out) {  // a later launch (kr_attn_decode_merge) merges the partials: plain stores
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int t = tid + it * NTHR, d4 = (t & 31) << 2;
            if (t < nq) {
                float* w = ws + rec_of(t) + (int64_t)split * REC;
                *reinterpret_cast<f32x4*>(w + d4) = acc4[it];
                if (d4 == 0) *reinterpret_cast<f32x4*>(w + HD) = (f32x4){mm[it], ll[it], 0.f, 0.f};
            }
        }
        return;
    }
    // ---- in-launch merge by the last-arriving split of this (sequence, kv head).  Hand-off in the form the guide
    // measures (MI355X_MICROARCH.md, visibility, "Valid forms" row 1): every payload byte leaves as a 16-byte sc1
    // (write-through) store, every storing wave drains its stores (vmcnt(0)) before the workgroup barrier, ONE lane
    // then adds to the group's counter; the workgroup whose add returns n_split - 1 is the last one and reads all
    // records with 16-byte sc1 loads (never a plain load of these bytes), all requested at once.  Nobody waits:
    // the other workgroups just leave.  Correct for any placement of the splits on XCDs / CUs.
    // (Measured r2: 1.2200 ms per step against 1.2033 with the separate merge launch — the hand-off costs what the
    // launch costs; kept for the ABI and as the tested example of the protocol.  CAUTION before reusing it: the same form
    // with 64 KB payloads per workgroup — a split-K GEMM fix-up, profiles/r02_decode_experiments.txt — let the last arriver
    // read a few 16-byte pieces too early in 1 of ~10^6; nothing on the default path depends on an in-launch hand-off.)
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(ws, 0, ws_bytes, 0x00020000);
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int t = tid + it * NTHR, d4 = (t & 31) << 2;
        if (t < nq) {
            const int off = (int)((rec_of(t) + (int64_t)split * REC + d4) * 4);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc4[it]), rsrc, off, 0, 16);
            if (d4 == 0)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, (f32x4){mm[it], ll[it], 0.f, 0.f}), rsrc,
                                                       (int)((rec_of(t) + (int64_t)split * REC + HD) * 4), 0, 16);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's record stores have been acknowledged
    __syncthreads();
    if (tid == 0) {
        const int old = __hip_atomic_fetch_add(counters + b * kv_heads + kvh, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_s = (old == n_split - 1);
        if (old == n_split - 1) __hip_atomic_store(counters + b * kv_heads + kvh, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!last_s) return;
    constexpr int MAXS = 16;
    for (int it = 0; it < IT; ++it) {
        const int t = tid + it * NTHR, d4 = (t & 31) << 2;
        if (t >= nq) continue;
        u32x4 ro[MAXS], rm[MAXS];
#pragma unroll
        for (int p = 0; p < MAXS; ++p) {
            if (p < n_split) {
                ro[p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((rec_of(t) + (int64_t)p * REC + d4) * 4), 0, 16);
                rm[p] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((rec_of(t) + (int64_t)p * REC + HD) * 4), 0, 16);
            }
        }
        float mt = -1e30f;
#pragma unroll
        for (int p = 0; p < MAXS; ++p)
            if (p < n_split) mt = fmaxf(mt, __builtin_bit_cast(f32x4, rm[p])[0]);
        f32x4 at = {0.f, 0.f, 0.f, 0.f};
        float lt = 0.f;
#pragma unroll
        for (int p = 0; p < MAXS; ++p) {
            if (p < n_split) {
                const f32x4 mlp = __builtin_bit_cast(f32x4, rm[p]), op = __builtin_bit_cast(f32x4, ro[p]);
                const float sc = __builtin_amdgcn_exp2f(mlp[0] - mt);
#pragma unroll
                for (int j = 0; j < 4; ++j) at[j] += op[j] * sc;
                lt += mlp[1] * sc;
            }
        }
        store_out(t, at, lt);
    }
}

// Merge of the split-KV partials as a launch of its own: one 128-thread workgroup per (sequence, head).
// (Cheaper end-to-end than replicating the merge in every o_proj workgroup's prologue: measured.)
// NS > 0: all n_split records are requested at once (one memory round trip instead of two dependent loops).
template <int NS>
__global__ void __launch_bounds__(128) attn_merge_kernel(const float* __restrict__ ws, kr_bf16* __restrict__ out, int n_split, int heads_xp) {
    // heads_xp > 0: out is the XP layout of a 17..32-row batch (row = sequence, column = head * 128 + d), heads_xp = heads
    constexpr int HD = 128, REC = HD + 4;
    const int bh = blockIdx.x, d = threadIdx.x;
    float acc = 0.f, ll = 0.f;
    if constexpr (NS > 0) {
        const float* w = ws + (int64_t)bh * NS * REC;
        float m[NS], l[NS], o[NS];
#pragma unroll
        for (int p = 0; p < NS; ++p) {
            m[p] = w[p * REC + HD];
            l[p] = w[p * REC + HD + 1];
            o[p] = w[p * REC + d];
        }
        float mm = -1e30f;
#pragma unroll
        for (int p = 0; p < NS; ++p) mm = fmaxf(mm, m[p]);
#pragma unroll
        for (int p = 0; p < NS; ++p) {
            const float sc = __builtin_amdgcn_exp2f(m[p] - mm);
            acc += o[p] * sc;
            ll += l[p] * sc;
        }
    } else {
        const float* w = ws + (int64_t)bh * n_split * REC;
        float mm = -1e30f;
        for (int p = 0; p < n_split; ++p) mm = fmaxf(mm, w[p * REC + HD]);
        for (int p = 0; p < n_split; ++p) {
            const float sc = __builtin_amdgcn_exp2f(w[p * REC + HD] - mm);
            acc += w[p * REC + d] * sc;
            ll += w[p * REC + HD + 1] * sc;
        }
    }
    const kr_bf16 r = __builtin_bit_cast(kr_bf16, f2bf(ll > 0.f ? acc / ll : 0.f));
    if (heads_xp > 0) {
        const int b = bh / heads_xp, hh = bh - b * heads_xp;
        *reinterpret_cast<kr_bf16*>(reinterpret_cast<char*>(out) + kr_xp_byte_offset(b, hh * HD + d)) = r;
    } else {
        out[(int64_t)bh * HD + d] = r;
    }
}

// =====================================================================================
// temperature sampling as an argmax (Gumbel-max): token = argmax_i( logit_i / T + G_i )
// =====================================================================================
// G_i = -ln(-ln(u_i)), u_i = ((h_i >> 9) + 0.5) * 2^-23, h_i = mix(mix(seed ^ n * 0x9E3779B1) + i) with
// mix = the "lowbias32" integer finaliser and n = the index of the token being generated in its sequence
// (ctx_len + 1 - prompt_len).  A counter-based generator: no state, any (sequence, step, token) draw can be
// recomputed — the oracle does exactly that.  T == 0 rows get no noise: plain argmax, ties to the lowest index.
// The reference's requests carry temperature 0.1 (first attempt, karanta/pipeline.py:281,301) or 0.7
// (VLLMClient.generate default, bulk_processing/workers/vllm_client.py:155); vLLM's own sampler draws from the
// same softmax(logits / T) distribution with a different generator.
__device__ __forceinline__ unsigned kr_mix32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__global__ void __launch_bounds__(256) gumbel_argmax_kernel(const float* __restrict__ logits, int64_t ld, int vocab,
                                                            const float* __restrict__ temperature,
                                                            const unsigned* __restrict__ seed,
                                                            const int32_t* __restrict__ ctx_len,
                                                            const int32_t* __restrict__ prompt_len,
                                                            float* __restrict__ amax_val, int32_t* __restrict__ amax_idx,
                                                            const uint64_t* __restrict__ guide_masks,
                                                            const int32_t* __restrict__ guide_state, int mask_words,
                                                            int fallback_token) {
    __shared__ float s_v[4];
    __shared__ int s_i[4];
    const int p = blockIdx.x, n_part = gridDim.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (((vocab + n_part - 1) / n_part) + 3) & ~3;
    const int i0 = p * per, i1 = min(vocab, i0 + per);
    const float T = temperature[b];
    const float inv_t = T > 0.f ? 1.0f / T : 1.0f;
    const unsigned base = kr_mix32(seed[b] ^ ((unsigned)(ctx_len[b] + 1 - prompt_len[b]) * 0x9E3779B1u));
    const float* row = logits + (int64_t)b * ld;
    // guided slot: allowed-token bits of its DFA state (kr_guide_build_masks); 0 pointer = unconstrained
    const uint32_t* allow = nullptr;
    if (guide_masks != nullptr && guide_masks[b] != 0)
        allow = reinterpret_cast<const uint32_t*>(guide_masks[b]) + (int64_t)guide_state[b] * mask_words;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = i0 + tid; i < i1; i += 256) {
        if (allow != nullptr && !((allow[i >> 5] >> (i & 31)) & 1u)) continue;
        float v = row[i] * inv_t;
        if (T > 0.f) {
            const unsigned h = kr_mix32(base + (unsigned)i);
            // 23-bit integer + 0.5 is exact in f32 (24 significant bits): u in [2^-24, 1 - 2^-24], never 0 or 1
            // (a 24-bit integer + 0.5 rounds to 2^24 at the top code: u = 1, noise = +inf)
            const float u = ((float)(h >> 9) + 0.5f) * 1.1920928955078125e-07f;  // 2^-23
            v += -logf(-logf(u));
        }
        better(bv, bi, v, i);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        better(bv, bi, ov, oi);
    }
    if (lane == 0) {
        s_v[wave] = bv;
        s_i[wave] = bi;
    }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w) better(bv, bi, s_v[w], s_i[w]);
        // a row whose mask allows nothing (cannot happen for a live DFA state) must still yield a valid token id
        if (p == 0 && bi == 0x7fffffff) bi = fallback_token;
        amax_val[(int64_t)b * n_part + p] = bv;
        amax_idx[(int64_t)b * n_part + p] = bi;
    }
}

// =====================================================================================
// greedy sampling from the lm_head partials + per-step bookkeeping
// =====================================================================================
__global__ void __launch_bounds__(256) sample_greedy_kernel(const float* __restrict__ amax_val,
                                                            const int32_t* __restrict__ amax_idx, int n_part,
                                                            const kr_bf16* __restrict__ table, int d,
                                                            int32_t* __restrict__ tokens_out, int32_t* __restrict__ history,
                                                            int hist_stride, const int32_t* __restrict__ prompt_len,
                                                            int32_t* __restrict__ ctx_len, int32_t* __restrict__ finished,
                                                            const int32_t* __restrict__ eos, int n_eos, int pad_id,
                                                            int ignore_eos, kr_bf16* __restrict__ x_next) {
    __shared__ float s_v[4];
    __shared__ int s_i[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    // eight partials per thread requested at once (clamped indices, no branch around the loads): the lm_head leaves up to 2048
    // partials per row, and one load pair per loop iteration was eight dependent L2 round trips in a 6 us launch
    for (int i0 = tid; i0 < n_part; i0 += 256 * 8) {
        float v[8];
        int ix[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = min(i0 + u * 256, n_part - 1);
            v[u] = amax_val[(int64_t)b * n_part + i];
            ix[u] = amax_idx[(int64_t)b * n_part + i];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + u * 256 < n_part) better(bv, bi, v[u], ix[u]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        better(bv, bi, ov, oi);
    }
    if (lane == 0) {
        s_v[wave] = bv;
        s_i[wave] = bi;
    }
    __syncthreads();
    bv = s_v[0];
    bi = s_i[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) better(bv, bi, s_v[w], s_i[w]);
    int tok = bi;
    const int was_finished = finished[b];
    const bool freeze = (ignore_eos & 2) != 0;  // bit 1: a finished sequence stops advancing (slot scheduler)
    ignore_eos &= 1;
    if (was_finished && !ignore_eos) tok = pad_id;
    const int new_ctx = ctx_len[b] + 1;  // tokens cached once the sampled token has been fed back
    __syncthreads();
    if (freeze && was_finished && !ignore_eos) {
        // the slot idles at its last position (its KV row is rewritten in place, its history stays as it is)
        // until the host admits a new request into it
        if (tid == 0) tokens_out[b] = tok;
        for (int c = tid; c < (d >> 3); c += 256) st8(x_next + (int64_t)b * d + c * 8, ld8(table + (int64_t)tok * d + c * 8));
        return;
    }
    if (tid == 0) {
        tokens_out[b] = tok;
        history[(int64_t)(new_ctx - prompt_len[b]) * hist_stride + b] = tok;  // generated-token index of this sequence
        ctx_len[b] = new_ctx;
        if (!ignore_eos && !was_finished) {
            int hit = 0;
            for (int i = 0; i < n_eos; ++i) hit |= (tok == eos[i]);
            if (hit) finished[b] = 1;
        }
    }
    for (int c = tid; c < (d >> 3); c += 256) st8(x_next + (int64_t)b * d + c * 8, ld8(table + (int64_t)tok * d + c * 8));
}

template <int NT, int EPI, int WAVES>
int launch_dec(DecLinArgs& a, int groups, int max_blocks, kr_stream s) {
    a.groups = groups;
    const int nchunks = a.K >> 6;
    const int cpb = (nchunks + a.ksplit - 1) / a.ksplit;
    const int grid_x = (a.ksplit == 1 && max_blocks > 0 && groups > max_blocks) ? max_blocks : groups;
    size_t red = (size_t)(grid_x < groups ? 2 : 1) * WAVES * NT * 256 * 4;  // double buffered when persistent
    // x goes through LDS whenever it fits next to the reduction buffer (wide-K layers at large M read
    // their x fragments from L2 instead: twice the vector-memory instructions, measurably slower)
    size_t xbytes = ((size_t)a.M * (cpb * 128 + 16) + 127) & ~(size_t)127;
    if (a.xmode == 1 && xbytes + red > 160 * 1024) a.xmode = 0;
    if (a.xmode == 0) xbytes = 0;
    if (a.xmode == 3) red = red > (size_t)a.M * (a.K >> 7) * a.attn_split * 4 ? red : (size_t)a.M * (a.K >> 7) * a.attn_split * 4;
    const size_t lds = xbytes + red;
    KR_CHECK_ARG(lds <= 160 * 1024, "kr_linear_decode: needs %zu bytes of LDS (M=%d K=%d ksplit=%d)", lds, a.M, a.K, a.ksplit);
    auto fn = &dec_linear_kernel<NT, EPI, WAVES>;
    static KrPerDeviceOnce attr;
    if (attr.need()) {
        KR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    fn<<<dim3(grid_x, a.ksplit), WAVES * 64, lds, kr_hs(s)>>>(a);
    KR_CHECK_LAUNCH();
    return KR_OK;
}

template <int NT, int EPI>
int launch_dec_w(DecLinArgs& a, int groups, int waves, int max_blocks, kr_stream s) {
    switch (waves) {
        case 4: return launch_dec<NT, EPI, 4>(a, groups, max_blocks, s);
        case 8: return launch_dec<NT, EPI, 8>(a, groups, max_blocks, s);
        case 16: return launch_dec<NT, EPI, 16>(a, groups, max_blocks, s);
        default: kr_set_error("kr_linear_decode: waves=%d (4, 8 or 16)", waves); return KR_ERR_ARG;
    }
}

}  // namespace

// =====================================================================================
// C-ABI
// =====================================================================================
extern "C" int kr_linear_decode(int mode, const kr_bf16* x, int64_t ldx, const kr_bf16* w_packed, const kr_bf16* bias,
                                const kr_bf16* norm_w, float norm_eps, const kr_bf16* residual, int64_t ldr, kr_bf16* out,
                                float* out_f32, int64_t ldc, int M, int N, int K, int waves, int max_blocks, int ksplit, float* ws,
                                int32_t* counters, const float* attn_partials, int attn_split, const float* cs_table,
                                int cs_stride, const int32_t* prompt_len, const int32_t* ctx_len, kr_bf16* q_out,
                                kr_bf16* kcache, kr_bf16* vtcache, int heads, int kv_heads, int s_max, float* amax_val,
                                int32_t* amax_idx, kr_stream s) {
    KR_CHECK_ARG(w_packed && (x || attn_partials), "kr_linear_decode: null pointer");
    KR_CHECK_ARG(M >= 1 && M <= 16, "kr_linear_decode: M=%d must be in 1..16", M);
    KR_CHECK_ARG(N > 0 && N % 16 == 0 && K > 0 && K % 64 == 0, "kr_linear_decode: N=%d K=%d (N%%16, K%%64)", N, K);
    KR_CHECK_ARG(attn_partials || (ldx >= K && (ldx & 7) == 0), "kr_linear_decode: ldx");
    KR_CHECK_ARG(ksplit >= 1 && ksplit <= (K >> 6), "kr_linear_decode: ksplit=%d", ksplit);
    KR_CHECK_ARG(ksplit == 1 || (ws && counters), "kr_linear_decode: split-K needs workspace and counters");
    KR_CHECK_ARG(!norm_w || K <= 4096, "kr_linear_decode: fused RMSNorm supports K <= 4096");
    KR_CHECK_ARG(!(norm_w && attn_partials), "kr_linear_decode: norm and attention-merge prologues are exclusive");
    KR_CHECK_ARG(!attn_partials || (K % 128 == 0 && attn_split >= 1 && ksplit == 1),
                 "kr_linear_decode: attention merge needs K = heads*128, ksplit 1");
    DecLinArgs a{};
    a.x = x; a.ldx = ldx; a.wp = w_packed; a.bias = bias; a.norm_w = norm_w; a.norm_eps = norm_eps;
    a.residual = residual; a.ldr = ldr; a.out = out; a.out_f32 = out_f32; a.ldc = ldc;
    a.M = M; a.N = N; a.K = K; a.ksplit = ksplit; a.ws = ws; a.counters = counters;
    a.attn_ws = attn_partials; a.attn_split = attn_split;
    // x staged in LDS unless the row slice is too large (wide-K layers read x fragments from L2)
    a.xmode = attn_partials ? 3 : norm_w ? 2 : 1;  // 1 falls back to 0 (x from L2) in launch_dec when LDS is short
    a.cs_table = cs_table; a.cs_stride = cs_stride; a.prompt_len = prompt_len; a.ctx_len = ctx_len;
    a.q_out = q_out; a.kcache = kcache; a.vtcache = vtcache;
    a.heads = heads; a.kv_heads = kv_heads; a.s_max = s_max; a.amax_val = amax_val; a.amax_idx = amax_idx;
    const int ntiles = N >> 4;
    switch (mode) {
        case DEPI_PLAIN: {
            KR_CHECK_ARG((out || out_f32) && ldc >= N && (ldc & 3) == 0, "kr_linear_decode: PLAIN output");
            KR_CHECK_ARG(!residual || (ldr & 3) == 0, "kr_linear_decode: ldr");
            if (ntiles >= 1024) return launch_dec_w<2, DEPI_PLAIN>(a, (ntiles + 1) / 2, waves, max_blocks, s);
            return launch_dec_w<1, DEPI_PLAIN>(a, ntiles, waves, max_blocks, s);
        }
        case DEPI_SILU:
            KR_CHECK_ARG(out && N % 32 == 0 && ldc >= N / 2 && (ldc & 3) == 0, "kr_linear_decode: SILU output");
            return launch_dec_w<2, DEPI_SILU>(a, ntiles / 2, waves, max_blocks, s);
        case DEPI_SILU8:
            KR_CHECK_ARG(out && ldc >= N / 2 && (ldc & 3) == 0, "kr_linear_decode: SILU8 output");
            return launch_dec_w<1, DEPI_SILU8>(a, ntiles, waves, max_blocks, s);
        case DEPI_ROPE_KV:
            KR_CHECK_ARG(bias && cs_table && prompt_len && ctx_len && q_out && kcache && vtcache && cs_stride > 0,
                         "kr_linear_decode: ROPE_KV pointers");
            KR_CHECK_ARG(N == (heads + 2 * kv_heads) * 128 && s_max % 64 == 0, "kr_linear_decode: ROPE_KV needs head_dim 128");
            return launch_dec_w<2, DEPI_ROPE_KV>(a, ntiles / 2, waves, max_blocks, s);
        case DEPI_ARGMAX:
            KR_CHECK_ARG(amax_val && amax_idx && ksplit == 1, "kr_linear_decode: ARGMAX pointers / ksplit");
            KR_CHECK_ARG(!out_f32 || ldc >= N, "kr_linear_decode: ARGMAX logits ldc");
            return launch_dec_w<2, DEPI_ARGMAX>(a, (ntiles + 1) / 2, waves, max_blocks, s);
        default:
            kr_set_error("kr_linear_decode: unknown mode %d", mode);
            return KR_ERR_ARG;
    }
}

static int wide_impl(int mode, const kr_bf16* x, int64_t ldx, const void* w_packed, const float* w_scale, const kr_bf16* bias,
                     const kr_bf16* norm_w, float norm_eps, const kr_bf16* residual, int64_t ldr,
                     kr_bf16* out, float* out_f32, int64_t ldc, int M, int N, int K, int blocks, int waves,
                     float* amax_val, int32_t* amax_idx, kr_stream s, bool x_is_f32 = false, kr_bf16* x_out = nullptr,
                     int64_t ldxo = 0) {
    KR_CHECK_ARG(x && w_packed, "kr_linear_decode_wide: null pointer");
    KR_CHECK_ARG(M >= 1 && M <= 32, "kr_linear_decode_wide: M=%d must be in 1..32", M);
    KR_CHECK_ARG(N > 0 && N % 16 == 0 && K > 0 && K % 512 == 0 && K <= 4096,
                 "kr_linear_decode_wide: N=%d K=%d (N%%16, K%%512, K<=4096)", N, K);
    KR_CHECK_ARG(ldx >= K && (ldx & 7) == 0, "kr_linear_decode_wide: ldx");
    KR_CHECK_ARG(blocks > 0 && waves >= 1 && waves <= 8, "kr_linear_decode_wide: blocks=%d waves=%d", blocks, waves);
    DecLinArgs a{};
    a.x_is_f32 = x_is_f32 ? 1 : 0; a.wide_x_out = x_out; a.wide_ldxo = ldxo;
    a.x = x; a.ldx = ldx; a.wp = reinterpret_cast<const kr_bf16*>(w_packed); a.w_scale = w_scale; a.bias = bias;
    a.norm_w = norm_w; a.norm_eps = norm_eps;
    a.residual = residual; a.ldr = ldr; a.out = out; a.out_f32 = out_f32; a.ldc = ldc;
    a.M = M; a.N = N; a.K = K; a.ksplit = 1; a.amax_val = amax_val; a.amax_idx = amax_idx;
    a.wide_blocks = blocks; a.wide_waves = waves;
    if (mode & KR_DEC_OUT_XP) {   // SILU8 output in the packed layout of 17..32-row batches
        KR_CHECK_ARG((mode & ~KR_DEC_OUT_XP) == DEPI_SILU8 && M > 16 && (N / 2) % 64 == 0 && ((uintptr_t)out & 15) == 0,
                     "kr_linear_decode_wide: KR_DEC_OUT_XP is for SILU8 at 17..32 rows, N/2 %% 64 == 0");
        a.out_xp = 1;
        mode &= ~KR_DEC_OUT_XP;
    }
    switch (mode) {
        case DEPI_PLAIN:
            KR_CHECK_ARG((out || out_f32) && ldc >= N && (ldc & 3) == 0 && (!residual || (ldr & 3) == 0), "kr_linear_decode_wide: PLAIN output");
            return launch_wide<DEPI_PLAIN>(a, blocks, waves, s);
        case DEPI_SILU8:
            KR_CHECK_ARG(out && ldc >= N / 2 && (ldc & 3) == 0, "kr_linear_decode_wide: SILU8 output");
            return launch_wide<DEPI_SILU8>(a, blocks, waves, s);
        case DEPI_ARGMAX:
            KR_CHECK_ARG(amax_val && amax_idx && (!out_f32 || ldc >= N), "kr_linear_decode_wide: ARGMAX pointers");
            return launch_wide<DEPI_ARGMAX>(a, blocks, waves, s);
        default:
            kr_set_error("kr_linear_decode_wide: mode %d not supported (PLAIN, SILU8, ARGMAX)", mode);
            return KR_ERR_ARG;
    }
}

extern "C" int kr_linear_decode_wide(int mode, const kr_bf16* x, int64_t ldx, const kr_bf16* w_packed, const kr_bf16* bias,
                                     const kr_bf16* norm_w, float norm_eps, const kr_bf16* residual, int64_t ldr,
                                     kr_bf16* out, float* out_f32, int64_t ldc, int M, int N, int K, int blocks, int waves,
                                     float* amax_val, int32_t* amax_idx, kr_stream s) {
    return wide_impl(mode, x, ldx, w_packed, nullptr, bias, norm_w, norm_eps, residual, ldr, out, out_f32, ldc, M, N, K, blocks,
                     waves, amax_val, amax_idx, s);
}

extern "C" int kr_linear_decode_wide_fp8(int mode, const kr_bf16* x, int64_t ldx, const uint8_t* w_packed_fp8,
                                         const float* w_scale, const kr_bf16* bias, const kr_bf16* norm_w, float norm_eps,
                                         const kr_bf16* residual, int64_t ldr, kr_bf16* out, float* out_f32, int64_t ldc,
                                         int M, int N, int K, int blocks, int waves, float* amax_val, int32_t* amax_idx,
                                         kr_stream s) {
    KR_CHECK_ARG(w_scale, "kr_linear_decode_wide_fp8: w_scale is NULL");
    return wide_impl(mode, x, ldx, w_packed_fp8, w_scale, bias, norm_w, norm_eps, residual, ldr, out, out_f32, ldc, M, N, K,
                     blocks, waves, amax_val, amax_idx, s);
}

static int narrow_impl(int mode, const kr_bf16* x, int64_t ldx, const float* part_in, int n_part_in,
                                       kr_bf16* x_out, int64_t ldxo, const void* w_packed, const float* w_scale, const kr_bf16* bias,
                                       const kr_bf16* norm_w, float norm_eps, const kr_bf16* residual, int64_t ldr,
                                       kr_bf16* out, float* out_f32, int64_t ldc, int M, int N, int K, int waves, int ksplit,
                                       const float* cs_table, int cs_stride, const int32_t* prompt_len, const int32_t* ctx_len,
                                       kr_bf16* q_out, kr_bf16* kcache, kr_bf16* vtcache, int heads, int kv_heads, int s_max,
                                       const kr_narrow_opts* opts, kr_stream s, float* x_out_f32 = nullptr, int64_t ldxf = 0,
                                       const void* pf_ptr = nullptr, size_t pf_bytes = 0, int pf_blocks = 0) {
    KR_CHECK_ARG(x && w_packed, "kr_linear_decode_narrow: null pointer");
    const kr_narrow_opts o = opts ? *opts : kr_narrow_opts{};
    KR_CHECK_ARG((o.zero_ptr || o.zero_bytes == 0) && ((uintptr_t)o.zero_ptr & 15) == 0 && (o.zero_bytes & 15) == 0 && o.zero_bytes < (1u << 30),
                 "kr_linear_decode_narrow: opts zero range");
    KR_CHECK_ARG(o.part_rows >= 0 && o.part_rows <= 32, "kr_linear_decode_narrow: opts part_rows=%d", o.part_rows);
    KR_CHECK_ARG(M >= 1 && M <= 32, "kr_linear_decode_narrow: M=%d must be in 1..32", M);
    KR_CHECK_ARG(N > 0 && N % 16 == 0 && K > 0 && K % 64 == 0, "kr_linear_decode_narrow: N=%d K=%d (N%%16, K%%64)", N, K);
    KR_CHECK_ARG(ldx >= K && (ldx & 7) == 0, "kr_linear_decode_narrow: ldx");
    KR_CHECK_ARG(waves == 8 || waves == 16, "kr_linear_decode_narrow: waves=%d (8 or 16)", waves);
    KR_CHECK_ARG(ksplit >= 1 && ksplit <= 8 && ksplit <= (K >> 6), "kr_linear_decode_narrow: ksplit=%d", ksplit);
    KR_CHECK_ARG(!norm_w || K <= 4096, "kr_linear_decode_narrow: fused RMSNorm supports K <= 4096");
    KR_CHECK_ARG(!part_in || (norm_w && (n_part_in == 1 || n_part_in == 2) && x_out && x_out != x && ldxo >= K && (ldxo & 7) == 0),
                 "kr_linear_decode_narrow: partial sums need the norm prologue, 1 or 2 slabs and a separate x_out");
    DecLinArgs a{};
    a.x = x; a.ldx = ldx; a.wp = reinterpret_cast<const kr_bf16*>(w_packed); a.w_scale = w_scale; a.bias = bias;
    a.norm_w = norm_w; a.norm_eps = norm_eps;
    a.residual = residual; a.ldr = ldr; a.out = out; a.out_f32 = out_f32; a.ldc = ldc;
    a.M = M; a.N = N; a.K = K; a.ksplit = ksplit;
    a.part_in = part_in; a.x_out = x_out; a.ldxo = ldxo;
    a.x_out_f32 = x_out_f32; a.ldxf = ldxf;
    a.part_rows = o.part_rows;
    a.n_part = part_in ? n_part_in : 0;
    a.zero_ptr = o.zero_bytes ? o.zero_ptr : nullptr; a.zero_n16 = (int)(o.zero_bytes >> 4); a.part_atomic = o.atomic_out ? 1 : 0;
    KR_CHECK_ARG(!a.part_atomic || (ksplit == 2 && mode == DEPI_PLAIN), "kr_linear_decode_narrow: atomic_out is for ksplit 2 (order-free sum)");
    KR_CHECK_ARG(a.part_rows == 0 || (part_in && a.part_rows >= M), "kr_linear_decode_narrow: part rows %d < M %d", a.part_rows, M);
    if (KR_EXP && pf_blocks > 0) {   // experiment builds: prefetch workgroups ride on this launch
        KR_CHECK_ARG(ksplit == 1 && pf_ptr && ((uintptr_t)pf_ptr & 15) == 0 && pf_blocks <= 1024, "kr_linear_decode_narrow_pf: bad args");
        a.pf_ptr = reinterpret_cast<const char*>(pf_ptr); a.pf_bytes = (int64_t)pf_bytes; a.pf_blocks = pf_bytes >= 16 ? pf_blocks : 0;
    }
    a.cs_table = cs_table; a.cs_stride = cs_stride; a.prompt_len = prompt_len; a.ctx_len = ctx_len;
    a.q_out = q_out; a.kcache = kcache; a.vtcache = vtcache;
    a.heads = heads; a.kv_heads = kv_heads; a.s_max = s_max;
    const int ntiles = N >> 4;
    switch (mode) {
        case DEPI_PLAIN:
            KR_CHECK_ARG(ldc >= N && (ldc & 3) == 0 && (!residual || (ldr & 3) == 0), "kr_linear_decode_narrow: PLAIN ldc / ldr");
            if (ksplit > 1) {  // deferred split-K: f32 slabs [ksplit][M][ldc], summed by the consumer
                KR_CHECK_ARG(out_f32 && !out && !bias && !residual && !norm_w,
                             "kr_linear_decode_narrow: split-K writes f32 slabs only (no bias / residual / norm)");
                return waves == 16 ? launch_narrow_direct<DEPI_PARTIAL, 16>(a, ntiles, s)
                                   : launch_narrow_direct<DEPI_PARTIAL, 8>(a, ntiles, s);
            }
            KR_CHECK_ARG(out || out_f32, "kr_linear_decode_narrow: PLAIN output");
            if (norm_w) return launch_narrow_norm<1, DEPI_PLAIN>(a, ntiles, s);  // x rows live in registers: 8 waves
            return waves == 16 ? launch_narrow_direct<DEPI_PLAIN, 16>(a, ntiles, s)
                               : launch_narrow_direct<DEPI_PLAIN, 8>(a, ntiles, s);
        case DEPI_ROPE_KV:
            KR_CHECK_ARG(bias && cs_table && prompt_len && ctx_len && q_out && kcache && vtcache && cs_stride > 0,
                         "kr_linear_decode_narrow: ROPE_KV pointers");
            KR_CHECK_ARG(N == (heads + 2 * kv_heads) * 128 && s_max % 64 == 0 && ksplit == 1,
                         "kr_linear_decode_narrow: ROPE_KV needs head_dim 128, ksplit 1");
            if (!norm_w) {
                // x is ALREADY normalised (kr_decode_resnorm ran the residual sum + RMSNorm once for the batch): the x
                // fragments come straight from L2, no staging and no norm arithmetic per workgroup — the form of decode
                // batches above 16 rows, where every one of the 64-144 workgroups used to stage 32 rows of x + slabs
                KR_CHECK_ARG(!part_in && waves == 8, "kr_linear_decode_narrow: ROPE_KV without a norm takes no partial sums, 8 waves");
                if (narrow_share(a, 8) <= 3) return launch_narrow_u<2, DEPI_ROPE_KV, 8, 0, 0, false, 3>(a, ntiles / 2, s);
                return launch_narrow_u<2, DEPI_ROPE_KV, 8, 0, 0, false, 5>(a, ntiles / 2, s);
            }
            return launch_narrow_norm<2, DEPI_ROPE_KV>(a, ntiles / 2, s);
        default:
            kr_set_error("kr_linear_decode_narrow: mode %d not supported (PLAIN, ROPE_KV)", mode);
            return KR_ERR_ARG;
    }
}

extern "C" int kr_linear_decode_narrow(int mode, const kr_bf16* x, int64_t ldx, const float* part_in, int n_part_in,
                                       kr_bf16* x_out, int64_t ldxo, const kr_bf16* w_packed, const kr_bf16* bias,
                                       const kr_bf16* norm_w, float norm_eps, const kr_bf16* residual, int64_t ldr,
                                       kr_bf16* out, float* out_f32, int64_t ldc, int M, int N, int K, int waves, int ksplit,
                                       const float* cs_table, int cs_stride, const int32_t* prompt_len, const int32_t* ctx_len,
                                       kr_bf16* q_out, kr_bf16* kcache, kr_bf16* vtcache, int heads, int kv_heads, int s_max,
                                       const kr_narrow_opts* opts, kr_stream s) {
    return narrow_impl(mode, x, ldx, part_in, n_part_in, x_out, ldxo, w_packed, nullptr, bias, norm_w, norm_eps, residual, ldr, out,
                       out_f32, ldc, M, N, K, waves, ksplit, cs_table, cs_stride, prompt_len, ctx_len, q_out, kcache, vtcache,
                       heads, kv_heads, s_max, opts, s);
}

extern "C" int kr_linear_decode_narrow_fp8(int mode, const kr_bf16* x, int64_t ldx, const float* part_in, int n_part_in,
                                           kr_bf16* x_out, int64_t ldxo, const uint8_t* w_packed_fp8, const float* w_scale,
                                           const kr_bf16* bias, const kr_bf16* norm_w, float norm_eps, const kr_bf16* residual,
                                           int64_t ldr, kr_bf16* out, float* out_f32, int64_t ldc, int M, int N, int K, int waves,
                                           int ksplit, const float* cs_table, int cs_stride, const int32_t* prompt_len,
                                           const int32_t* ctx_len, kr_bf16* q_out, kr_bf16* kcache, kr_bf16* vtcache, int heads,
                                           int kv_heads, int s_max, const kr_narrow_opts* opts, kr_stream s) {
    KR_CHECK_ARG(w_scale, "kr_linear_decode_narrow_fp8: w_scale is NULL");
    return narrow_impl(mode, x, ldx, part_in, n_part_in, x_out, ldxo, w_packed_fp8, w_scale, bias, norm_w, norm_eps, residual, ldr,
                       out, out_f32, ldc, M, N, K, waves, ksplit, cs_table, cs_stride, prompt_len, ctx_len, q_out, kcache, vtcache,
                       heads, kv_heads, s_max, opts, s);
}

static int attn_decode_impl(const kr_bf16* q, const kr_bf16* kcache, const kr_bf16* vtcache, const int32_t* ctx_len,
                            const int32_t* finished, kr_bf16* out, float* workspace, int32_t* counters, int batch, int heads,
                            int kv_heads, int hd, int s_max, int n_split, float scale, kr_stream s) {
    KR_CHECK_ARG(q && kcache && vtcache && ctx_len && (out || workspace), "kr_attn_decode_fused: null pointer");
    KR_CHECK_ARG(hd == 128, "kr_attn_decode_fused: hd=%d (only 128)", hd);
    KR_CHECK_ARG(heads % kv_heads == 0 && heads / kv_heads <= 16, "kr_attn_decode_fused: GQA group must be <= 16");
    KR_CHECK_ARG(batch > 0 && n_split > 0 && s_max % 64 == 0, "kr_attn_decode_fused: bad sizes");
    KR_CHECK_ARG(KR_EXP || !out || n_split == 1,
                 "kr_attn_decode_fused: the in-launch merge (out != NULL with n_split > 1) is an experiment build (-DKR_EXPERIMENTS); "
                 "pass out = NULL and run kr_attn_decode_merge");
    KR_CHECK_ARG(!out || n_split <= 16, "kr_attn_decode_fused: the in-launch merge takes at most 16 splits");
    KR_CHECK_ARG(!out || n_split == 1 || (workspace && counters), "kr_attn_decode_fused: split needs workspace + counters");
    KR_CHECK_ARG(workspace || n_split == 1, "kr_attn_decode_fused: the split partials need a workspace");
    // Workgroup shape: n_split x WAVES parts of 32-key units, so that at page contexts (1.4k .. 2.4k keys = 44 .. 76 units)
    // a wave fetches ONE unit (one memory round trip).  r2 chain timings (B = 8, ctx 1906, launch + dependent-launch gap):
    // 8 splits x 8 waves (128 workgroups) 7.8 us; 16 splits x 4 waves (256 workgroups, every CU loads) 5.8 us;
    // 6 x 8 (two units per wave) 12.7 us.  Hence: up to 8 splits 8 waves, up to 16 splits 4 waves, beyond 2 waves
    // (KARANTA_ATTN_WAVES = 2 / 4 / 8 overrides for A/B runs).
    static const int waves_env = [] { const char* e = getenv("KARANTA_ATTN_WAVES"); return e ? atoi(e) : 0; }();
    const int waves = waves_env ? waves_env : (n_split <= 8 ? 8 : n_split <= 16 ? 4 : 2);
    KR_CHECK_ARG(waves == 2 || waves == 4 || waves == 8, "kr_attn_decode_fused: KARANTA_ATTN_WAVES=%d (2, 4 or 8)", waves);
    const int64_t ws_bytes = (int64_t)batch * heads * n_split * (hd + 4) * 4;
    KR_CHECK_ARG(ws_bytes < ((int64_t)1 << 31), "kr_attn_decode_fused: workspace of %lld bytes", (long long)ws_bytes);
    const dim3 grid(n_split, kv_heads, batch);
    const float sl2 = scale * 1.4426950408889634f;
    const int group = heads / kv_heads;
    if (waves == 8)
        attn_decode2_kernel<8><<<grid, 512, 0, kr_hs(s)>>>(q, kcache, vtcache, ctx_len, finished, heads, kv_heads, group, n_split, s_max, sl2, out,
                                                           workspace, counters, (int)ws_bytes);
    else if (waves == 4)
        attn_decode2_kernel<4><<<grid, 256, 0, kr_hs(s)>>>(q, kcache, vtcache, ctx_len, finished, heads, kv_heads, group, n_split, s_max, sl2, out,
                                                           workspace, counters, (int)ws_bytes);
    else
        attn_decode2_kernel<2><<<grid, 128, 0, kr_hs(s)>>>(q, kcache, vtcache, ctx_len, finished, heads, kv_heads, group, n_split, s_max, sl2, out,
                                                           workspace, counters, (int)ws_bytes);
    KR_CHECK_LAUNCH();
    return KR_OK;
}

extern "C" int kr_attn_decode_fused(const kr_bf16* q, const kr_bf16* kcache, const kr_bf16* vtcache, const int32_t* ctx_len,
                                    kr_bf16* out, float* workspace, int32_t* counters, int batch, int heads, int kv_heads,
                                    int hd, int s_max, int n_split, float scale, kr_stream s) {
    return attn_decode_impl(q, kcache, vtcache, ctx_len, nullptr, out, workspace, counters, batch, heads, kv_heads, hd, s_max, n_split, scale, s);
}

extern "C" int kr_attn_decode_slots(const kr_bf16* q, const kr_bf16* kcache, const kr_bf16* vtcache, const int32_t* ctx_len,
                                    const int32_t* finished, float* workspace, int batch, int heads, int kv_heads, int hd, int s_max,
                                    int n_split, float scale, kr_stream s) {
    KR_CHECK_ARG(finished && workspace, "kr_attn_decode_slots: null pointer");
    return attn_decode_impl(q, kcache, vtcache, ctx_len, finished, nullptr, workspace, nullptr, batch, heads, kv_heads, hd, s_max, n_split, scale, s);
}

// ---- residual sum + RMSNorm ONCE per batch (decode batches above 16 rows)
namespace {
// x_new = bf16(x + slab_0 + slab_1 + ...) (in that order), h = norm_w * bf16(x_new * rsqrt(mean(x_new^2) + eps)).
// One wave per row, lane l owns the 8-element chunks l, l + 64, ...: exactly the summation structure of the narrow NORM
// kernel's prologue (dec_narrow_kernel, section 3), so the rows it writes are bit-identical to the ones that kernel
// stages — a page's tokens stay independent of the batch it decodes in.
// NP = number of slabs as a template constant (0, 1, 2: every slab load is requested up front, next to x and the norm weight —
// with a run-time count the loads sat inside a loop over the row's pieces, one memory round trip per piece: 5.8 us per launch at
// K = 3584, r4) or -1 for any count (the general form, slab loads per piece).
template <int RL, int NP>
__global__ void __launch_bounds__(256) dec_resnorm_kernel(const kr_bf16* __restrict__ x, int64_t ldx, const float* __restrict__ parts,
                                                          int n_part, int part_rows, kr_bf16* __restrict__ x_out, int64_t ldxo,
                                                          const kr_bf16* __restrict__ norm_w, float eps, kr_bf16* __restrict__ h,
                                                          int64_t ldh, int M, int K, int h_xp) {
    const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6), kc = K >> 3;
    if (b >= M) return;
    constexpr int NPR = NP > 0 ? NP : 1;
    bf16x8 xv[RL], nw[RL];
    f32x4 pv[NPR][RL][2];
#pragma unroll
    for (int i = 0; i < RL; ++i) {
        const int c = min(lane + i * 64, kc - 1);        // clamped, not branched: the loads of a masked piece are discarded
        xv[i] = ld8(x + (int64_t)b * ldx + c * 8);
        nw[i] = ld8(norm_w + c * 8);
        if constexpr (NP > 0) {
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const float* pp = parts + ((int64_t)k * part_rows + b) * K + c * 8;
                pv[k][i][0] = *reinterpret_cast<const f32x4*>(pp);
                pv[k][i][1] = *reinterpret_cast<const f32x4*>(pp + 4);
            }
        }
    }
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < RL; ++i) {
        const int c = lane + i * 64;
        if (c < kc) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = bf2f(xv[i][j]);
            if constexpr (NP == 2) {
                if (h_xp & 2) {   // the two slabs are the two K ranges of a group-split down_proj: x + (s0 + s1), the <= 16-row sum
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j] += pv[0][i][0][j] + pv[1][i][0][j];
                        v[4 + j] += pv[0][i][1][j] + pv[1][i][1][j];
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < NP; ++k) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            v[j] += pv[k][i][0][j];
                            v[4 + j] += pv[k][i][1][j];
                        }
                    }
                }
            } else if constexpr (NP > 0) {
#pragma unroll
                for (int k = 0; k < NP; ++k) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j] += pv[k][i][0][j];
                        v[4 + j] += pv[k][i][1][j];
                    }
                }
            } else if constexpr (NP < 0) {
                for (int k = 0; k < n_part; ++k) {
                    const float* pp = parts + ((int64_t)k * part_rows + b) * K + c * 8;
                    const f32x4 p0 = *reinterpret_cast<const f32x4*>(pp), p1 = *reinterpret_cast<const f32x4*>(pp + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j] += p0[j];
                        v[4 + j] += p1[j];
                    }
                }
            }
            if (NP != 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) xv[i][j] = f2bf(v[j]);
                *reinterpret_cast<bf16x8*>(x_out + (int64_t)b * ldxo + c * 8) = xv[i];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) ss += bf2f(xv[i][j]) * bf2f(xv[i][j]);
        }
    }
    ss = wave_sum(ss);
    const float rs = rsqrtf(ss / (float)K + eps);
#pragma unroll
    for (int i = 0; i < RL; ++i) {
        const int c = lane + i * 64;
        if (c < kc) {
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(nw[i][j]) * bfround(bf2f(xv[i][j]) * rs));
            if (h_xp & 1) *reinterpret_cast<bf16x8*>(reinterpret_cast<char*>(h) + kr_xp_byte_offset(b, c * 8)) = o;
            else *reinterpret_cast<bf16x8*>(h + (int64_t)b * ldh + c * 8) = o;
        }
    }
}

template <int RL>
void launch_resnorm(int blocks, kr_stream s, const kr_bf16* x, int64_t ldx, const float* part_in, int n_part_in, int pr, kr_bf16* x_out,
                    int64_t ldxo, const kr_bf16* norm_w, float norm_eps, kr_bf16* h, int64_t ldh, int M, int K, int h_xp) {
#define KR_RESNORM(NP) dec_resnorm_kernel<RL, NP><<<blocks, 256, 0, kr_hs(s)>>>(x, ldx, part_in, n_part_in, pr, x_out, ldxo, norm_w, norm_eps, h, ldh, M, K, h_xp)
    if (n_part_in == 0) KR_RESNORM(0);
    else if (n_part_in == 1) KR_RESNORM(1);
    else if (n_part_in == 2 && (RL <= 4 || (h_xp & 2))) KR_RESNORM(2);
    else KR_RESNORM(-1);
#undef KR_RESNORM
}

int resnorm_impl(const kr_bf16* x, int64_t ldx, const float* part_in, int n_part_in, int part_rows, kr_bf16* x_out, int64_t ldxo,
                 const kr_bf16* norm_w, float norm_eps, kr_bf16* h, int64_t ldh, int M, int K, int h_xp, kr_stream s) {
    KR_CHECK_ARG(x && norm_w && h, "kr_decode_resnorm: null pointer");
    KR_CHECK_ARG(M >= 1 && M <= 32 && K > 0 && K % 8 == 0 && K <= 4096, "kr_decode_resnorm: M=%d K=%d (M <= 32, K %% 8, K <= 4096)", M, K);
    KR_CHECK_ARG(ldx >= K && (ldx & 7) == 0 && ((h_xp & 1) ? (K % 64 == 0 && ((uintptr_t)h & 15) == 0) : (ldh >= K && (ldh & 7) == 0)),
                 "kr_decode_resnorm: ldx / ldh");
    KR_CHECK_ARG(!(h_xp & 2) || n_part_in == 2, "kr_decode_resnorm32: sum_slabs_first takes exactly two slabs");
    KR_CHECK_ARG(n_part_in >= 0 && n_part_in <= 8 && (n_part_in == 0 || (part_in && x_out && x_out != x && ldxo >= K && (ldxo & 7) == 0)),
                 "kr_decode_resnorm: partial sums need a separate x_out");
    KR_CHECK_ARG(n_part_in == 0 || part_rows == 0 || part_rows >= M, "kr_decode_resnorm: part_rows %d < M %d", part_rows, M);
    const int pr = part_rows > 0 ? part_rows : M, blocks = (M + 3) / 4;
    if (K <= 1536) launch_resnorm<3>(blocks, s, x, ldx, part_in, n_part_in, pr, x_out, ldxo, norm_w, norm_eps, h, ldh, M, K, h_xp);
    else if (K <= 2048) launch_resnorm<4>(blocks, s, x, ldx, part_in, n_part_in, pr, x_out, ldxo, norm_w, norm_eps, h, ldh, M, K, h_xp);
    else if (K <= 3584) launch_resnorm<7>(blocks, s, x, ldx, part_in, n_part_in, pr, x_out, ldxo, norm_w, norm_eps, h, ldh, M, K, h_xp);
    else launch_resnorm<8>(blocks, s, x, ldx, part_in, n_part_in, pr, x_out, ldxo, norm_w, norm_eps, h, ldh, M, K, h_xp);
    KR_CHECK_LAUNCH();
    return KR_OK;
}
}  // namespace

extern "C" int kr_decode_resnorm(const kr_bf16* x, int64_t ldx, const float* part_in, int n_part_in, int part_rows, kr_bf16* x_out,
                                 int64_t ldxo, const kr_bf16* norm_w, float norm_eps, kr_bf16* h, int64_t ldh, int M, int K, kr_stream s) {
    return resnorm_impl(x, ldx, part_in, n_part_in, part_rows, x_out, ldxo, norm_w, norm_eps, h, ldh, M, K, 0, s);
}

extern "C" int kr_decode_resnorm32(const kr_bf16* x, int64_t ldx, const float* part_in, int n_part_in, int part_rows, kr_bf16* x_out,
                                   int64_t ldxo, const kr_bf16* norm_w, float norm_eps, kr_bf16* h_xp, int M, int K, int sum_slabs_first,
                                   kr_stream s) {
    return resnorm_impl(x, ldx, part_in, n_part_in, part_rows, x_out, ldxo, norm_w, norm_eps, h_xp, 0, M, K, 1 | (sum_slabs_first ? 2 : 0), s);
}

static int merge_impl(const float* workspace, kr_bf16* out, int batch, int heads, int hd, int n_split, int xp, kr_stream s) {
    KR_CHECK_ARG(workspace && out && batch > 0 && heads > 0 && n_split > 0, "kr_attn_decode_merge: bad args");
    KR_CHECK_ARG(hd == 128, "kr_attn_decode_merge: hd=%d (only 128)", hd);
    KR_CHECK_ARG(!xp || (batch <= 32 && ((uintptr_t)out & 15) == 0), "kr_attn_decode_merge32: batch=%d (<= 32)", batch);
    const int hx = xp ? heads : 0;
    switch (n_split) {
        case 4: attn_merge_kernel<4><<<batch * heads, 128, 0, kr_hs(s)>>>(workspace, out, n_split, hx); break;
        case 8: attn_merge_kernel<8><<<batch * heads, 128, 0, kr_hs(s)>>>(workspace, out, n_split, hx); break;
        case 16: attn_merge_kernel<16><<<batch * heads, 128, 0, kr_hs(s)>>>(workspace, out, n_split, hx); break;
        case 32: attn_merge_kernel<32><<<batch * heads, 128, 0, kr_hs(s)>>>(workspace, out, n_split, hx); break;
        default: attn_merge_kernel<0><<<batch * heads, 128, 0, kr_hs(s)>>>(workspace, out, n_split, hx);
    }
    KR_CHECK_LAUNCH();
    return KR_OK;
}

extern "C" int kr_attn_decode_merge(const float* workspace, kr_bf16* out, int batch, int heads, int hd, int n_split,
                                    kr_stream s) {
    return merge_impl(workspace, out, batch, heads, hd, n_split, 0, s);
}

extern "C" int kr_attn_decode_merge32(const float* workspace, kr_bf16* out_xp, int batch, int heads, int hd, int n_split,
                                      kr_stream s) {
    return merge_impl(workspace, out_xp, batch, heads, hd, n_split, 1, s);
}

extern "C" int kr_gumbel_argmax_guided(const float* logits, int64_t ld_logits, int vocab, const float* temperature,
                                       const uint32_t* seed, const int32_t* ctx_len, const int32_t* prompt_len,
                                       float* amax_val, int32_t* amax_idx, int n_part, int batch,
                                       const uint64_t* guide_masks, const int32_t* guide_state, int mask_words,
                                       int fallback_token, kr_stream s) {
    KR_CHECK_ARG(logits && temperature && seed && ctx_len && prompt_len && amax_val && amax_idx, "kr_gumbel_argmax: null pointer");
    KR_CHECK_ARG(vocab > 0 && ld_logits >= vocab && n_part > 0 && n_part <= 65535 && batch > 0, "kr_gumbel_argmax: bad sizes");
    KR_CHECK_ARG(guide_masks == nullptr || (guide_state != nullptr && (int64_t)mask_words * 32 >= vocab),
                 "kr_gumbel_argmax_guided: guide_state missing or mask_words * 32 < vocab");
    KR_CHECK_ARG(fallback_token >= 0 && fallback_token < vocab, "kr_gumbel_argmax_guided: fallback_token out of vocabulary");
    gumbel_argmax_kernel<<<dim3(n_part, batch), 256, 0, kr_hs(s)>>>(logits, ld_logits, vocab, temperature, seed, ctx_len, prompt_len,
                                                                    amax_val, amax_idx, guide_masks, guide_state, mask_words,
                                                                    fallback_token);
    KR_CHECK_LAUNCH();
    return KR_OK;
}

extern "C" int kr_gumbel_argmax(const float* logits, int64_t ld_logits, int vocab, const float* temperature,
                                const uint32_t* seed, const int32_t* ctx_len, const int32_t* prompt_len, float* amax_val,
                                int32_t* amax_idx, int n_part, int batch, kr_stream s) {
    return kr_gumbel_argmax_guided(logits, ld_logits, vocab, temperature, seed, ctx_len, prompt_len, amax_val, amax_idx, n_part,
                                   batch, nullptr, nullptr, 0, 0, s);
}

extern "C" int kr_sample_greedy(const float* amax_val, const int32_t* amax_idx, int n_part, const kr_bf16* embed_table,
                                int d, int32_t* tokens_out, int32_t* history, int hist_stride, const int32_t* prompt_len,
                                int32_t* ctx_len, int32_t* finished, const int32_t* eos, int n_eos, int pad_id,
                                int ignore_eos, kr_bf16* x_next, int batch, kr_stream s) {
    KR_CHECK_ARG(amax_val && amax_idx && embed_table && tokens_out && history && prompt_len && ctx_len && finished && x_next,
                 "kr_sample_greedy: null pointer");
    KR_CHECK_ARG(n_part > 0 && batch > 0 && (d & 7) == 0 && hist_stride >= batch && (n_eos == 0 || eos),
                 "kr_sample_greedy: bad sizes");
    sample_greedy_kernel<<<batch, 256, 0, kr_hs(s)>>>(amax_val, amax_idx, n_part, embed_table, d, tokens_out, history,
                                                      hist_stride, prompt_len, ctx_len, finished, eos, n_eos, pad_id,
                                                      ignore_eos, x_next);
    KR_CHECK_LAUNCH();
    return KR_OK;
}


#ifdef KR_EXPERIMENTS
// ---- fast-residual mode entry points (see dec_oproj_heads_kernel); include/karanta_hip_experiments.h
extern "C" int kr_oproj_heads(const float* attn_partials, int n_split, const void* w_packed, const float* w_scale, float* x_acc,
                              int64_t ld_acc, int M, int N, int heads, kr_stream s) {
    KR_CHECK_ARG(attn_partials && w_packed && x_acc, "kr_oproj_heads: null pointer");
    KR_CHECK_ARG(M >= 1 && M <= 32 && heads >= 1 && n_split >= 1 && n_split <= 16, "kr_oproj_heads: M=%d heads=%d n_split=%d", M, heads, n_split);
    KR_CHECK_ARG(N > 0 && N % 16 == 0 && ld_acc >= N, "kr_oproj_heads: N=%d ld_acc=%lld", N, (long long)ld_acc);
    const int tiles = N / 16;
    // tiles per workgroup: a divisor of the tile count that 8 waves cover in two rounds (<= 16), the largest one that
    // still gives the launch >= 96 workgroups (2B: 12 tiles -> 8 x 12 heads = 96; 7B: 16 -> 14 x 28 = 392)
    int tw = 0;
    for (int t = 16; t >= 1 && !tw; --t)
        if (tiles % t == 0 && (tiles / t) * heads >= 96) tw = t;
    for (int t = 1; t <= 16 && !tw; ++t)
        if (tiles % t == 0) tw = t;
    KR_CHECK_ARG(tw >= 1, "kr_oproj_heads: N=%d", N);
    if (w_scale)
        return tw > 8 ? launch_oproj_heads<true, 2>(attn_partials, w_packed, w_scale, x_acc, ld_acc, M, N, heads, n_split, tw, s)
                      : launch_oproj_heads<true, 1>(attn_partials, w_packed, w_scale, x_acc, ld_acc, M, N, heads, n_split, tw, s);
    return tw > 8 ? launch_oproj_heads<false, 2>(attn_partials, w_packed, nullptr, x_acc, ld_acc, M, N, heads, n_split, tw, s)
                  : launch_oproj_heads<false, 1>(attn_partials, w_packed, nullptr, x_acc, ld_acc, M, N, heads, n_split, tw, s);
}

extern "C" int kr_linear_decode_narrow_x32(int mode, const kr_bf16* x, int64_t ldx, const float* part_in, int n_part_in,
                                           kr_bf16* x_out, int64_t ldxo, float* x_out_f32, int64_t ldxf, const void* w_packed,
                                           const float* w_scale, const kr_bf16* bias, const kr_bf16* norm_w, float norm_eps,
                                           const kr_bf16* residual, int64_t ldr, kr_bf16* out, float* out_f32, int64_t ldc, int M,
                                           int N, int K, int waves, int ksplit, const float* cs_table, int cs_stride,
                                           const int32_t* prompt_len, const int32_t* ctx_len, kr_bf16* q_out, kr_bf16* kcache,
                                           kr_bf16* vtcache, int heads, int kv_heads, int s_max, const kr_narrow_opts* opts,
                                           const void* pf_ptr, size_t pf_bytes, int pf_blocks, kr_stream s) {
    KR_CHECK_ARG(!x_out_f32 || (norm_w && ldxf >= K && (ldxf & 3) == 0), "kr_linear_decode_narrow_x32: x_out_f32 needs the norm prologue");
    return narrow_impl(mode, x, ldx, part_in, n_part_in, x_out, ldxo, w_packed, w_scale, bias, norm_w, norm_eps, residual, ldr, out,
                       out_f32, ldc, M, N, K, waves, ksplit, cs_table, cs_stride, prompt_len, ctx_len, q_out, kcache, vtcache, heads,
                       kv_heads, s_max, opts, s, x_out_f32, ldxf, pf_ptr, pf_bytes, pf_blocks);
}

extern "C" int kr_linear_decode_wide_x32(int mode, const float* x_f32, int64_t ldx, kr_bf16* x_out, int64_t ldxo, const void* w_packed,
                                         const float* w_scale, const kr_bf16* norm_w, float norm_eps, kr_bf16* out, float* out_f32,
                                         int64_t ldc, int M, int N, int K, int blocks, int waves, float* amax_val, int32_t* amax_idx,
                                         kr_stream s) {
    KR_CHECK_ARG(x_f32 && (ldx & 3) == 0 && (!x_out || (ldxo >= K && (ldxo & 7) == 0)), "kr_linear_decode_wide_x32: x");
    return wide_impl(mode, reinterpret_cast<const kr_bf16*>(x_f32), ldx, w_packed, w_scale, nullptr, norm_w, norm_eps, nullptr, 0, out,
                     out_f32, ldc, M, N, K, blocks, waves, amax_val, amax_idx, s, true, x_out, ldxo);
}
#endif  // KR_EXPERIMENTS
