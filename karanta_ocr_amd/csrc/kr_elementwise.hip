// HBM-bound row kernels: cast+pad, LayerNorm, RMSNorm, embedding gather/scatter, argmax.
// All of them move 16 B per lane per access (8 bf16) and keep a row in registers.
#include "kr_common.h"

// ---------------------------------------------------------------- cast + pad (pixel_values fp32 -> bf16)
__global__ void __launch_bounds__(256) cast_pad_kernel(const float* __restrict__ src, kr_bf16* __restrict__ dst,
                                                       int64_t rows, int k, int k_pad) {
    const int chunks = k_pad >> 3;
    const int64_t total = rows * chunks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / chunks;
        const int c = (int)(i - r * chunks) << 3;
        bf16x8 o;
        const float* p = src + r * k + c;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = f2bf(c + j < k ? p[j] : 0.0f);
        st8(dst + r * k_pad + c, o);
    }
}

extern "C" int kr_cast_pad_f32_bf16(const float* src, kr_bf16* dst, int64_t rows, int k, int k_pad, kr_stream s) {
    KR_CHECK_ARG(src && dst && rows >= 0 && k > 0 && k_pad >= k && (k_pad & 7) == 0, "kr_cast_pad_f32_bf16: bad args");
    if (rows == 0) return KR_OK;
    const int64_t total = rows * (k_pad >> 3);
    int grid = (int)((total + 255) / 256);
    if (grid > 8192) grid = 8192;
    cast_pad_kernel<<<grid, 256, 0, kr_hs(s)>>>(src, dst, rows, k, k_pad);
    KR_CHECK_LAUNCH();
    return KR_OK;
}

// ---------------------------------------------------------------- LayerNorm / RMSNorm: one wave per row
template <int MAXC, bool RMS>
__global__ void __launch_bounds__(256) norm_kernel(const kr_bf16* __restrict__ x, int64_t ldx,
                                                   const kr_bf16* __restrict__ w, const kr_bf16* __restrict__ b,
                                                   kr_bf16* __restrict__ y, int64_t rows, int d, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int chunks = d >> 3;
    const kr_bf16* xr = x + row * ldx;
    float v[MAXC][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < chunks) {
            bf16x8 t = ld8(xr + c * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[i][j] = bf2f(t[j]);
                sum += RMS ? v[i][j] * v[i][j] : v[i][j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
        }
    }
    sum = wave_sum(sum);
    float mean = 0.f, rstd;
    if (RMS) {
        rstd = rsqrtf(sum / (float)d + eps);
    } else {
        mean = sum / (float)d;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + i * 64;
            if (c < chunks) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float t = v[i][j] - mean;
                    sq += t * t;
                }
            }
        }
        sq = wave_sum(sq);
        rstd = rsqrtf(sq / (float)d + eps);
    }
    kr_bf16* yr = y + row * (int64_t)d;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + i * 64;
        if (c < chunks) {
            bf16x8 wv = ld8(w + c * 8);
            bf16x8 o;
            if (RMS) {
                // Qwen2VLRMSNorm: the normalised value is cast to the activation dtype BEFORE the
                // multiply by the weight (TF:modeling_qwen2_vl.py:105-110).
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = f2bf(bf2f(wv[j]) * bfround(v[i][j] * rstd));
            } else {
                bf16x8 bv = ld8(b + c * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = f2bf((v[i][j] - mean) * rstd * bf2f(wv[j]) + bf2f(bv[j]));
            }
            st8(yr + c * 8, o);
        }
    }
}

template <bool RMS>
static int launch_norm(const kr_bf16* x, int64_t ldx, const kr_bf16* w, const kr_bf16* b, kr_bf16* y, int64_t rows,
                       int d, float eps, kr_stream s) {
    if (rows == 0) return KR_OK;
    const unsigned grid = (unsigned)((rows + 3) / 4);
    const int chunks = d >> 3;
    if (chunks <= 64 * 2)
        norm_kernel<2, RMS><<<grid, 256, 0, kr_hs(s)>>>(x, ldx, w, b, y, rows, d, eps);
    else if (chunks <= 64 * 4)
        norm_kernel<4, RMS><<<grid, 256, 0, kr_hs(s)>>>(x, ldx, w, b, y, rows, d, eps);
    else
        norm_kernel<8, RMS><<<grid, 256, 0, kr_hs(s)>>>(x, ldx, w, b, y, rows, d, eps);
    KR_CHECK_LAUNCH();
    return KR_OK;
}

extern "C" int kr_layernorm(const kr_bf16* x, const kr_bf16* w, const kr_bf16* b, kr_bf16* y, int64_t rows, int d,
                            float eps, kr_stream s) {
    KR_CHECK_ARG(x && w && b && y && rows >= 0 && d > 0 && (d & 7) == 0 && d <= 4096, "kr_layernorm: d=%d unsupported", d);
    return launch_norm<false>(x, d, w, b, y, rows, d, eps, s);
}

extern "C" int kr_rmsnorm(const kr_bf16* x, int64_t ldx, const kr_bf16* w, kr_bf16* y, int64_t rows, int d, float eps,
                          kr_stream s) {
    KR_CHECK_ARG(x && w && y && rows >= 0 && d > 0 && (d & 7) == 0 && d <= 4096 && ldx >= d && (ldx & 7) == 0,
                 "kr_rmsnorm: d=%d ldx=%lld unsupported", d, (long long)ldx);
    return launch_norm<true>(x, ldx, w, nullptr, y, rows, d, eps, s);
}

// ---------------------------------------------------------------- embedding gather + image-embed scatter
__global__ void __launch_bounds__(256) embed_scatter_kernel(const int32_t* __restrict__ src, const kr_bf16* __restrict__ table,
                                                            const kr_bf16* __restrict__ img, kr_bf16* __restrict__ out,
                                                            int64_t rows, int d) {
    const int chunks = d >> 3;
    const int64_t total = rows * chunks;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / chunks;
        const int c = (int)(i - r * chunks) << 3;
        const int32_t sidx = src[r];
        const kr_bf16* p = sidx >= 0 ? table + (int64_t)sidx * d : img + (int64_t)(-sidx - 1) * d;
        st8(out + r * d + c, ld8(p + c));
    }
}

extern "C" int kr_embed_scatter(const int32_t* src, const kr_bf16* table, const kr_bf16* image_embeds, kr_bf16* out,
                                int64_t rows, int d, kr_stream s) {
    KR_CHECK_ARG(src && table && out && rows >= 0 && d > 0 && (d & 7) == 0, "kr_embed_scatter: bad args");
    if (rows == 0) return KR_OK;
    const int64_t total = rows * (d >> 3);
    int grid = (int)((total + 255) / 256);
    if (grid > 8192) grid = 8192;
    embed_scatter_kernel<<<grid, 256, 0, kr_hs(s)>>>(src, table, image_embeds, out, rows, d);
    KR_CHECK_LAUNCH();
    return KR_OK;
}

// ---------------------------------------------------------------- argmax (+ next-token embedding gather)
// One 1024-thread block per sequence.  Ties resolve to the lowest index (torch / numpy argmax).
__device__ __forceinline__ void argmax_block(const float* __restrict__ row, int vocab, float& best, int& best_i) {
    __shared__ float s_v[16];
    __shared__ int s_i[16];
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    // 4 floats per lane per step (vocab rows are 16-byte aligned when vocab % 4 == 0)
    for (int i = threadIdx.x; i < vocab; i += blockDim.x) {
        const float v = row[i];
        if (v > bv || (v == bv && i < bi)) {
            bv = v;
            bi = i;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) {
            bv = ov;
            bi = oi;
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        s_v[wave] = bv;
        s_i[wave] = bi;
    }
    __syncthreads();
    if (wave == 0) {
        const int nw = blockDim.x >> 6;
        bv = lane < nw ? s_v[lane] : -INFINITY;
        bi = lane < nw ? s_i[lane] : 0x7fffffff;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
            }
        }
        if (lane == 0) {
            s_v[0] = bv;
            s_i[0] = bi;
        }
    }
    __syncthreads();
    best = s_v[0];
    best_i = s_i[0];
}

__global__ void __launch_bounds__(1024) argmax_kernel(const float* __restrict__ logits, int64_t ld, int vocab,
                                                      int32_t* __restrict__ out) {
    float bv;
    int bi;
    argmax_block(logits + (int64_t)blockIdx.x * ld, vocab, bv, bi);
    if (threadIdx.x == 0) out[blockIdx.x] = bi;
}

extern "C" int kr_argmax(const float* logits, int64_t ld_logits, int vocab, int32_t* out, int batch, kr_stream s) {
    KR_CHECK_ARG(logits && out && vocab > 0 && batch >= 0 && ld_logits >= vocab, "kr_argmax: bad args");
    if (batch == 0) return KR_OK;
    argmax_kernel<<<batch, 1024, 0, kr_hs(s)>>>(logits, ld_logits, vocab, out);
    KR_CHECK_LAUNCH();
    return KR_OK;
}

__global__ void __launch_bounds__(1024) argmax_embed_kernel(const float* __restrict__ logits, int64_t ld, int vocab,
                                                            const kr_bf16* __restrict__ table, int d,
                                                            int32_t* __restrict__ tokens_out, int32_t* __restrict__ history,
                                                            int32_t* __restrict__ step_ptr, int32_t* __restrict__ ctx_len,
                                                            int32_t* __restrict__ finished, const int32_t* __restrict__ eos,
                                                            int n_eos, int pad_id, int ignore_eos,
                                                            kr_bf16* __restrict__ x_next, int hist_stride) {
    const int b = blockIdx.x;
    float bv;
    int tok;
    argmax_block(logits + (int64_t)b * ld, vocab, bv, tok);
    const int was_finished = finished[b];
    if (was_finished && !ignore_eos) tok = pad_id;
    // every thread sees the same step (read before block 0's thread 0 may bump it: the bump happens
    // after a grid-independent point only in block 0, so read it into a register first)
    const int step = step_ptr[0];
    __syncthreads();
    if (threadIdx.x == 0) {
        tokens_out[b] = tok;
        history[(int64_t)step * hist_stride + b] = tok;
        ctx_len[b] += 1;
        if (!ignore_eos && !was_finished) {
            int hit = 0;
            for (int i = 0; i < n_eos; ++i) hit |= (tok == eos[i]);
            if (hit) finished[b] = 1;
        }
    }
    const int chunks = d >> 3;
    for (int c = threadIdx.x; c < chunks; c += blockDim.x) st8(x_next + (int64_t)b * d + c * 8, ld8(table + (int64_t)tok * d + c * 8));
}

// step_ptr is advanced by a separate 1-thread kernel so that every block of argmax_embed_kernel
// reads the same value regardless of block scheduling order.
__global__ void bump_kernel(int32_t* p) { p[0] += 1; }

extern "C" int kr_argmax_embed(const float* logits, int64_t ld_logits, int vocab, const kr_bf16* embed_table, int d,
                               int32_t* tokens_out, int32_t* history, int32_t* step_ptr, int32_t* ctx_len,
                               int32_t* finished, const int32_t* eos, int n_eos, int pad_id, int ignore_eos,
                               kr_bf16* x_next, int batch, int hist_stride, kr_stream s) {
    KR_CHECK_ARG(logits && embed_table && tokens_out && history && step_ptr && ctx_len && finished && x_next &&
                     vocab > 0 && batch > 0 && hist_stride >= batch && (d & 7) == 0 && ld_logits >= vocab &&
                     (n_eos == 0 || eos),
                 "kr_argmax_embed: bad args");
    argmax_embed_kernel<<<batch, 1024, 0, kr_hs(s)>>>(logits, ld_logits, vocab, embed_table, d, tokens_out, history,
                                                       step_ptr, ctx_len, finished, eos, n_eos, pad_id, ignore_eos,
                                                       x_next, hist_stride);
    KR_CHECK_LAUNCH();
    bump_kernel<<<1, 1, 0, kr_hs(s)>>>(step_ptr);
    KR_CHECK_LAUNCH();
    return KR_OK;
}
