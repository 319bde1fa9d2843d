// Guided decoding and log-probabilities on the device.
//
// The reference sends `guided_regex` (karanta/pipeline.py:304-307) and `response_format` json_schema
// (karanta/data/utils.py:322-440 via bulk_processing/workers/vllm_client.py:196) to vLLM, which masks the logits of
// every step with an automaton over the vocabulary, and asks for `logprobs` / `top_logprobs`
// (karanta/data/create_batch_data_prompts.py:117-118).  Here the host compiles the pattern to a byte DFA
// (karanta_ocr_amd/guided.py) and everything per token stays on the GPU, inside the replayed decode graph:
//   guide_build_masks_kernel   once per pattern: allowed-token bits of every DFA state      (HBM: S x V/8 bytes written)
//   gumbel_argmax_kernel       (kr_decode.hip) skips the tokens whose bit is clear in the row of the slot's state
//   guide_advance_kernel       after the sampler: state <- walk(state, bytes(token))
//   logprob_partial_kernel     per vocabulary slice (LDS resident): max, sum exp, top-k by k rounds of block argmax
//   logprob_merge_kernel       per sequence: log-sum-exp, log-prob of the sampled token, merged top-k -> history
// All of it is integer / selection work except the log-sum-exp (fp32, expf / logf).
#include "kr_common.h"

namespace {

__device__ __forceinline__ void better_lp(float& bv, int& bi, float v, int i) {
    if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
}

// one thread per (state, token): grid (mask_words * 32 / 256, n_states)
__global__ void __launch_bounds__(256) guide_build_masks_kernel(const uint16_t* __restrict__ trans,
                                                                const uint8_t* __restrict__ accept,
                                                                const int32_t* __restrict__ vocab_off,
                                                                const uint8_t* __restrict__ vocab_bytes, int vocab,
                                                                const int32_t* __restrict__ eos, int n_eos,
                                                                uint32_t* __restrict__ masks, int mask_words) {
    const int s = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    bool ok = false;
    if (i < vocab && s != 0) {
        const int o0 = vocab_off[i], o1 = vocab_off[i + 1];
        unsigned st = (unsigned)s;
        for (int o = o0; o < o1 && st != 0; ++o) st = trans[st * 256u + vocab_bytes[o]];
        ok = o1 > o0 && st != 0;              // tokens without bytes (specials) are never allowed ...
        for (int e = 0; e < n_eos; ++e)
            if (i == eos[e]) ok = accept[s] != 0;   // ... except EOS, exactly in the accepting states
    }
    const unsigned long long bal = __ballot(ok);
    const int w = i >> 5;                     // lane 0: i is a multiple of 64, mask_words is even
    if (lane == 0 && w < mask_words) {
        masks[(int64_t)s * mask_words + w] = (uint32_t)bal;
        masks[(int64_t)s * mask_words + w + 1] = (uint32_t)(bal >> 32);
    }
}

__global__ void __launch_bounds__(64) guide_advance_kernel(const int32_t* __restrict__ tokens,
                                                           const int32_t* __restrict__ finished,
                                                           const uint64_t* __restrict__ guide_trans,
                                                           int32_t* __restrict__ guide_state,
                                                           const int32_t* __restrict__ vocab_off,
                                                           const uint8_t* __restrict__ vocab_bytes, int vocab, int batch) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= batch) return;
    const uint64_t tp = guide_trans[b];
    if (tp == 0 || finished[b]) return;
    const uint16_t* trans = reinterpret_cast<const uint16_t*>(tp);
    const int tok = tokens[b];
    if (tok < 0 || tok >= vocab) return;
    unsigned st = (unsigned)guide_state[b];
    for (int o = vocab_off[tok], o1 = vocab_off[tok + 1]; o < o1 && st != 0; ++o) st = trans[st * 256u + vocab_bytes[o]];
    guide_state[b] = (int)st;
}

// ---------------------------------------------------------------- log-probabilities
constexpr int LP_MAX_K = 20;       // OpenAI's top_logprobs bound
constexpr int LP_SLICE = 4096;     // floats of one vocabulary slice held in LDS

__device__ __forceinline__ void block_argmax(float& bv, int& bi, float* s_v, int* s_i) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        better_lp(bv, bi, ov, oi);
    }
    __syncthreads();                 // s_v / s_i of the previous round have been read by everyone
    if (lane == 0) { s_v[wave] = bv; s_i[wave] = bi; }
    __syncthreads();
    bv = s_v[0]; bi = s_i[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) better_lp(bv, bi, s_v[w], s_i[w]);
}

// grid (n_part, batch); part_val/part_idx [batch][n_part][k], part_ms [batch][n_part][2] = (max, sum exp(v - max))
__global__ void __launch_bounds__(256) logprob_partial_kernel(const float* __restrict__ logits, int64_t ld, int vocab, int k,
                                                              float* __restrict__ part_val, int32_t* __restrict__ part_idx,
                                                              float* __restrict__ part_ms) {
    __shared__ float sl[LP_SLICE];
    __shared__ float s_v[4];
    __shared__ int s_i[4];
    const int p = blockIdx.x, n_part = gridDim.x, b = blockIdx.y, tid = threadIdx.x;
    const int per = (vocab + n_part - 1) / n_part;          // <= LP_SLICE (checked by the launcher)
    const int i0 = p * per, n = max(0, min(vocab, i0 + per) - i0);
    const float* row = logits + (int64_t)b * ld + i0;
    float mx = -INFINITY;
    for (int i = tid; i < n; i += 256) { const float v = row[i]; sl[i] = v; mx = fmaxf(mx, v); }
    int dummy = 0;
    block_argmax(mx, dummy, s_v, s_i);                       // (value only; the index is unused)
    float se = 0.f;
    for (int i = tid; i < n; i += 256) se += expf(sl[i] - mx);
    se = wave_sum(se);
    __shared__ float s_se[4];
    if ((tid & 63) == 0) s_se[tid >> 6] = se;
    __syncthreads();
    if (tid == 0) {
        part_ms[((int64_t)b * n_part + p) * 2] = mx;
        part_ms[((int64_t)b * n_part + p) * 2 + 1] = n > 0 ? s_se[0] + s_se[1] + s_se[2] + s_se[3] : 0.f;
    }
    for (int r = 0; r < k; ++r) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = tid; i < n; i += 256) {
            const float v = sl[i];
            if (v != -INFINITY) better_lp(bv, bi, v, i);     // retired elements never come back
        }
        block_argmax(bv, bi, s_v, s_i);
        if (tid == 0) {
            part_val[((int64_t)b * n_part + p) * k + r] = bv;
            part_idx[((int64_t)b * n_part + p) * k + r] = bi == 0x7fffffff ? 0x7fffffff : i0 + bi;
        }
        if (bi != 0x7fffffff && tid == (bi & 255)) sl[bi] = -INFINITY;   // the owner of that element retires it
    }
}

// one workgroup per sequence.  out_lp [hist][batch][1 + k]: log-prob of the sampled token, then of the top-k;
// out_idx [hist][batch][k]: their token ids.  hist index = ctx_len - prompt_len (the sampler has advanced ctx_len:
// same index as the token history).  Sequences that are finished (EOS seen, or frozen) record nothing.
__global__ void __launch_bounds__(256) logprob_merge_kernel(const float* __restrict__ part_val,
                                                            const int32_t* __restrict__ part_idx,
                                                            const float* __restrict__ part_ms, int n_part, int k,
                                                            const float* __restrict__ logits, int64_t ld, int vocab,
                                                            const int32_t* __restrict__ tokens,
                                                            const int32_t* __restrict__ ctx_len,
                                                            const int32_t* __restrict__ prompt_len,
                                                            const int32_t* __restrict__ finished,
                                                            float* __restrict__ out_lp, int32_t* __restrict__ out_idx,
                                                            int hist_len, int hist_batch, int k_stride) {
    extern __shared__ float cand[];                          // n_part * k values, then as many indices
    __shared__ float s_v[4];
    __shared__ int s_i[4];
    const int b = blockIdx.x, tid = threadIdx.x, nc = n_part * k;
    int* cidx = reinterpret_cast<int*>(cand + nc);
    if (finished[b]) return;                                 // uniform per workgroup
    const int h = ctx_len[b] - prompt_len[b];
    if (h < 0 || h >= hist_len) return;
    for (int i = tid; i < nc; i += 256) { cand[i] = part_val[(int64_t)b * nc + i]; cidx[i] = part_idx[(int64_t)b * nc + i]; }
    float mx = -INFINITY;
    for (int i = tid; i < n_part; i += 256) mx = fmaxf(mx, part_ms[((int64_t)b * n_part + i) * 2]);
    int dummy = 0;
    block_argmax(mx, dummy, s_v, s_i);
    float se = 0.f;
    for (int i = tid; i < n_part; i += 256) {
        const float m = part_ms[((int64_t)b * n_part + i) * 2], sv = part_ms[((int64_t)b * n_part + i) * 2 + 1];
        if (sv > 0.f) se += sv * expf(m - mx);
    }
    se = wave_sum(se);
    __shared__ float s_se[4];
    if ((tid & 63) == 0) s_se[tid >> 6] = se;
    __syncthreads();
    const float lse = mx + logf(s_se[0] + s_se[1] + s_se[2] + s_se[3]);
    float* lp = out_lp + ((int64_t)h * hist_batch + b) * (1 + k_stride);
    int32_t* ix = out_idx + ((int64_t)h * hist_batch + b) * k_stride;
    if (tid == 0) {
        const int tok = tokens[b];
        lp[0] = (tok >= 0 && tok < vocab) ? logits[(int64_t)b * ld + tok] - lse : -INFINITY;
    }
    for (int r = 0; r < k; ++r) {
        float bv = -INFINITY;
        int bi = 0x7fffffff, bslot = -1;
        for (int i = tid; i < nc; i += 256) {
            const float v = cand[i];
            const int id = cidx[i];
            if (v != -INFINITY && (v > bv || (v == bv && id < bi))) { bv = v; bi = id; bslot = i; }
        }
        // block argmax on (value, token id); the winning candidate slot is found again by its unique token id
        block_argmax(bv, bi, s_v, s_i);
        if (tid == 0) { lp[1 + r] = bv - lse; ix[r] = bi; }
        if (bslot >= 0 && cidx[bslot] == bi && cand[bslot] == bv) cand[bslot] = -INFINITY;
        __syncthreads();
    }
}

}  // namespace

extern "C" int kr_guide_build_masks(const uint16_t* trans, const uint8_t* accept, int n_states, const int32_t* vocab_off,
                                    const uint8_t* vocab_bytes, int vocab, const int32_t* eos, int n_eos, uint32_t* masks,
                                    int mask_words, kr_stream s) {
    KR_CHECK_ARG(trans && accept && vocab_off && vocab_bytes && masks && (eos || n_eos == 0), "kr_guide_build_masks: null pointer");
    KR_CHECK_ARG(n_states > 0 && n_states <= 65535 && vocab > 0 && n_eos >= 0, "kr_guide_build_masks: bad sizes");
    KR_CHECK_ARG(mask_words % 2 == 0 && (int64_t)mask_words * 32 >= vocab, "kr_guide_build_masks: mask_words must be even and cover the vocabulary");
    const int gx = (mask_words * 32 + 255) / 256;
    guide_build_masks_kernel<<<dim3(gx, n_states), 256, 0, kr_hs(s)>>>(trans, accept, vocab_off, vocab_bytes, vocab, eos, n_eos, masks,
                                                                       mask_words);
    KR_CHECK_LAUNCH();
    return KR_OK;
}

extern "C" int kr_guide_advance(const int32_t* tokens, const int32_t* finished, const uint64_t* guide_trans, int32_t* guide_state,
                                const int32_t* vocab_off, const uint8_t* vocab_bytes, int vocab, int batch, kr_stream s) {
    KR_CHECK_ARG(tokens && finished && guide_trans && guide_state && vocab_off && vocab_bytes, "kr_guide_advance: null pointer");
    KR_CHECK_ARG(vocab > 0 && batch > 0, "kr_guide_advance: bad sizes");
    guide_advance_kernel<<<(batch + 63) / 64, 64, 0, kr_hs(s)>>>(tokens, finished, guide_trans, guide_state, vocab_off, vocab_bytes, vocab,
                                                                  batch);
    KR_CHECK_LAUNCH();
    return KR_OK;
}

extern "C" int kr_logprobs_topk(const float* logits, int64_t ld_logits, int vocab, int k, int n_part, float* part_val,
                                int32_t* part_idx, float* part_ms, const int32_t* tokens, const int32_t* ctx_len,
                                const int32_t* prompt_len, const int32_t* finished, float* out_lp, int32_t* out_idx,
                                int hist_len, int hist_batch, int k_stride, int batch, kr_stream s) {
    KR_CHECK_ARG(logits && part_val && part_idx && part_ms && tokens && ctx_len && prompt_len && finished && out_lp && out_idx,
                 "kr_logprobs_topk: null pointer");
    KR_CHECK_ARG(vocab > 0 && ld_logits >= vocab && batch > 0 && batch <= hist_batch && hist_len > 0, "kr_logprobs_topk: bad sizes");
    KR_CHECK_ARG(k >= 0 && k <= LP_MAX_K && k <= k_stride, "kr_logprobs_topk: k must be in 0..%d and <= k_stride", LP_MAX_K);
    KR_CHECK_ARG(n_part > 0 && n_part <= 1024 && (vocab + n_part - 1) / n_part <= LP_SLICE,
                 "kr_logprobs_topk: a vocabulary slice (vocab / n_part) must fit %d floats", LP_SLICE);
    logprob_partial_kernel<<<dim3(n_part, batch), 256, 0, kr_hs(s)>>>(logits, ld_logits, vocab, k, part_val, part_idx, part_ms);
    KR_CHECK_LAUNCH();
    const size_t lds = (size_t)n_part * (k > 0 ? k : 1) * 8;
    KR_CHECK_ARG(lds <= 160 * 1024 - 1024, "kr_logprobs_topk: n_part * k candidates exceed the LDS");
    logprob_merge_kernel<<<batch, 256, lds, kr_hs(s)>>>(part_val, part_idx, part_ms, n_part, k, logits, ld_logits, vocab, tokens, ctx_len,
                                                        prompt_len, finished, out_lp, out_idx, hist_len, hist_batch, k_stride);
    KR_CHECK_LAUNCH();
    return KR_OK;
}
