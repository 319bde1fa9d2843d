/*
 * karanta_hip.h — C-ABI of libkaranta_hip.so, the MI355X (gfx950) kernel library behind the
 * karanta-ocr VLM inference hot path.
 *
 * What it replaces.  The reference has NO in-process kernel interface: the Qwen2-VL
 * image-encoder + text-decoder forward runs in an external `vllm serve` process
 * (reference karanta/pipeline.py:707-742 spawn, :317-319 HTTP POST;
 * bulk_processing/workers/vllm_client.py:209 POST) or inside Hugging Face `model.generate`
 * (karanta/training/test_trained_model.py:76-99).  The entry points below are therefore the
 * operator set those third-party forwards execute, one C function per
 * operator, each citing the Hugging Face symbol whose arithmetic it reproduces
 * ("TF:" = transformers/models/qwen2_vl/modeling_qwen2_vl.py, line numbers as in SURVEY.md §8 a-ii).
 * INTEGRATION.md shows the ctypes binding a reference maintainer adds.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes only.  Device pointers are raw HIP device addresses
 *    (e.g. torch.Tensor.data_ptr()); the caller owns every buffer.
 *  - Every launch takes `kr_stream` (a hipStream_t cast to void*; NULL = default stream) and is
 *    asynchronous on it.  No function synchronises unless its name says so.
 *  - Return value: 0 = ok, <0 = error; kr_last_error() gives a thread-local message.
 *    No exceptions cross the ABI.  No global mutable state besides the error string.
 *  - "bf16" buffers are raw uint16 bit patterns (round-to-nearest-even); fp32 accumulate inside
 *    every kernel.  Row-major unless stated.
 *  - Shapes that a kernel cannot serve are rejected on the host with an error (never launched).
 */
#ifndef KARANTA_HIP_H
#define KARANTA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* kr_stream; /* hipStream_t */
typedef uint16_t kr_bf16;

/* kr_version() of the library this header describes: major * 100 + minor.  The major changes with every incompatible
 * change of a signature or struct below (r4: kr_narrow_opts argument of round 3, packed 17..32-row family); a caller built
 * against major X must refuse a library whose kr_version() / 100 != X. */
#define KR_ABI_VERSION 402

#define KR_OK 0
#define KR_ERR_ARG (-1)    /* unsupported shape / null pointer */
#define KR_ERR_HIP (-2)    /* HIP runtime error */
#define KR_ERR_RCCL (-3)   /* RCCL error */
#define KR_ERR_STATE (-4)  /* wrong call order (graph capture etc.) */

/* GEMM / GEMV epilogues (applied to the fp32 accumulator, then one rounding to the output type) */
#define KR_EPI_NONE 0
#define KR_EPI_QUICK_GELU 1 /* x*sigmoid(1.702x): ViT fc1, TF:293-301 */
#define KR_EPI_GELU_ERF 2   /* exact GELU: PatchMerger, TF:277-290 */
#define KR_EPI_SILU_MUL 3   /* silu(gate)*up with gate/up rows interleaved in 16-row groups: Qwen2MLP, TF:453-466 */
#define KR_EPI_SILU_MUL8 4  /* same, interleaved in 8-row groups (g0..g7,u0..u7,g8..): one 16-row MFMA tile = 8 features;
                             * kr_gemm_bf16 takes an optional bias here, interleaved like the rows (the biased SwiGLU
                             * MLP of the Qwen2.5-VL vision blocks, TF25:85-97) */

/* ------------------------------------------------------------------ library / device */
int kr_version(void);
const char* kr_last_error(void);
/* fills name (<=63 chars), CU count, total memory bytes of `device`. */
int kr_device_info(int device, char* name64, int* compute_units, size_t* total_mem);
int kr_set_device(int device);
int kr_stream_synchronize(kr_stream s);
/* A stream restricted to `cus_enabled` of the device's compute units (hipExtStreamCreateWithCUMask; the first cus_enabled
 * bits of the mask: on an 8-XCD part cus_enabled / 8 CUs of every XCD).  The continuous-batching server runs its admissions
 * (ViT + prefill: MFMA-bound, long-running workgroups) on such a stream beside the decode graph (HBM-bound, short
 * launches) on the main one — the reference gets the same overlap from vLLM's chunked prefill.  kr_stream_destroy frees it. */
int kr_stream_create_cu_mask(kr_stream* out, int cus_enabled);
/* The same with the mask bits [first_cu, first_cu + n_cus): a second stream on the COMPLEMENT of a kr_stream_create_cu_mask stream shares no
 * compute unit with it (the decode graph beside an admission: neither launch waits for a workgroup slot of the other). */
int kr_stream_create_cu_range(kr_stream* out, int first_cu, int n_cus);
int kr_stream_destroy(kr_stream s);

/* ------------------------------------------------------------------ profiling events
 * HIP events on the caller's stream, used by bench.py to time kernels on the stream they are
 * launched on (torch.cuda.Event only sees torch's current stream). */
int kr_event_create(void** ev);
int kr_event_destroy(void* ev);
int kr_event_record(void* ev, kr_stream s);
/* hipStreamWaitEvent: inside a stream capture this forks / joins the graph (a second stream pulled in by an event of
 * the capturing stream becomes a parallel branch; an event of that stream waited on by the capturing one is the join). */
int kr_stream_wait_event(kr_stream s, void* ev);
int kr_event_synchronize(void* ev);
int kr_event_elapsed_ms(void* start, void* stop, float* ms);

/* ------------------------------------------------------------------ HIP graphs (decode step replay) */
int kr_graph_begin_capture(kr_stream s);
int kr_graph_end_capture(kr_stream s, void** graph_exec);
int kr_graph_launch(void* graph_exec, kr_stream s);
int kr_graph_destroy(void* graph_exec);

/* ------------------------------------------------------------------ elementwise / norms */

/* pixel_values fp32 [rows, k] -> bf16 [rows, k_pad], zero padded (k_pad % 8 == 0).
 * Front of PatchEmbed (TF:251-274): `hidden_states.to(dtype=target_dtype)`. */
int kr_cast_pad_f32_bf16(const float* src, kr_bf16* dst, int64_t rows, int k, int k_pad, kr_stream s);

/* nn.LayerNorm with bias, fp32 statistics (ViT blocks TF:425-449, merger ln_q TF:277-290).
 * x,y bf16 [rows, d]; w,b bf16 [d]; d % 8 == 0, d <= 8192. */
int kr_layernorm(const kr_bf16* x, const kr_bf16* w, const kr_bf16* b, kr_bf16* y,
                 int64_t rows, int d, float eps, kr_stream s);

/* Qwen2VLRMSNorm (TF:96-110): y = w * bf16(x * rsqrt(mean(x^2)+eps)).
 * x has row stride ldx (elements); y is dense [rows, d]. */
int kr_rmsnorm(const kr_bf16* x, int64_t ldx, const kr_bf16* w, kr_bf16* y,
               int64_t rows, int d, float eps, kr_stream s);

/* ------------------------------------------------------------------ dense linears */

/* C[M,N] = epi(A[M,K] * W[N,K]^T + bias[N]) (+ residual[M,N]); nn.Linear with the weight in its
 * native [out,in] layout.  bf16 in, fp32 MFMA accumulate, bf16 out.  K % 64 == 0.
 * bias / residual may be NULL.  With KR_EPI_SILU_MUL, W holds gate/up rows interleaved in groups
 * of 16 (g0..g15,u0..u15,g16..), N counts both, C is [M, N/2] and N % 32 == 0.
 * lda / ldc / ldr are row strides in elements.  w_packed != 0: W is stored in the decode layout
 * [N/16][K/32][4][16][8] (see kr_linear_decode) instead of row-major — one copy of the decoder
 * weights serves prefill and decode.  Used for every ViT Linear, the merger, and every decoder
 * Linear at prefill. */
int kr_gemm_bf16(const kr_bf16* A, int64_t lda, const kr_bf16* W, const kr_bf16* bias,
                 const kr_bf16* residual, int64_t ldr, kr_bf16* C, int64_t ldc,
                 int64_t M, int N, int K, int epilogue, int w_packed, kr_stream s);

/* kr_gemm_bf16 with a CALLER-OWNED split-K scratch (>= KR_GEMM_SCRATCH_BYTES, 16-byte aligned; NULL = kr_gemm_bf16).
 * A launch whose last round of 256x256 tiles is at most half full runs that round as 128x128 quarters; with K >= 4096
 * and a scratch each quarter is also cut along K (partials into the scratch, then a reduce-in-split-order + epilogue
 * launch; prefill down_proj 353 -> 259 us).  The library allocates nothing and keeps nothing between calls: the same
 * call with the same scratch argument gives the same bits whatever ran before it, inside a stream capture or not.  The
 * scratch must not be used by another stream's GEMM at the same time. */
#define KR_GEMM_SCRATCH_BYTES ((size_t)512 * 65536)
int kr_gemm_bf16_ws(const kr_bf16* A, int64_t lda, const kr_bf16* W, const kr_bf16* bias,
                    const kr_bf16* residual, int64_t ldr, kr_bf16* C, int64_t ldc,
                    int64_t M, int N, int K, int epilogue, int w_packed, float* scratch, size_t scratch_bytes, kr_stream s);

/* kr_gemm_bf16 with weight-only fp8 (BASELINE.json config 5: "fp8 weights, bf16 activations"; SURVEY.md §8(b2)
 * kr_gemm_fp8): W as OCP e4m3fn codes in the decode layout [N/16][K/64][4][16][16] (weights.pack_w16x64_fp8) with one
 * f32 scale per output row, C = epi(A (scale * W)^T + bias) (+ residual).  The codes are converted to bf16 on the way
 * from LDS to the MFMA, accumulation is fp32, the scale multiplies the accumulator.  Epilogues KR_EPI_NONE and
 * KR_EPI_SILU_MUL8 (the decoder's prefill linears).  K % 64 == 0, N % 16 == 0, ldc % 8 == 0, C 16-byte aligned. */
int kr_gemm_fp8(const kr_bf16* A, int64_t lda, const uint8_t* w_packed_fp8, const float* w_scale, const kr_bf16* bias,
                const kr_bf16* residual, int64_t ldr, kr_bf16* C, int64_t ldc, int64_t M, int N, int K, int epilogue,
                kr_stream s);

/* W8A8 prefill on the fp8 matrix instruction (BASELINE.json config 5: "Qwen2-VL-7B fp8 weights (CDNA4 fp8 MFMA)"; the
 * reference's default OLMO_7B_0725_FP8, /root/reference/karanta/constants.py:23, is served by vLLM with dynamic per-token
 * activation scales):
 *   kr_quantize_rows_fp8 : x bf16 [rows, K] (row stride ldx elements) -> q e4m3fn codes [rows, K] (row stride ldq BYTES,
 *       a multiple of 16) + scale f32 [rows]: scale = max|row| / 448 (1 for an all-zero row), code = e4m3(x / scale), round
 *       to nearest even (bit-identical to weights.quantize_fp8_rows on the host);
 *   kr_gemm_fp8a : C = epi((a_scale * A8) (w_scale * W8)^T + bias) (+ residual) with both operands as codes through
 *       v_mfma_f32_16x16x32_fp8_fp8 (products of two e4m3 values are exact in f32; f32 accumulation), the two scales applied
 *       to the accumulators.  A8 row-major [M, lda bytes]; weights, epilogues and constraints as kr_gemm_fp8. */
int kr_quantize_rows_fp8(const kr_bf16* x, int64_t ldx, uint8_t* q, int64_t ldq, float* scale, int64_t rows, int K, kr_stream s);
int kr_gemm_fp8a(const uint8_t* A8, int64_t lda, const float* a_scale, const uint8_t* w_packed_fp8, const float* w_scale,
                 const kr_bf16* bias, const kr_bf16* residual, int64_t ldr, kr_bf16* C, int64_t ldc, int64_t M, int N, int K,
                 int epilogue, kr_stream s);

/* Decode-time Linear for M <= 16 rows (one row per live sequence): the weight matrix is streamed
 * exactly once from HBM (this is the HBM-roofline kernel of the decode loop).
 * Same semantics and epilogues as kr_gemm_bf16.  If out_f32 != NULL the result is written there
 * as fp32 [M, ldc] instead of bf16 (lm_head logits, TF:1320-1323).
 * If norm_w != NULL, x is first RMS-normalised (Qwen2VLRMSNorm, eps=norm_eps) on the fly. */
int kr_gemv_bf16(const kr_bf16* x, int64_t ldx, const kr_bf16* W, const kr_bf16* bias,
                 const kr_bf16* residual, int64_t ldr, kr_bf16* out, float* out_f32, int64_t ldc,
                 int M, int N, int K, int epilogue,
                 const kr_bf16* norm_w, float norm_eps, kr_stream s);

/* ------------------------------------------------------------------ ViT attention path */

/* Rotary + re-layout in 64-token blocks, shared by the ViT (apply_rotary_pos_emb_vision,
 * TF:225-236, rotate_half TF:173-177) and the decoder prefill (apply_multimodal_rotary_pos_emb,
 * TF:180-222, with section-interleaved cos/sin precomputed on the host).
 * `qkv` is the fused projection output, row stride ld_qkv, with q / k / v starting at column
 * q_off / k_off / v_off (elements), heads contiguous.  cos,sin fp32 [n, hd] by token row.
 * Block i covers tokens blk_tok0[i] .. +blk_ntok[i] (<= 64) of one segment, starting at a
 * multiple of 64 inside that segment:
 *   q_out  [q_heads][token row][hd]                    (head stride q_head_stride elements)
 *   k_out  [kv_heads][rows][hd]: token j -> row blk_k_row0[i] + j   (head stride k_head_stride)
 *   vt_out [kv_heads][blocks][2][hd][32]: V transposed in 64-token blocks, each block as two contiguous 32-token halves
 *          (element (channel d, token j of the block) at ((j / 32) * hd + d) * 32 + j % 32; round 4: a decode-attention wave reads
 *          a 32-key unit as whole cache lines — rounds 1-3 stored [hd][64]), block blk_vt_blk[i], zero padded
 * For the decoder, k_out / vt_out are the KV cache of one layer. hd in {80, 128}. */
int kr_qkv_prep(const kr_bf16* qkv, int64_t ld_qkv, int q_off, int k_off, int v_off,
                const float* cos, const float* sin,
                const int32_t* blk_tok0, const int32_t* blk_ntok,
                const int64_t* blk_k_row0, const int64_t* blk_vt_blk, int n_blk,
                kr_bf16* q_out, int64_t q_head_stride, kr_bf16* k_out, int64_t k_head_stride,
                kr_bf16* vt_out, int64_t vt_head_stride, int q_heads, int kv_heads, int hd,
                kr_stream s);

/* Plain in-place rotary on a [n, heads, hd] tensor (kept for API parity with SURVEY §8 b2
 * `kr_rope2d_vision`; the engine uses the fused kr_vit_qkv_prep). */
int kr_rope2d_vision(kr_bf16* x, const float* cos, const float* sin, int64_t n, int heads, int hd,
                     int64_t row_stride, kr_stream s);

/* Flash-style attention over variable-length segments, fp32 online softmax, bf16 P (as
 * eager_attention_forward TF:317-339 computes it: softmax in fp32, cast, P*V).
 *   q : [q_heads, nq_total, hd]   k : [kv_heads, *, hd] rows   vt : [kv_heads, *, 2, hd, 32] blocks (kr_qkv_prep's layout)
 *   out : [nq_total, q_heads*hd]
 * Work list: qblk[4*i+0..3] = {q_row0 (global row of first query), n_q_rows (<=128),
 *   k_row0 (global k row of the segment's key 0), vt_block0}; qblk_len[2*i+0..1] =
 *   {kv_len visible to the LAST query of the block when causal / segment length otherwise,
 *    position of the block's first query inside its segment}.
 * kv head = q head / (q_heads / kv_heads).  k_head_stride / vt_head_stride in elements.
 * hd in {80, 128}.  causal=0: ViT full attention per image (VisionAttention TF:342-422, segments
 * from cu_seqlens TF:399-418); causal=1: decoder prefill (Qwen2VLAttention TF:469-556). */
int kr_attn_varlen(const kr_bf16* q, const kr_bf16* k, const kr_bf16* vt, kr_bf16* out,
                   const int32_t* qblk, const int32_t* qblk_len, int n_qblk,
                   int64_t nq_total, int q_heads, int kv_heads, int hd,
                   int64_t k_head_stride, int64_t vt_head_stride, float scale, int causal,
                   kr_stream s);
/* The same with the work list's query-block size stated: 128 (4-wave workgroups, what kr_attn_varlen assumes) or 256
 * (8-wave workgroups: twice the queries per staged K / V^T tile; the list must have been built for that block size). */
int kr_attn_varlen_q(const kr_bf16* q, const kr_bf16* k, const kr_bf16* vt, kr_bf16* out,
                     const int32_t* qblk, const int32_t* qblk_len, int n_qblk,
                     int64_t nq_total, int q_heads, int kv_heads, int hd,
                     int64_t k_head_stride, int64_t vt_head_stride, float scale, int causal,
                     int q_block, kr_stream s);

/* ------------------------------------------------------------------ decoder: embedding, M-RoPE, KV cache */

/* inputs_embeds (TF:1159-1168): row i of out = table[src[i]] if src[i] >= 0, else
 * image_embeds[-src[i]-1] (masked_scatter order precomputed on the host). bf16, d % 8 == 0. */
int kr_embed_scatter(const int32_t* src, const kr_bf16* table, const kr_bf16* image_embeds,
                     kr_bf16* out, int64_t rows, int d, kr_stream s);

/* KV cache of one model:
 *   kcache  : [layers, batch, kv_heads, s_max, hd]
 *   vtcache : [layers, batch, kv_heads, s_max/64, 2, hd, 32]   (V transposed in 64-token blocks of two contiguous 32-token
 *             halves: kr_qkv_prep's layout; every writer and reader of the cache in this library uses it)
 * zero-initialised by the caller (masked keys are multiplied by P = 0, so they must be finite). */

/* SURVEY §8 b2 names kept as thin standalone operators (tests call them directly). */
int kr_mrope(kr_bf16* x, const float* cos, const float* sin, int64_t n, int heads, int hd,
             int64_t row_stride, kr_stream s);
int kr_kv_append(const kr_bf16* k, const kr_bf16* v, int64_t row_stride,
                 const int32_t* tok_seq, const int32_t* tok_pos,
                 kr_bf16* kcache, kr_bf16* vtcache,
                 int64_t n, int kv_heads, int hd, int layer, int batch, int s_max, kr_stream s);

/* Decode step front: for each live sequence b (one new token): M-RoPE at position
 * ctx_len[b] + rope_delta[b] (all three axes equal for text tokens, TF:1124-1136), computed from
 * inv_freq fp32 [hd/2]; writes q_out [batch, heads, hd] and appends K / V^T at cache position
 * ctx_len[b].  ctx_len is NOT modified (kr_argmax_embed advances it once per step). */
int kr_decode_qkv_prep(const kr_bf16* qkv, const float* inv_freq, const int32_t* ctx_len,
                       const int32_t* rope_delta, kr_bf16* q_out, kr_bf16* kcache, kr_bf16* vtcache,
                       int batch, int heads, int kv_heads, int hd, int layer, int s_max, kr_stream s);

/* Decode attention, q_len = 1, GQA, context = ctx_len[b] + 1 tokens (the new token included),
 * split over `n_split` key ranges; partials go to `workspace` (fp32,
 * batch*heads*n_split*(hd+2) floats) and are merged by the same call's second kernel.
 * out : [batch, heads*hd] bf16.  (Qwen2VLAttention with cache, TF:469-556.) */
int kr_attn_decode_gqa(const kr_bf16* q, const kr_bf16* kcache, const kr_bf16* vtcache,
                       const int32_t* ctx_len, kr_bf16* out, float* workspace,
                       int batch, int heads, int kv_heads, int hd, int layer, int s_max,
                       int n_split, float scale, kr_stream s);

/* Greedy sampling (`generate(do_sample=False)`, reference test_trained_model.py:91): argmax of
 * fp32 logits [batch, vocab] (lowest index wins ties), token written to tokens_out[b] and to
 * history[step_ptr[0]*hist_stride + b]; the token's embedding row is gathered into x_next [batch, d];
 * ctx_len[b] += 1; step_ptr[0] += 1 (by a trailing 1-thread kernel).  Sequences with finished[b] != 0 emit pad_id; a
 * sequence becomes finished when it emits one of eos[0..n_eos).  */
int kr_argmax_embed(const float* logits, int64_t ld_logits, int vocab,
                    const kr_bf16* embed_table, int d,
                    int32_t* tokens_out, int32_t* history, int32_t* step_ptr, int32_t* ctx_len,
                    int32_t* finished, const int32_t* eos, int n_eos, int pad_id, int ignore_eos,
                    kr_bf16* x_next, int batch, int hist_stride, kr_stream s);

/* Plain argmax (SURVEY §8 b2 `kr_argmax`). */
int kr_argmax(const float* logits, int64_t ld_logits, int vocab, int32_t* out, int batch, kr_stream s);

/* ------------------------------------------------------------------ fused decode step (5 launches per layer + 2)
 * Decode linears over PACKED weights: W[N][K] stored as [N/16][K/32][4][16][8] bf16 (element
 * (n, k) at ((((n/16)*(K/32) + k/32)*4 + (k%32)/8)*16 + n%16)*8 + k%8): one (16 x 32) block is an
 * MFMA fragment set in lane order, so the matrix is read from HBM as one linear stream, 1 KiB of
 * contiguous memory per wave-level load.  y[M<=16, N] = epi(x W^T).  K is split over the `waves` (4, 8 or 16) waves of a
 * workgroup.  max_blocks > 0 (and ksplit == 1): at most that many PERSISTENT workgroups walk the
 * 16-row tile groups (the prologue runs once per workgroup, the next group's weights are put in
 * flight before the current group's reduction); 0 = one workgroup per tile group.
 * ksplit > 1 additionally splits K over workgroups with a deterministic in-launch
 * reduction (ws: f32 [N/16][ksplit][256] slabs, counters: int32 [N/16], zero-initialised, left
 * zero).  Prologues: Qwen2VLRMSNorm on x (norm_w, K <= 4096), or x = merge of the decode
 * attention partials (attn_partials [M][K/128][attn_split][hd+4] f32 as kr_attn_decode_fused
 * leaves them with out == NULL; x itself may then be NULL).
 * mode 0 PLAIN   : +bias, +residual, bf16 `out` or fp32 `out_f32` [M, ldc]      (o_proj, down_proj)
 *      1 SILU    : silu(gate)*up, rows interleaved in 16-row groups, out [M, N/2] (gate/up)
 *      2 ROPE_KV : fused q/k/v projection (+bias) -> M-RoPE -> q_out [M, heads, 128]; k -> kcache
 *                  row ctx_len[b]; v -> vtcache column ctx_len[b] (cache pointers = this layer's
 *                  base, layout as kr_decode_qkv_prep).  cos/sin come from cs_table
 *                  [M][cs_stride][128] fp32 (cos[0..64), sin[0..64)), entry ctx_len[b] -
 *                  prompt_len[b] = index of the decode position.  head_dim 128 only.
 *                  (TF:469-556, :180-222)
 *      3 ARGMAX  : lm_head: per-workgroup (max, lowest index) partials amax_val/amax_idx
 *                  [M][ceil(N/32)]; fp32 logits also written when out_f32 != NULL   (TF:1320-1323) */
#define KR_DEC_PLAIN 0
#define KR_DEC_SILU 1
#define KR_DEC_ROPE_KV 2
#define KR_DEC_ARGMAX 3
#define KR_DEC_SILU8 4      /* SILU with 8-row interleave: one tile per workgroup -> 2x the workgroups of KR_DEC_SILU */
#define KR_DEC_OUT_XP 0x100 /* kr_linear_decode_wide*, OR-ed into KR_DEC_SILU8 at 17..32 rows: `out` is written in the packed
                             * activation layout of kr_pack_rows32 (ldc ignored) — the input format of kr_linear_decode32 */
int kr_linear_decode(int mode, const kr_bf16* x, int64_t ldx, const kr_bf16* w_packed, const kr_bf16* bias,
                     const kr_bf16* norm_w, float norm_eps, const kr_bf16* residual, int64_t ldr,
                     kr_bf16* out, float* out_f32, int64_t ldc, int M, int N, int K, int waves, int max_blocks, int ksplit,
                     float* ws, int32_t* counters, const float* attn_partials, int attn_split,
                     const float* cs_table, int cs_stride, const int32_t* prompt_len, const int32_t* ctx_len,
                     kr_bf16* q_out, kr_bf16* kcache, kr_bf16* vtcache, int heads, int kv_heads, int s_max,
                     float* amax_val, int32_t* amax_idx, kr_stream s);

/* Wide decode linears (gate/up, lm_head): one WAVE per 16-row weight tile over the full K — no cross-wave
 * reduction and no barrier after the prologue; an 8-deep register ring streams on across tile boundaries
 * (K % 512 == 0, K <= 4096).  `blocks` workgroups of `waves` (<= 8) waves; wave (b, w) walks tiles
 * b + blocks*(w + waves*i).  Modes PLAIN / SILU8 / ARGMAX as kr_linear_decode; ARGMAX partials are per
 * wave: amax_val / amax_idx [M][blocks*waves], slot b + blocks*w (value -inf, index INT_MAX for a wave
 * without tiles). */
int kr_linear_decode_wide(int mode, const kr_bf16* x, int64_t ldx, const kr_bf16* w_packed, const kr_bf16* bias,
                          const kr_bf16* norm_w, float norm_eps, const kr_bf16* residual, int64_t ldr,
                          kr_bf16* out, float* out_f32, int64_t ldc, int M, int N, int K, int blocks, int waves,
                          float* amax_val, int32_t* amax_idx, kr_stream s);

/* Options of one narrow launch (NULL = all zero).  They replace round 2's thread-local one-shot setters
 * (kr_decode_slab_next / kr_decode_part_rows_next): nothing about a launch is decided by an earlier call.
 *   atomic_out != 0 : a ksplit == 2 launch ADDS its two K ranges into ONE f32 slab out_f32 [M][ldc] with float atomics
 *                     instead of writing two slabs (the ONE-slab form of the deferred split-K between a layer's down_proj
 *                     and the next layer's qkv prologue).  The slab must hold zeros; two addends onto zero give the same
 *                     bits in either arrival order, so the path stays reproducible.  The consumer passes it as part_in
 *                     with n_part_in = 1 and reads half the slab bytes in every workgroup's prologue.
 *   zero_ptr, zero_bytes : the launch also zeroes that f32 range (16-byte aligned, a multiple of 16 bytes): the engine
 *                     lets layer L's qkv launch zero the slab layer L's down_proj will add into.
 *   part_rows       : the part_in slabs have `part_rows` rows each ([n_part_in][part_rows][K]) instead of M — the launch
 *                     covers a row range of a larger batch whose down_proj wrote the slabs (decode batches above 16 rows
 *                     at the 7B width run the norm-prologue launches once per 16-row range).  0: = M. */
typedef struct kr_narrow_opts {
    float* zero_ptr;
    uint64_t zero_bytes;
    int32_t atomic_out;
    int32_t part_rows;
} kr_narrow_opts;

/* Narrow decode linears (qkv, o_proj, down_proj): one workgroup per 16-row tile (ROPE_KV: per rotary tile
 * pair), K split over its `waves` (8 or 16; the norm prologue always runs 8) waves; x / norm weight / epilogue
 * operands are requested before the weights.  Modes PLAIN and ROPE_KV as kr_linear_decode.  ROPE_KV with norm_w ==
 * NULL: x is already normalised (kr_decode_resnorm) and its fragments are read straight from L2 (M up to 32, 8 waves).
 *   ksplit > 1 (PLAIN only): K is also split over `ksplit` workgroups and the reduction is DEFERRED: out_f32
 *     receives f32 slabs [ksplit][M][ldc] (no bias / residual / norm) — or, with opts->atomic_out, one slab the
 *     K ranges add into; the consumer adds them.
 *   part_in (with norm_w): n_part_in (1 or 2) slabs [n][M][K] f32 are added to x in the prologue,
 *     x_new = bf16(x + sum of slabs) is RMS-normalised and, by workgroup 0, stored to x_out (ldxo), which
 *     must not alias x (other workgroups still read x).  K must be 1536, 2048 or 3584 for this. */
int kr_linear_decode_narrow(int mode, const kr_bf16* x, int64_t ldx, const float* part_in, int n_part_in,
                            kr_bf16* x_out, int64_t ldxo, const kr_bf16* w_packed, const kr_bf16* bias,
                            const kr_bf16* norm_w, float norm_eps, const kr_bf16* residual, int64_t ldr,
                            kr_bf16* out, float* out_f32, int64_t ldc, int M, int N, int K, int waves, int ksplit,
                            const float* cs_table, int cs_stride, const int32_t* prompt_len, const int32_t* ctx_len,
                            kr_bf16* q_out, kr_bf16* kcache, kr_bf16* vtcache, int heads, int kv_heads, int s_max,
                            const kr_narrow_opts* opts, kr_stream s);

/* Residual sum + RMSNorm ONCE for a decode batch (the form of batches above 16 rows):
 *   x_new = bf16(x + part_in[0] + part_in[1] + ...)  (n_part_in slabs [n][part_rows][K] f32, in that order; part_rows 0 = M),
 *   stored to x_out (ldxo; must not alias x; untouched when n_part_in == 0), and
 *   h = norm_w * bf16(x_new * rsqrt(mean(x_new^2) + eps))  (Qwen2VLRMSNorm, TF:96-110) stored to h [M, ldh].
 * One wave per row with the narrow NORM kernel's summation structure: the rows are bit-identical to the ones
 * kr_linear_decode_narrow's fused prologue computes, so kr_linear_decode_narrow(ROPE_KV, x = h, norm_w = NULL) — x fragments
 * straight from L2, no per-workgroup staging of 32 rows and their slabs — gives the fused launch's q / K / V bits. */
int kr_decode_resnorm(const kr_bf16* x, int64_t ldx, const float* part_in, int n_part_in, int part_rows, kr_bf16* x_out,
                      int64_t ldxo, const kr_bf16* norm_w, float norm_eps, kr_bf16* h, int64_t ldh, int M, int K, kr_stream s);

/* The same two kernels on fp8 (OCP e4m3fn) weights — BASELINE.json config 5: decoder Linears in fp8 with one f32 scale
 * per output row, activations bf16.  w_packed_fp8 = weights.pack_w16x64_fp8 (one 16-row x 64-column block = 1 KiB in
 * lane order: half the bytes per launch), converted to bf16 in registers (exact) and fed to the bf16 MFMA; w_scale
 * [N] (in the row order of the packed matrix, i.e. interleaved like the rows for SILU8) multiplies the f32
 * accumulators before bias / activation / rotary / slab store.  Everything else as the bf16 entry points. */
int kr_linear_decode_wide_fp8(int mode, const kr_bf16* x, int64_t ldx, const uint8_t* w_packed_fp8, const float* w_scale,
                              const kr_bf16* bias, const kr_bf16* norm_w, float norm_eps, const kr_bf16* residual,
                              int64_t ldr, kr_bf16* out, float* out_f32, int64_t ldc, int M, int N, int K, int blocks,
                              int waves, float* amax_val, int32_t* amax_idx, kr_stream s);
int kr_linear_decode_narrow_fp8(int mode, const kr_bf16* x, int64_t ldx, const float* part_in, int n_part_in,
                                kr_bf16* x_out, int64_t ldxo, const uint8_t* w_packed_fp8, const float* w_scale,
                                const kr_bf16* bias, const kr_bf16* norm_w, float norm_eps, const kr_bf16* residual,
                                int64_t ldr, kr_bf16* out, float* out_f32, int64_t ldc, int M, int N, int K, int waves,
                                int ksplit, const float* cs_table, int cs_stride, const int32_t* prompt_len,
                                const int32_t* ctx_len, kr_bf16* q_out, kr_bf16* kcache, kr_bf16* vtcache, int heads,
                                int kv_heads, int s_max, const kr_narrow_opts* opts, kr_stream s);
/* ---- decode batches of 17..32 rows (two 16-row MFMA column tiles): the continuous server's steady state and BASELINE
 * config 3's 32-rows-per-GPU variant (the callers: bulk_processing/workers/inference_worker.py:331-339, one in-flight
 * request per worker x workers per port).
 *
 * PACKED ACTIVATIONS ("XP").  The input of a decode linear at more than 16 rows is stored
 *     [K/64 chunks][2 column tiles][2 k-steps of 32][64 lanes = 16 * (k/8 % 4) + row % 16][8] bf16      (4 KiB per chunk)
 * i.e. element (row b, column k) at byte (k/64)*4096 + (b/16)*2048 + (k/32 % 2)*1024 + (16*(k/8 % 4) + b%16)*16 + (k%8)*2:
 * the operand of one v_mfma_f32_16x16x32_bf16 is 1 KiB of contiguous memory in lane order, so a wave fetches it as eight
 * whole cache lines (row-major x costs sixteen half lines per fragment, and at two column tiles a wave moves twice as many
 * x bytes from L2 as weight bytes from HBM).  Producers: kr_pack_rows32 (from row-major rows; rows >= M are zero),
 * kr_decode_resnorm32, kr_attn_decode_merge32, kr_linear_decode_wide(KR_DEC_SILU8 | KR_DEC_OUT_XP).  A buffer holds 32 rows
 * whatever M is: K * 64 bytes. */
int kr_pack_rows32(const kr_bf16* x, int64_t ldx, int M, int K, kr_bf16* xp, kr_stream s);

/* kr_linear_decode_narrow's contract (modes KR_DEC_PLAIN and KR_DEC_ROPE_KV, bf16 or fp8 weights, deferred split-K into
 * f32 slabs or one atomically accumulated slab, the zeroing job) for M <= 32 on packed activations, WITH THE NARROW LAUNCH'S
 * SUMMATION ORDER: `waves_ref` and `ksplit` name the K partition the <= 16-row launch of the same layer uses (its `waves`
 * and `ksplit` arguments) — the same chunk ranges accumulated from zero in ascending k, folded in the same order — so row b
 * of the result has the bits kr_linear_decode_narrow gives for that row alone: a page's tokens do not depend on its batch.
 * 8-wave workgroups; a 16-wave partition runs as two ranges per wave.  tiles_per_wg: 0 = automatic (two weight tiles share
 * one ring of x fragments where that leaves >= 192 workgroups), 1, 2 (4 with group_split).  No norm prologue: x is kr_decode_resnorm32's output. */
typedef struct kr_dec32 {
    const kr_bf16* xp;            /* packed activations, K * 64 bytes */
    const void* w_packed;         /* weights.pack_w16x32 (bf16) or pack_w16x64_fp8 (with w_scale) */
    const float* w_scale;         /* fp8 weights: one f32 per output row; NULL for bf16 */
    const kr_bf16* bias;          /* [N] or NULL (ROPE_KV: required) */
    const kr_bf16* residual;      /* PLAIN: [M, ldr] or NULL */
    int64_t ldr;
    kr_bf16* out;                 /* PLAIN: bf16 [M, ldc] ... */
    float* out_f32;               /* ... or f32 [M, ldc]; ksplit > 1: slabs [ksplit][M][ldc], or one slab with atomic_out */
    int64_t ldc;
    int32_t M, N, K;
    int32_t waves_ref, ksplit;    /* the <= 16-row launch's K partition */
    int32_t atomic_out;           /* ksplit == 2: both K ranges ADD into out_f32 (zeroed by an earlier launch) */
    int32_t tiles_per_wg;
    int32_t group_split;          /* atomic_out, ksplit 2, bf16 weights: each K range's atoms in TWO workgroups (first / second half
                                   * of the partition), each adding its half's fold into slab [ks] of out_f32 = [2][M][ldc]: the slab
                                   * holds (half 0) + (half 1) = the narrow launch's fold of that range, and the consumer adds
                                   * x + (slab 0 + slab 1) (kr_decode_resnorm32, sum_slabs_first) = x + the narrow launch's one slab.
                                   * Twice the weight tiles per workgroup at the same number of workgroups: half the x bytes */
    int32_t reserved0;
    float* zero_ptr;              /* this launch also zeroes [zero_ptr, zero_ptr + zero_bytes) (16-byte multiples) */
    uint64_t zero_bytes;
    /* KR_DEC_ROPE_KV (as kr_linear_decode): */
    const float* cs_table; int32_t cs_stride; const int32_t* prompt_len; const int32_t* ctx_len;
    kr_bf16* q_out; kr_bf16* kcache; kr_bf16* vtcache; int32_t heads, kv_heads, s_max;
} kr_dec32;
int kr_linear_decode32(int mode, const kr_dec32* args, kr_stream s);

/* kr_decode_resnorm with h written in the packed layout (h_xp: K * 64 bytes); x_out stays row-major.  sum_slabs_first
 * (n_part_in == 2): x + (slab 0 + slab 1) instead of (x + slab 0) + slab 1 — the slabs of a group_split down_proj. */
int kr_decode_resnorm32(const kr_bf16* x, int64_t ldx, const float* part_in, int n_part_in, int part_rows, kr_bf16* x_out,
                        int64_t ldxo, const kr_bf16* norm_w, float norm_eps, kr_bf16* h_xp, int M, int K, int sum_slabs_first,
                        kr_stream s);

/* n fp8 e4m3fn codes -> bf16 through the hardware conversion the kernels use (test hook: pins the number format). */
int kr_fp8_to_bf16(const uint8_t* src, kr_bf16* dst, int64_t n, kr_stream s);

/* Decode attention (q_len 1, GQA, MFMA, split over n_split key ranges), kcache / vtcache = the
 * layer's base pointers.  workspace: fp32 [batch*heads][n_split][hd+4] partials (o[hd], m, l, 2 pad: 16-byte aligned records).
 * out == NULL (the product path): only the partials are produced; kr_attn_decode_merge (or kr_linear_decode's
 * attn_partials prologue) merges them.  out != NULL with n_split == 1: the single range is written as bf16
 * [batch, heads*hd].  out != NULL with n_split > 1 — the in-launch merge by the last-arriving workgroup (counters:
 * int32 [batch*kv_heads], zero-initialised, left zero), measured slower than the merge launch — exists in experiment
 * builds only (-DKR_EXPERIMENTS, include/karanta_hip_experiments.h) and is refused otherwise. */
int kr_attn_decode_fused(const kr_bf16* q, const kr_bf16* kcache, const kr_bf16* vtcache,
                         const int32_t* ctx_len, kr_bf16* out, float* workspace, int32_t* counters,
                         int batch, int heads, int kv_heads, int hd, int s_max, int n_split,
                         float scale, kr_stream s);
/* The serving form: kr_attn_decode_fused(out = NULL) for a batch of decode SLOTS — the workgroups of a slot whose `finished` flag
 * is set (EOS seen by kr_sample_greedy, or retired by the host: an idle slot of the continuous server,
 * bulk_processing/workers/inference_worker.py:331-339 keeps the slots filled one request at a time) return at once and leave its
 * partials as they were; nothing downstream consumes a finished row.  Live rows: the same partials, bit for bit. */
int kr_attn_decode_slots(const kr_bf16* q, const kr_bf16* kcache, const kr_bf16* vtcache, const int32_t* ctx_len,
                         const int32_t* finished, float* workspace, int batch, int heads, int kv_heads, int hd, int s_max,
                         int n_split, float scale, kr_stream s);

/* Merge of the partials kr_attn_decode_fused leaves with out == NULL into bf16 out [batch, heads*hd]
 * (one small launch; the alternative to merging inside the consumer's prologue). */
int kr_attn_decode_merge(const float* workspace, kr_bf16* out, int batch, int heads, int hd, int n_split,
                         kr_stream s);

/* kr_attn_decode_merge with the merged heads written in the packed activation layout (row = sequence, column =
 * head * 128 + d; batch <= 32; out_xp: heads * 128 * 64 bytes) — the o_proj launch of a 17..32-row batch reads it. */
int kr_attn_decode_merge32(const float* workspace, kr_bf16* out_xp, int batch, int heads, int hd, int n_split,
                           kr_stream s);

/* GPU image front end.  kr_image_resize_bicubic_u8: HWC uint8 RGB [h][w][3] -> [rh][rw][3], bit-identical to
 * PIL's Image.resize(..., BICUBIC) (the HF PIL processor's resize, image_processing_pil_qwen2_vl.py:126-150):
 * horizontal pass then vertical pass of Pillow's 8-bit resample with the host-built integer tables
 * (image_processing.resample_tables: bounds [out][2] = (first input index, taps), coeffs [out][ksize], 22
 * fractional bits); an axis whose size does not change is skipped (tables may be NULL), tmp [h][rw][3] is needed
 * when both change.  kr_image_normalize_patchify: x/255 -> (x - mean) / std in fp32 (mean3 / std3 are HOST
 * arrays of 3 floats) written as pixel_values [gh*gw][3*temporal*patch*patch] in the processor's patch order
 * (:152-187, :226-229). */
int kr_image_resize_bicubic_u8(const uint8_t* src, int h, int w, uint8_t* dst, int rh, int rw, uint8_t* tmp,
                               const int32_t* h_bounds, const int32_t* h_coeffs, int h_ksize,
                               const int32_t* v_bounds, const int32_t* v_coeffs, int v_ksize, kr_stream s);
int kr_image_normalize_patchify(const uint8_t* img, int rh, int rw, const float* mean3, const float* std3,
                                int patch, int merge, int temporal, float* out, kr_stream s);

/* Temperature sampling as an argmax (Gumbel-max): per row b, partial argmax over n_part vocabulary slices of
 * logits[b][i] / T_b + G(seed_b, n_b, i), n_b = ctx_len[b] + 1 - prompt_len[b] (index of the token being
 * generated), G = -ln(-ln(u)) from a counter-based hash (definition in kr_decode.hip, restated in the oracle);
 * T_b == 0: plain argmax.  Writes amax_val / amax_idx [batch][n_part] for kr_sample_greedy, replacing the
 * greedy partials of the lm_head.  Replaces vLLM's sampler for requests with temperature > 0
 * (/root/reference/karanta/pipeline.py:281,301; bulk_processing/workers/vllm_client.py:155). */
int kr_gumbel_argmax(const float* logits, int64_t ld_logits, int vocab, const float* temperature,
                     const uint32_t* seed, const int32_t* ctx_len, const int32_t* prompt_len, float* amax_val,
                     int32_t* amax_idx, int n_part, int batch, kr_stream s);

/* ------------------------------------------------------------------ guided decoding + log-probabilities
 * Replaces vLLM's guided-decoding logits processor and its logprobs output for the requests the reference sends:
 * `guided_regex` (/root/reference/karanta/pipeline.py:304-307), `response_format` json_schema
 * (/root/reference/karanta/data/utils.py:322-440, bulk_processing/workers/vllm_client.py:196), `logprobs` /
 * `top_logprobs` (/root/reference/karanta/data/create_batch_data_prompts.py:117-118).  The host compiles the pattern
 * to a byte DFA (state 0 = dead; karanta_ocr_amd/guided.py); per token everything stays on the device.
 *
 * kr_guide_build_masks: masks[s][w] bit j = token 32*w+j allowed in DFA state s <=> the token has bytes and walking
 * them from s through trans[state][byte] (uint16, [n_states][256]) never reaches state 0; a token listed in eos is
 * allowed exactly where accept[s] != 0.  vocab_off [vocab+1] / vocab_bytes: the byte string of every token id.
 * mask_words: words per row, even, mask_words * 32 >= vocab. */
int kr_guide_build_masks(const uint16_t* trans, const uint8_t* accept, int n_states, const int32_t* vocab_off,
                         const uint8_t* vocab_bytes, int vocab, const int32_t* eos, int n_eos, uint32_t* masks,
                         int mask_words, kr_stream s);

/* kr_gumbel_argmax with a token mask per row: guide_masks[b] = device address (as an integer) of the mask table of
 * row b's pattern (0: row unconstrained), guide_state[b] its current DFA state; tokens whose bit is clear do not
 * take part.  guide_masks == NULL: identical to kr_gumbel_argmax.  fallback_token: what a row with no allowed token
 * yields (cannot happen in a live state; keeps the token id valid). */
int kr_gumbel_argmax_guided(const float* logits, int64_t ld_logits, int vocab, const float* temperature,
                            const uint32_t* seed, const int32_t* ctx_len, const int32_t* prompt_len, float* amax_val,
                            int32_t* amax_idx, int n_part, int batch, const uint64_t* guide_masks,
                            const int32_t* guide_state, int mask_words, int fallback_token, kr_stream s);

/* After the sampler: guide_state[b] <- walk(guide_state[b], bytes(tokens[b])) through the table at device address
 * guide_trans[b] (0: row unconstrained); rows with finished[b] != 0 keep their state. */
int kr_guide_advance(const int32_t* tokens, const int32_t* finished, const uint64_t* guide_trans, int32_t* guide_state,
                     const int32_t* vocab_off, const uint8_t* vocab_bytes, int vocab, int batch, kr_stream s);

/* Log-probabilities of one decode step from the fp32 logits [batch][ld_logits] (log-softmax over the vocabulary,
 * unscaled and unmasked): out_lp[h][b][0] = log p(tokens[b]), out_lp[h][b][1 + j] / out_idx[h][b][j] = the j-th most
 * probable token (ties: lowest id), j < k <= 20, rows strided by hist_batch * (1 + k_stride) / hist_batch * k_stride;
 * h = ctx_len[b] - prompt_len[b] as left by kr_sample_greedy (the token-history index); rows with finished[b] != 0
 * record nothing.  part_val / part_idx [batch][n_part][k] and part_ms [batch][n_part][2] are scratch;
 * ceil(vocab / n_part) <= 4096. */
int kr_logprobs_topk(const float* logits, int64_t ld_logits, int vocab, int k, int n_part, float* part_val,
                     int32_t* part_idx, float* part_ms, const int32_t* tokens, const int32_t* ctx_len,
                     const int32_t* prompt_len, const int32_t* finished, float* out_lp, int32_t* out_idx, int hist_len,
                     int hist_batch, int k_stride, int batch, kr_stream s);

/* Greedy sampling from the ARGMAX partials + per-step bookkeeping: token -> tokens_out[b] and
 * history[(ctx_len[b] + 1 - prompt_len[b]) * hist_stride + b] (= this sequence's generated-token
 * index), EOS / pad handling as kr_argmax_embed, ctx_len[b] += 1, embedding gather into x_next.
 * ignore_eos is a bit set: bit 0 = ignore EOS (fixed-length benchmarking), bit 1 = a finished sequence is
 * frozen (no history write, ctx_len not advanced; the pad token is still fed back) — the slot scheduler's mode:
 * a finished slot idles in place until the host prefills a new request into it. */
int kr_sample_greedy(const float* amax_val, const int32_t* amax_idx, int n_part,
                     const kr_bf16* embed_table, int d, int32_t* tokens_out, int32_t* history,
                     int hist_stride, const int32_t* prompt_len, int32_t* ctx_len, int32_t* finished,
                     const int32_t* eos, int n_eos, int pad_id, int ignore_eos, kr_bf16* x_next,
                     int batch, kr_stream s);

/* ------------------------------------------------------------------ multi-GPU: one-time weight broadcast (RCCL)
 * One process per GPU.  Rank 0 calls kr_comm_unique_id and shares the 128 bytes through any host
 * channel (the Python host uses torch.distributed's store); every rank then calls kr_comm_init.
 * kr_bcast_weights is ncclBroadcast of the packed weight arena from `root`; it is
 * the only collective on the path — steady state has none. */
#define KR_UNIQUE_ID_BYTES 128
int kr_comm_unique_id(uint8_t* id128);
int kr_comm_init(void** comm, int n_ranks, int rank, const uint8_t* id128);
/* Number of ranks RCCL itself reports for the communicator (ncclCommCount): bench.py prints it, so "did RCCL see N
 * ranks?" is answered by the library, not by the launcher's environment. */
int kr_comm_count(void* comm, int* n_ranks);
int kr_comm_destroy(void* comm);
/* ncclGetVersion of the RCCL this library is bound to (major * 10000 + minor * 100 + patch): reported beside the broadcast rate. */
int kr_rccl_version(int* version);
int kr_bcast_weights(void* comm, void* buf, size_t bytes, int root, kr_stream s);

/* ------------------------------------------------------------------ device self-tests (used by tests/ -m gpu) */
/* Runs an MFMA lane-layout check (A=I, asymmetric B) on the device; returns 0 if exact. */
int kr_selftest_mfma(kr_stream s);
/* Diagnostic: microseconds per kernel of a dependent chain of `n` tiny kernels (`blocks` x 256
 * threads; dirty != 0: each block also writes 4 KiB) replayed from a hipGraph on stream `s`. */
int kr_probe_launch_floor(kr_stream s, int n, int blocks, int dirty, float* us_per_kernel);
/* Diagnostic: the read-only streaming rate (GB/s, best of `reps` launches timed with HIP events on `s`) of `blocks` x 256
 * threads reading [ptr, ptr + bytes) once with 16-byte nontemporal loads, 8 in flight per lane — the measured counterpart of
 * the vendor HBM peak when the range is far larger than the 256 MB Infinity Cache. */
int kr_probe_stream_read(const void* ptr, size_t bytes, int blocks, int reps, kr_stream s, float* gbytes_per_s);
/* Launches an empty kernel (1 wave): calibrates the cost of a HIP-event bracket around one launch. */
int kr_launch_null(kr_stream s);

#ifdef __cplusplus
}
#endif
#endif /* KARANTA_HIP_H */
