/* karanta_hip_experiments.h — entry points of the decode experiments that were built, measured and NOT adopted
 * (profiles/r02_decode_experiments.txt).  They are compiled only with -DKR_EXPERIMENTS
 * (python karanta_ocr_amd/csrc/tools/build_variant.py exp kr_decode.hip kr_selftest.hip -DKR_EXPERIMENTS); the shipped
 * libkaranta_hip.so does not export them (tests/test_abi.py asserts that).  Same conventions as karanta_hip.h. */
#ifndef KARANTA_HIP_EXPERIMENTS_H
#define KARANTA_HIP_EXPERIMENTS_H

#include "karanta_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* FAST-RESIDUAL MODE (the deterministic path stays the parity mode).  Three entry points that remove the
 * attention-merge launch from the decode step:
 *   kr_linear_decode_narrow_x32 : kr_linear_decode_narrow (+ w_scale: fp8 codes when non-NULL) whose workgroup 0 ALSO
 *       stores x_new as f32 to x_out_f32 [M, ldxf] — the start value of the residual accumulator; pf_ptr / pf_bytes /
 *       pf_blocks: the launch (ksplit 1) gets `pf_blocks` extra workgroups whose only work is to read [pf_ptr, pf_ptr +
 *       pf_bytes) with plain loads, so the range sits in the memory-side Infinity Cache when a later launch streams it
 *       (round 2's KARANTA_PREFETCH = 4 / 5 / 6; 0 blocks: none);
 *   kr_oproj_heads : o_proj with K split BY ATTENTION HEAD: workgroup (tile group, head) merges that head's split-KV
 *       partials (attn_partials [M][heads][n_split][hd+4] f32, as kr_attn_decode_fused leaves them with out == NULL),
 *       multiplies by W_o[rows, head columns] and adds the product to x_acc [M, ld_acc] f32 with float atomics
 *       (`heads` adders per element; sums in arrival order: low f32 bits vary from run to run);
 *   kr_linear_decode_wide_x32 : kr_linear_decode_wide (SILU8 / ARGMAX) reading its x rows from that f32 accumulator,
 *       rounding them to bf16 once; workgroup 0 stores the rounded rows to x_out [M, ldxo] (may be NULL). */
int kr_linear_decode_narrow_x32(int mode, const kr_bf16* x, int64_t ldx, const float* part_in, int n_part_in,
                                kr_bf16* x_out, int64_t ldxo, float* x_out_f32, int64_t ldxf, const void* w_packed,
                                const float* w_scale, const kr_bf16* bias, const kr_bf16* norm_w, float norm_eps,
                                const kr_bf16* residual, int64_t ldr, kr_bf16* out, float* out_f32, int64_t ldc, int M,
                                int N, int K, int waves, int ksplit, const float* cs_table, int cs_stride,
                                const int32_t* prompt_len, const int32_t* ctx_len, kr_bf16* q_out, kr_bf16* kcache,
                                kr_bf16* vtcache, int heads, int kv_heads, int s_max, const kr_narrow_opts* opts,
                                const void* pf_ptr, size_t pf_bytes, int pf_blocks, kr_stream s);
int kr_oproj_heads(const float* attn_partials, int n_split, const void* w_packed, const float* w_scale, float* x_acc,
                   int64_t ld_acc, int M, int N, int heads, kr_stream s);
int kr_linear_decode_wide_x32(int mode, const float* x_f32, int64_t ldx, kr_bf16* x_out, int64_t ldxo, const void* w_packed,
                              const float* w_scale, const kr_bf16* norm_w, float norm_eps, kr_bf16* out, float* out_f32,
                              int64_t ldc, int M, int N, int K, int blocks, int waves, float* amax_val, int32_t* amax_idx,
                              kr_stream s);

/* Reads [ptr, ptr+bytes) with `blocks` workgroups of plain 16-byte loads and discards the data:
 * a software prefetch into the 256 MiB memory-side Infinity Cache for a later streaming kernel
 * (round 1 / 2: KARANTA_PREFETCH = 1 serial, 2 / 3 on a second graph branch). */
int kr_prefetch(const void* ptr, size_t bytes, int blocks, kr_stream s);

#ifdef __cplusplus
}
#endif
#endif
