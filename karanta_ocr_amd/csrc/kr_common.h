// Shared device/host helpers for libkaranta_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/karanta_hip.h"
#ifdef KR_EXPERIMENTS
#include "../../include/karanta_hip_experiments.h"
#endif

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define KR_WAVE 64

// ---------------------------------------------------------------- error plumbing (host)
void kr_set_error(const char* fmt, ...);

#define KR_CHECK_ARG(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            kr_set_error(__VA_ARGS__);          \
            return KR_ERR_ARG;                  \
        }                                       \
    } while (0)

#define KR_CHECK_HIP(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            kr_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return KR_ERR_HIP;                                                          \
        }                                                                               \
    } while (0)

#define KR_CHECK_LAUNCH()                                                               \
    do {                                                                                \
        hipError_t _e = hipGetLastError();                                              \
        if (_e != hipSuccess) {                                                         \
            kr_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return KR_ERR_HIP;                                                          \
        }                                                                               \
    } while (0)

static inline hipStream_t kr_hs(kr_stream s) { return reinterpret_cast<hipStream_t>(s); }

// hipFuncSetAttribute is per (function, DEVICE): a launcher keeps one flag per device, not per process, so the
// attribute is also set the day one process drives several GPUs (one process per GPU is the engine's model; a
// capture in progress never reaches this because the engine runs every launch eagerly once before capturing).
#define KR_MAX_DEVICES 64
struct KrPerDeviceOnce {
    bool done[KR_MAX_DEVICES] = {};
    bool need() {                       // true exactly once per device (the caller then sets its attributes)
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= KR_MAX_DEVICES) return true;
        if (done[dev]) return false;
        done[dev] = true;
        return true;
    }
};

// ---------------------------------------------------------------- bf16 <-> f32 (device)
__device__ __forceinline__ float bf2f(__bf16 v) { return (float)v; }
__device__ __forceinline__ __bf16 f2bf(float v) { return (__bf16)v; }  // v_cvt_pk_bf16_f32: RNE, NaN kept
__device__ __forceinline__ float bfbits2f(unsigned short b) { return __uint_as_float(((unsigned int)b) << 16); }
__device__ __forceinline__ float bfround(float v) { return (float)((__bf16)v); }

// 16-byte vector load/store of 8 bf16
__device__ __forceinline__ bf16x8 ld8(const kr_bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ void st8(kr_bf16* p, bf16x8 v) { *reinterpret_cast<bf16x8*>(p) = v; }
__device__ __forceinline__ bf16x8 ld8_nt(const kr_bf16* p) {
    return __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p));
}

// ---------------------------------------------------------------- packed activations of 17..32-row decode batches (XP layout)
// [K/64 chunks][2 column tiles of 16 rows][2 k-steps of 32][64 lanes = (fg, fr)][8] bf16: the operand of one
// v_mfma_f32_16x16x32_bf16 (16 rows x 32 k) is 1 KiB of contiguous memory in lane order, as a weight fragment is
// (include/karanta_hip.h, kr_pack_rows32).  Byte offset of element (row b, column k):
__host__ __device__ __forceinline__ int64_t kr_xp_byte_offset(int b, int k) {
    return (int64_t)(k >> 6) * 4096 + (b >> 4) * 2048 + ((k >> 5) & 1) * 1024 + ((((k >> 3) & 3) * 16 + (b & 15)) * 16) + (k & 7) * 2;
}

// ---------------------------------------------------------------- V^T blocks of the KV cache / the ViT's V^T buffer
// A block holds 64 keys of one head, transposed, as TWO CONTIGUOUS 32-KEY HALVES: [2][HD channels][32 keys] (rounds 1-3:
// [HD][64]).  The decode attention reads a 32-key unit per wave: with [HD][64] that was one half of every 128-byte line of the
// block (16 rows x 64 B per wave-instruction), which streams at 4.0 TB/s where whole lines reach 6.5 (profiles/r04_halfline_read.txt);
// with the halves contiguous a unit is 8 KiB (hd 128) of whole lines.  Element offset of (channel d, key 0..63) inside a block:
__host__ __device__ __forceinline__ int kr_vt_off(int d, int key, int hd) { return ((key >> 5) * hd + d) * 32 + (key & 31); }

// ---------------------------------------------------------------- wave reductions (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------- activations
// x * sigmoid(k x) with v_rcp_f32 (1 ulp) instead of the IEEE division sequence (v_div_scale / fmas / fixup: ~10
// instructions per element in every GEMM epilogue); the result is rounded to bf16 right after
__device__ __forceinline__ float act_quick_gelu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x)); }
__device__ __forceinline__ float act_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float act_silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// XCD-aware, bijective remap of a 1-D block id: blocks that share an XCD (bid % 8) get a
// contiguous chunk of the work list, so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
    const unsigned q = nwg >> 3, r = nwg & 7u, xcd = bid & 7u;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}
