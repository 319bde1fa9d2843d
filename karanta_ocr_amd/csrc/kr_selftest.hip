// Device self-test: pins the MFMA lane layouts the kernels rely on with exact integer data
// (A = asymmetric integers, B = asymmetric integers, compared against an integer matmul), so
// that a transposed C/D map cannot pass (cdna_hip_programming.md §3).
#include "kr_common.h"

namespace {

__global__ void selftest_mfma_kernel(int* fail) {
    const int lane = threadIdx.x & 63;
    int bad = 0;
    // ---- 16x16x32: A[i][k] = (i*3 + k) % 7 - 3, B[k][j] = (k*5 + j*2) % 9 - 4
    {
        const int r = lane & 15, g = lane >> 4;
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * g + j;
            a[j] = f2bf((float)((r * 3 + k) % 7 - 3));
            b[j] = f2bf((float)((k * 5 + r * 2) % 9 - 4));
        }
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
        for (int reg = 0; reg < 4; ++reg) {
            const int row = 4 * g + reg, col = r;
            int ref = 0;
            for (int k = 0; k < 32; ++k) ref += ((row * 3 + k) % 7 - 3) * ((k * 5 + col * 2) % 9 - 4);
            if (c[reg] != (float)ref) bad |= 1;
        }
    }
    // ---- 32x32x16
    {
        const int r = lane & 31, h = lane >> 5;
        bf16x8 a, b;
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * h + j;
            a[j] = f2bf((float)((r * 3 + k) % 7 - 3));
            b[j] = f2bf((float)((k * 5 + r * 2) % 9 - 4));
        }
        f32x16 c;
        for (int i = 0; i < 16; ++i) c[i] = 0.f;
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
        for (int reg = 0; reg < 16; ++reg) {
            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h, col = r;
            int ref = 0;
            for (int k = 0; k < 16; ++k) ref += ((row * 3 + k) % 7 - 3) * ((k * 5 + col * 2) % 9 - 4);
            if (c[reg] != (float)ref) bad |= 2;
        }
    }
    if (bad) atomicOr(fail, bad);
}

}  // namespace

extern "C" int kr_selftest_mfma(kr_stream s) {
    int* d = nullptr;
    KR_CHECK_HIP(hipMalloc(&d, sizeof(int)));
    KR_CHECK_HIP(hipMemsetAsync(d, 0, sizeof(int), kr_hs(s)));
    selftest_mfma_kernel<<<1, 64, 0, kr_hs(s)>>>(d);
    int h = -1;
    hipError_t e = hipMemcpyAsync(&h, d, sizeof(int), hipMemcpyDeviceToHost, kr_hs(s));
    if (e == hipSuccess) e = hipStreamSynchronize(kr_hs(s));
    hipFree(d);
    if (e != hipSuccess) {
        kr_set_error("kr_selftest_mfma: %s", hipGetErrorString(e));
        return KR_ERR_HIP;
    }
    if (h != 0) {
        kr_set_error("kr_selftest_mfma: layout mismatch mask=%d (1: 16x16x32, 2: 32x32x16)", h);
        return KR_ERR_STATE;
    }
    return KR_OK;
}

// ---------------------------------------------------------------------------------------------
// Launch-floor probe (diagnostic, used by tools/ and tests): time a dependent chain of `n` tiny
// kernels replayed from a hipGraph on the caller's stream; returns microseconds per kernel.
namespace {
__global__ void floor_tiny_kernel(int* p) {
    if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1;
}
__global__ void floor_dirty_kernel(float4* p, int n4) {  // every block dirties 4 KiB
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
}  // namespace

extern "C" int kr_probe_launch_floor(kr_stream s, int n, int blocks, int dirty, float* us_per_kernel) {
    KR_CHECK_ARG(n > 0 && blocks > 0 && us_per_kernel, "kr_probe_launch_floor: bad args");
    int* d = nullptr;
    float4* buf = nullptr;
    KR_CHECK_HIP(hipMalloc(&d, 4));
    KR_CHECK_HIP(hipMalloc(&buf, (size_t)blocks * 256 * 16));
    KR_CHECK_HIP(hipMemsetAsync(d, 0, 4, kr_hs(s)));
    hipGraph_t g;
    hipGraphExec_t ge;
    KR_CHECK_HIP(hipStreamSynchronize(kr_hs(s)));
    KR_CHECK_HIP(hipStreamBeginCapture(kr_hs(s), hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n; ++i) {
        const bool use_dirty = dirty == 2 ? (i & 1) : dirty != 0;  // dirty == 2: alternate the two kernels
        if (use_dirty)
            floor_dirty_kernel<<<blocks, 256, 0, kr_hs(s)>>>(buf, blocks * 256);
        else
            floor_tiny_kernel<<<blocks, 256, 0, kr_hs(s)>>>(d);
    }
    KR_CHECK_HIP(hipStreamEndCapture(kr_hs(s), &g));
    KR_CHECK_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    KR_CHECK_HIP(hipEventCreate(&e0));
    KR_CHECK_HIP(hipEventCreate(&e1));
    KR_CHECK_HIP(hipGraphLaunch(ge, kr_hs(s)));
    KR_CHECK_HIP(hipStreamSynchronize(kr_hs(s)));
    KR_CHECK_HIP(hipEventRecord(e0, kr_hs(s)));
    for (int i = 0; i < 10; ++i) KR_CHECK_HIP(hipGraphLaunch(ge, kr_hs(s)));
    KR_CHECK_HIP(hipEventRecord(e1, kr_hs(s)));
    KR_CHECK_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    KR_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    *us_per_kernel = ms * 1e3f / (10.0f * n);
    hipGraphExecDestroy(ge);
    hipGraphDestroy(g);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(d);
    hipFree(buf);
    return KR_OK;
}

// Read-only streaming rate of this device ("report achieved-vs-measured-stream too"): every wave keeps
// 8 independent 16-byte nontemporal loads per lane in flight over a caller-owned range (the weight arena: far larger than the
// 256 MB Infinity Cache), nothing is written.  bench.py puts the figure next to the 8 TB/s vendor peak on its JSON line.
namespace {
__global__ void __launch_bounds__(256) stream_read_kernel(const u32x4* __restrict__ p, size_t n16, unsigned* sink) {
    u32x4 acc = {0u, 0u, 0u, 0u};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 7 * stride < n16; i += 8 * stride) {
        u32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(p + i + k * stride);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc ^= v[k];
    }
    for (; i < n16; i += stride) acc ^= __builtin_nontemporal_load(p + i);
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9e3779b9u && sink) sink[0] = 1;  // keeps the loads alive
}
}  // namespace

extern "C" int kr_probe_stream_read(const void* ptr_, size_t bytes, int blocks, int reps, kr_stream s, float* gbytes_per_s) {
    KR_CHECK_ARG(ptr_ && ((uintptr_t)ptr_ & 15) == 0 && bytes >= (1u << 20) && blocks > 0 && reps > 0 && gbytes_per_s,
                 "kr_probe_stream_read: bad args");
    hipEvent_t e0, e1;
    KR_CHECK_HIP(hipEventCreate(&e0));
    KR_CHECK_HIP(hipEventCreate(&e1));
    const u32x4* p = reinterpret_cast<const u32x4*>(ptr_);
    stream_read_kernel<<<blocks, 256, 0, kr_hs(s)>>>(p, bytes / 16, nullptr);   // warm-up (code object, TLB)
    float best = 0.f;
    for (int r = 0; r < reps; ++r) {
        KR_CHECK_HIP(hipEventRecord(e0, kr_hs(s)));
        stream_read_kernel<<<blocks, 256, 0, kr_hs(s)>>>(p, bytes / 16, nullptr);
        KR_CHECK_HIP(hipEventRecord(e1, kr_hs(s)));
        KR_CHECK_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        KR_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
        const float rate = (float)((double)(bytes / 16 * 16) / 1e9 / ((double)ms * 1e-3));
        best = rate > best ? rate : best;
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    KR_CHECK_LAUNCH();
    *gbytes_per_s = best;
    return KR_OK;
}

// A kernel that does nothing: bench.py brackets it with the same two HIP events it puts around the
// roofline kernel, to calibrate the bracket's own cost (event packets + dispatch latency) live.
namespace { __global__ void null_kernel() {} }
extern "C" int kr_launch_null(kr_stream s) {
    null_kernel<<<1, 64, 0, kr_hs(s)>>>();
    KR_CHECK_LAUNCH();
    return KR_OK;
}

// Touch a byte range with plain (allocating) 16-byte loads: pulls it into the memory-side Infinity
// Cache ahead of the kernels that will stream it.  8 independent loads in flight per lane and iteration, so a few
// hundred waves reach a useful rate while they run beside the latency-bound launches of the other graph branch.
#ifdef KR_EXPERIMENTS   // Infinity-Cache prefetch launch (decode experiments of rounds 1-2; include/karanta_hip_experiments.h)
namespace {
__global__ void __launch_bounds__(256) prefetch_kernel(const u32x4* __restrict__ p, size_t n16, unsigned* sink) {
    u32x4 acc = {0u, 0u, 0u, 0u};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 7 * stride < n16; i += 8 * stride) {
        u32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[i + k * stride];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc ^= v[k];
    }
    for (; i < n16; i += stride) acc ^= p[i];
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x9e3779b9u && sink) sink[0] = 1;  // keeps the loads alive
}
}  // namespace
extern "C" int kr_prefetch(const void* ptr_, size_t bytes, int blocks, kr_stream s) {
    KR_CHECK_ARG(ptr_ && blocks > 0 && ((uintptr_t)ptr_ & 15) == 0, "kr_prefetch: bad args");
    if (bytes < 16) return KR_OK;
    prefetch_kernel<<<blocks, 256, 0, kr_hs(s)>>>(reinterpret_cast<const u32x4*>(ptr_), bytes / 16, nullptr);
    KR_CHECK_LAUNCH();
    return KR_OK;
}
#endif  // KR_EXPERIMENTS


// ---- fp8 e4m3fn -> bf16 through the instruction the decode kernels use (pins the hardware's number format)
namespace {
__global__ void fp8_to_bf16_kernel(const uint8_t* __restrict__ src, kr_bf16* __restrict__ dst, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i >= n) return;
    const int w = (int)src[i] | (i + 1 < n ? (int)src[i + 1] << 8 : 0);
    const unsigned v = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, false));
    dst[i] = (kr_bf16)(v & 0xffffu);
    if (i + 1 < n) dst[i + 1] = (kr_bf16)(v >> 16);
}
}  // namespace

extern "C" int kr_fp8_to_bf16(const uint8_t* src, kr_bf16* dst, int64_t n, kr_stream s) {
    KR_CHECK_ARG(src && dst && n >= 0, "kr_fp8_to_bf16: bad args");
    if (n == 0) return KR_OK;
    fp8_to_bf16_kernel<<<(unsigned)((n / 2 + 256) / 256), 256, 0, kr_hs(s)>>>(src, dst, n);
    KR_CHECK_LAUNCH();
    return KR_OK;
}
