// Experiment: can DEPENDENT decode launches overlap?  Launch i+1 is put on a second stream of the same captured
// graph, so it starts while launch i still runs: it requests its weights at once, then waits on per-workgroup flags
// written by launch i (relaxed agent-scope atomics, no fences) before it reads launch i's output with cache-bypassing
// loads.  Compared with the plain chain (one stream, kernel boundaries as the dependency).
// Every launch computes x_out[j] = x_in[j] + 1 (+ 0 * weights), so after N launches x must equal N everywhere:
// a stale read or a broken dependency shows up as a wrong count.  Every spin is bounded (err[0] = 1 on timeout).
// Build: hipcc -O3 --offload-arch=gfx950 overlap_chain.hip -o /tmp/overlap_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int XN = 8 * 1536;          // floats of "x" (the residual rows)
constexpr int TILE_BYTES = 48 * 1024; // one 16-row weight tile at K = 1536
constexpr int U = 12;                 // 2 KiB chunks in flight per wave
constexpr int SPIN_MAX = 1 << 20;

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__global__ void __launch_bounds__(320) chain_kernel(const char* __restrict__ W, int ntiles, const float* x_in, float* x_out,
                                                    const unsigned* wait_flags, int wait_n, unsigned wait_epoch, unsigned* sig_flags,
                                                    unsigned epoch, int overlap, int* err) {
    __shared__ float xs[XN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, W_ = blockDim.x >> 6, nblk = gridDim.x;
    const int t = blockIdx.x + nblk * wave;
    // 1. weights first: they do not depend on the previous launch
    u32x4 wbuf[U][2];
    const char* wp = W + (size_t)(t < ntiles ? t : 0) * TILE_BYTES + lane * 16;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        wbuf[u][0] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp + u * 2048));
        wbuf[u][1] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp + u * 2048 + 1024));
    }
    // 2. the dependency
    const bool flags_on = overlap == 1 || overlap == 3, bypass = overlap == 1;
    if (flags_on && wait_n > 0) {
        if (wave == 0) {
            int spins = 0;
            for (;;) {
                bool ok = true;
                for (int i = lane; i < wait_n; i += 64)
                    ok &= __hip_atomic_load(wait_flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= wait_epoch;
                if (__all(ok)) break;
                if (++spins > SPIN_MAX) { if (lane == 0) err[0] = 1; break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
    }
    // 3. x -> LDS (cache-bypassing loads when the producer may still have been running when this kernel started)
    for (int i = tid; i < XN; i += blockDim.x)
        xs[i] = bypass ? __hip_atomic_load(x_in + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : x_in[i];
    __syncthreads();
    // 4. stream the rest of the tile, keep the data alive
    unsigned acc = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= wbuf[u][0][0] ^ wbuf[u][1][3];
    for (int c = U; c < 24; ++c) {
        const u32x4 a = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp + c * 2048));
        const u32x4 b = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp + c * 2048 + 1024));
        acc ^= a[1] ^ b[2];
    }
    const float zero = (acc == 0x12345678u) ? 1.0f : 0.0f;  // never true for the fill pattern: keeps the loads
    // 5. output: workgroup b owns a slice of x_out
    const int per = (XN + nblk - 1) / nblk;
    for (int i = blockIdx.x * per + tid; i < min(XN, (int)(blockIdx.x + 1) * per); i += blockDim.x) {
        const float v = xs[i] + 1.0f + zero;
        if (bypass) __hip_atomic_store(x_out + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else x_out[i] = v;
    }
    if (flags_on) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(sig_flags + blockIdx.x, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main() {
    const int L = 28, ntiles = 1120, blocks = 224, threads = 320, reps = 4;
    const size_t wbytes = (size_t)ntiles * TILE_BYTES;
    char* W; CK(hipMalloc(&W, wbytes * L)); CK(hipMemset(W, 0x5a, wbytes * L));
    float* x[2]; CK(hipMalloc(&x[0], XN * 4)); CK(hipMalloc(&x[1], XN * 4));
    unsigned* flags; CK(hipMalloc(&flags, 2 * 256 * 4));
    int* err; CK(hipMalloc(&err, 8));
    hipStream_t s0, s1; CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
    hipEvent_t e0, e1, fork, join; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    std::vector<float> hx(XN);
    const int n = L * reps;

    for (int overlap = 0; overlap < 4; ++overlap) {
        // ---- capture: launch i reads x[i&1], writes x[(i+1)&1]; waits on flag set (i&1), signals flag set ((i+1)&1)
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
        if (overlap) { CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0)); }
        for (int i = 0; i < n; ++i) {
            hipStream_t s = (overlap && (i & 1)) ? s1 : s0;
            chain_kernel<<<blocks, threads, 0, s>>>(W + (size_t)(i % L) * wbytes, ntiles, x[i & 1], x[(i + 1) & 1],
                                                    flags + (i & 1) * 256, i == 0 ? 0 : blocks, (unsigned)i, flags + ((i + 1) & 1) * 256,
                                                    (unsigned)(i + 1), overlap, err);
        }
        if (overlap) { CK(hipEventRecord(join, s1)); CK(hipStreamWaitEvent(s0, join, 0)); }
        CK(hipStreamEndCapture(s0, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        float best = 1e9; int bad = 0, herr[2] = {0, 0};
        for (int it = 0; it < 4; ++it) {
            CK(hipMemsetAsync(x[0], 0, XN * 4, s0)); CK(hipMemsetAsync(x[1], 0, XN * 4, s0));
            CK(hipMemsetAsync(flags, 0, 2 * 256 * 4, s0)); CK(hipMemsetAsync(err, 0, 8, s0));
            CK(hipEventRecord(e0, s0));
            CK(hipGraphLaunch(ge, s0));
            CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
            CK(hipMemcpy(hx.data(), x[n & 1], XN * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(herr, err, 8, hipMemcpyDeviceToHost));
            for (int i = 0; i < XN; ++i) bad += hx[i] != (float)n;
            if (herr[0]) break;
        }
        printf("%s: %d launches, %.2f us per launch (%.2f TB/s), wrong outputs %d, spin timeout %d\n",
               overlap == 0 ? "one stream (kernel boundaries)       " : overlap == 1 ? "two streams + flags + bypass x       " : overlap == 2 ? "two streams, NO dependency (ceiling) " : "two streams + flags, cached x (unsafe)", n, best * 1e3 / n,
               wbytes / (best * 1e-3 / n) / 1e12, bad, herr[0]);
        fflush(stdout);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
