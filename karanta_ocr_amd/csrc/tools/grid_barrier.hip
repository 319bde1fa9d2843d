// Micro-benchmark: what does a software grid barrier cost on gfx950 (8 XCDs, non-coherent L2s) compared with a
// kernel boundary (~1.7 us dispatch + ramp)?  Each round: every workgroup publishes a value, barrier, every
// workgroup checks values published by workgroups of OTHER XCDs (so the release/acquire really has to work).
// Every spin is bounded: a barrier that does not complete sets *err and all waves leave.
// Build: hipcc -O3 --offload-arch=gfx950 grid_barrier.hip -o /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int SPIN_MAX = 1 << 22;

// mode 0: one counter, monotonic (target = round * nblocks)
__device__ __forceinline__ bool barrier_counter(unsigned* cnt, unsigned target, int* err) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > SPIN_MAX) { *err = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    return true;
}

// mode 1: two levels: 8 group counters (by blockIdx & 7 = XCD), the last arrival of a group bumps the root
__device__ __forceinline__ void barrier_tree(unsigned* cnt, unsigned round, int nblocks, int* err) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const int g = blockIdx.x & 7;
        const unsigned per = (nblocks + 7 - g) / 8;
        const unsigned old = __hip_atomic_fetch_add(cnt + 32 * (1 + g), 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == round * per) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        const unsigned groups = nblocks < 8 ? nblocks : 8;
        while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < round * groups) {
            if (++spins > SPIN_MAX) { *err = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
}

// mode 2: flag array, no read-modify-write: block b stores flags[b] = round; wave 0 polls all flags
__device__ __forceinline__ void barrier_flags(unsigned* flags, unsigned round, int nblocks, int* err) {
    __syncthreads();
    if (threadIdx.x < 64) {
        if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.x, round, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        for (;;) {
            bool ok = true;
            for (int i = threadIdx.x; i < nblocks; i += 64)
                ok &= __hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= round;
            if (__all(ok)) break;
            if (++spins > SPIN_MAX) { *err = 1; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// mode 3: no cache maintenance at all: payload and flags move with relaxed agent-scope atomics (they bypass the
// non-coherent L2), ordering by s_waitcnt vmcnt(0) (__syncthreads) only
__device__ __forceinline__ void barrier_flags_nofence(unsigned* flags, unsigned round, int nblocks, int* err) {
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x < 64) {
        if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.x, round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        for (;;) {
            bool ok = true;
            for (int i = threadIdx.x; i < nblocks; i += 64)
                ok &= __hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= round;
            if (__all(ok)) break;
            if (++spins > SPIN_MAX) { *err = 1; break; }
        }
    }
    __syncthreads();
}

__global__ void __launch_bounds__(512) bar_kernel(unsigned* sync, float* data, int rounds, int mode, int payload, int* err) {
    const int nb = gridDim.x, b = blockIdx.x;
    for (int r = 1; r <= rounds; ++r) {
        // publish: payload floats per block (coalesced)
        float* buf = data + (size_t)(r & 1) * nb * payload;  // double buffer: a workgroup is at most one barrier ahead
        if (mode == 3) {
            for (int i = threadIdx.x; i < payload; i += blockDim.x)
                __hip_atomic_store(buf + (size_t)b * payload + i, (float)(r * 1000 + b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            for (int i = threadIdx.x; i < payload; i += blockDim.x) buf[(size_t)b * payload + i] = (float)(r * 1000 + b);
        }
        if (mode == 3) barrier_flags_nofence(sync, r, nb, err);
        else if (mode == 0) barrier_counter(sync, (unsigned)r * nb, err);
        else if (mode == 1) barrier_tree(sync, r, nb, err);
        else barrier_flags(sync, r, nb, err);
        if (*(volatile int*)err) return;
        // consume: values of 4 other blocks (other XCDs)
        if (threadIdx.x < 4) {
            const int ob = (b + 1 + threadIdx.x * 3 + r) % nb;
            const float v = mode == 3 ? __hip_atomic_load(buf + (size_t)ob * payload + (payload - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                      : buf[(size_t)ob * payload + (payload - 1)];
            if (v != (float)(r * 1000 + ob)) atomicAdd(err + 1, 1);
        }
    }
}

int main(int argc, char** argv) {
    hipStream_t s; CK(hipStreamCreate(&s));
    unsigned* sync; float* data; int* err;
    CK(hipMalloc(&sync, 4096 * 4)); CK(hipMalloc(&data, 1024 * 4096 * 4)); CK(hipMalloc(&err, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int rounds = 2000;
    for (int nb : {64, 256}) for (int threads : {512}) for (int payload : {64, 4096}) for (int mode : {0, 2, 3}) {
        if (nb == 512 && threads == 512 && 0) continue;
        float best = 1e9; int herr[2] = {0, 0};
        for (int it = 0; it < 3; ++it) {
            CK(hipMemsetAsync(sync, 0, 4096 * 4, s)); CK(hipMemsetAsync(err, 0, 8, s));
            CK(hipEventRecord(e0, s));
            bar_kernel<<<nb, threads, 0, s>>>(sync, data, rounds, mode, payload, err);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
            CK(hipMemcpy(herr, err, 8, hipMemcpyDeviceToHost));
            if (herr[0]) break;
        }
        printf("blocks=%3d threads=%3d payload=%4d floats mode=%d (%s): %.2f us/round  timeout=%d stale_reads=%d\n", nb, threads, payload,
               mode, mode == 0 ? "counter" : mode == 1 ? "tree" : mode == 2 ? "flags" : "flags, no fences", best * 1e3 / rounds, herr[0], herr[1]);
        fflush(stdout);
    }
    return 0;
}
