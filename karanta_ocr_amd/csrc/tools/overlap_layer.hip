// Experiment (round 3): what does a FLAG SEAM between kernels on different branches of one hipGraph cost on this part, and
// what does it buy when the waiting kernel already holds its weights in registers?
//
// A decoder layer at batch 8 is modelled by its three weight-streaming stages with the real byte counts of Qwen2-VL-2B:
//     small  ( 96 workgroups x 8 waves,  4.7 MB: o_proj-sized; stands for the latency-bound qkv -> attention -> o_proj chain)
//     big    (224 workgroups x 5 waves, 55.0 MB: gate/up, one whole 48 KB weight tile per wave)
//     mid    (192 workgroups x 8 waves, 27.5 MB: down_proj)
// chained small_i -> big_i -> mid_i -> small_{i+1} through a 24 KB activation buffer (8 rows x 1536 bf16).  Every kernel
// requests ALL its weights first (they do not depend on the predecessor), then takes the dependency, reads the activation,
// folds it with its weights into a checksum-free "+1" and stores its slice of the output.
//   mode 0: ONE stream, kernel boundaries are the dependencies (what the engine does today);
//   mode 1: three graph branches (one per stage kind); the dependency is a counter the producer's workgroups add to after
//           draining their write-through (sc1) stores, polled by one lane per consumer workgroup; the activation is read with
//           16-byte sc1 loads (MI355X_MICROARCH.md, visibility: valid forms, row 1);
//   mode 2: the same three branches with NO dependency at all (the concurrency ceiling: pure HBM time);
//   mode 3: two branches — small on one, big + mid in stream order on the other (big -> mid is a kernel boundary);
//   mode 4, 5: mode 1 with a PACED prefetch: a waiting kernel requests its weights four 1 KiB pieces at a time with an s_sleep
//           between the groups (~8 / ~16 us for a whole gate/up tile), so that the memory system never holds the whole stage's
//           requests at once — the throttled-loader form of a persistent engine, at launch level.
// After n layers every element of x must equal 3 n: a stale read or a broken dependency shows as a wrong count.  Every spin
// is bounded (err[0] = 1 on timeout: the run is reported as failed, nothing hangs).
// Build: hipcc -O3 --offload-arch=gfx950 overlap_layer.hip -o /tmp/overlap_layer
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int XB = 8 * 1536 * 2;      // bytes of the activation rows
constexpr int X16 = XB / 16;          // 1536 16-byte pieces
constexpr int SPIN_MAX = 1 << 18;

// NCH = 1 KiB pieces per wave held in registers (all requested before the dependency)
template <int NCH, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) stage_kernel(const char* __restrict__ W, size_t wave_bytes_stride, const char* x_in, char* x_out,
                                                           const unsigned* wait_cnt, unsigned wait_n, unsigned* sig_cnt, int flags, int* err,
                                                           unsigned long long* stamps, int pace) {
    extern __shared__ __attribute__((aligned(16))) char xs[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t widx = (size_t)blockIdx.x * WAVES + wave;
    // 1. weights first
    u32x4 wbuf[NCH];
    const char* wp = W + widx * wave_bytes_stride + lane * 16;
#pragma unroll
    for (int u = 0; u < NCH; ++u) {
        wbuf[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp + (size_t)u * 1024));
        if (pace > 0 && (u & 3) == 3) {            // paced prefetch: a pause after every 4 KiB per wave
            __builtin_amdgcn_sched_barrier(0);
            for (int q = 0; q < pace; ++q) __builtin_amdgcn_s_sleep(32);   // ~2048 cycles ~ 0.9 us per unit
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // 2. the dependency: one lane polls ONE counter
    unsigned long long t0 = 0, t1 = 0, t2 = 0;
    if (stamps && tid == 0) t0 = __builtin_amdgcn_s_memrealtime();
    if (flags && wait_n > 0) {
        if (tid == 0) {
            int spins = 0;
            while (__hip_atomic_load(wait_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < wait_n) {
                if (++spins > SPIN_MAX) { err[0] = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
    if (stamps && tid == 0) t1 = __builtin_amdgcn_s_memrealtime();
    // 3. activation -> LDS: 16-byte write-through-side (sc1) loads when the producer is not behind a kernel boundary
    {
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(x_in), 0, XB, 0x00020000);
        for (int i = tid; i < X16; i += WAVES * 64) {
            u32x4 v;
            if (flags) v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, i * 16, 0, 16);
            else v = *reinterpret_cast<const u32x4*>(x_in + i * 16);
            *reinterpret_cast<u32x4*>(xs + i * 16) = v;
        }
    }
    __syncthreads();
    if (stamps && tid == 0) t2 = __builtin_amdgcn_s_memrealtime();
    // 4. "compute": fold the weights (kept alive), +1 on this workgroup's slice of the activation (f32 words)
    unsigned acc = 0;
#pragma unroll
    for (int u = 0; u < NCH; ++u) acc ^= wbuf[u][0] ^ wbuf[u][3];
    const float zero = (acc == 0x12345678u) ? 1.0f : 0.0f;
    const int per = (X16 + gridDim.x - 1) / gridDim.x;
    const int i0 = blockIdx.x * per, i1 = min(X16, i0 + per);
    {
        const auto rdst = __builtin_amdgcn_make_buffer_rsrc(x_out, 0, XB, 0x00020000);
        for (int i = i0 + tid; i < i1; i += WAVES * 64) {
            f32x4 v = *reinterpret_cast<const f32x4*>(xs + i * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += 1.0f + zero;
            if (flags) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rdst, i * 16, 0, 16);
            else *reinterpret_cast<f32x4*>(x_out + i * 16) = v;
        }
    }
    if (flags) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(sig_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (stamps && tid == 0 && blockIdx.x < 4) {
        stamps[blockIdx.x * 4 + 0] = t0; stamps[blockIdx.x * 4 + 1] = t1; stamps[blockIdx.x * 4 + 2] = t2;
        stamps[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
    }
}

int main(int argc, char** argv) {
    const int L = 28, reps = 4, n = L * reps;
    // per-wave register-resident bytes: small 96 x 8 waves x 6 KiB = 4.7 MB; big 224 x 5 x 48 KiB = 55 MB; mid 192 x 8 x 18 KiB = 27.6 MB
    constexpr int NS = 6, NB = 48, NM = 18;
    const size_t small_b = (size_t)96 * 8 * NS * 1024, big_b = (size_t)224 * 5 * NB * 1024, mid_b = (size_t)192 * 8 * NM * 1024;
    char *Ws, *Wb, *Wm;
    CK(hipMalloc(&Ws, small_b * L)); CK(hipMalloc(&Wb, big_b * L)); CK(hipMalloc(&Wm, mid_b * L));
    CK(hipMemset(Ws, 0x5a, small_b * L)); CK(hipMemset(Wb, 0x5a, big_b * L)); CK(hipMemset(Wm, 0x5a, mid_b * L));
    char* x[2]; CK(hipMalloc(&x[0], XB)); CK(hipMalloc(&x[1], XB));
    unsigned* cnt; const int ncnt = 3 * n + 4; CK(hipMalloc(&cnt, ncnt * 4));
    int* err; CK(hipMalloc(&err, 16));
    unsigned long long* stamps; CK(hipMalloc(&stamps, 3 * n * 16 * 8)); CK(hipMemset(stamps, 0, 3 * n * 16 * 8));
    hipStream_t s[3]; for (auto& q : s) CK(hipStreamCreate(&q));
    hipEvent_t e0, e1, fork, join1, join2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join1, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&join2, hipEventDisableTiming));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stage_kernel<NS, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, XB));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stage_kernel<NB, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, XB));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stage_kernel<NM, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, XB));
    std::vector<float> hx(XB / 4);
    const char* names[6] = {"one stream (kernel boundaries)                 ", "three branches + counters + sc1 activations      ",
                            "three branches, NO dependency (HBM ceiling)     ", "two branches (small | big -> mid) + counters     ",
                            "three branches + counters, prefetch paced ~8 us ", "three branches + counters, prefetch paced ~16 us"};
    for (int mode = 0; mode < 6; ++mode) {
        const int flags = (mode == 1 || mode >= 3) ? 1 : 0;
        const int pace = mode == 4 ? 1 : mode == 5 ? 2 : 0;
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s[0], hipStreamCaptureModeThreadLocal));
        if (mode) {
            CK(hipEventRecord(fork, s[0])); CK(hipStreamWaitEvent(s[1], fork, 0));
            if (mode != 3) CK(hipStreamWaitEvent(s[2], fork, 0));
        }
        // kernel k of the chain (k = 3 i + stage) reads x[k & 1], writes x[(k + 1) & 1], waits on counter k - 1, signals counter k
        for (int i = 0; i < n; ++i) {
            const int l = i % L, k = 3 * i;
            hipStream_t sa = s[0], sb = mode ? s[1] : s[0], sc = (mode == 1 || mode == 2 || mode >= 4) ? s[2] : sb;
            stage_kernel<NS, 8><<<96, 512, XB, sa>>>(Ws + (size_t)l * small_b, (size_t)NS * 1024, x[k & 1], x[(k + 1) & 1], cnt + (k > 0 ? k - 1 : 0),
                                                      (flags && k > 0) ? 192u : 0u, cnt + k, flags, err, stamps + (size_t)(k) * 16, 0);
            stage_kernel<NB, 5><<<224, 320, XB, sb>>>(Wb + (size_t)l * big_b, (size_t)NB * 1024, x[(k + 1) & 1], x[k & 1], cnt + k, flags ? 96u : 0u,
                                                      cnt + k + 1, flags, err, stamps + (size_t)(k + 1) * 16, pace);
            // mode 3: mid follows big in stream order: no counter wait, plain loads would do — kept sc1 for equal code
            stage_kernel<NM, 8><<<192, 512, XB, sc>>>(Wm + (size_t)l * mid_b, (size_t)NM * 1024, x[k & 1], x[(k + 1) & 1], cnt + k + 1,
                                                      (flags && mode != 3) ? 224u : 0u, cnt + k + 2, flags, err, stamps + (size_t)(k + 2) * 16, pace);
        }
        if (mode) {
            CK(hipEventRecord(join1, s[1])); CK(hipStreamWaitEvent(s[0], join1, 0));
            if (mode != 3) { CK(hipEventRecord(join2, s[2])); CK(hipStreamWaitEvent(s[0], join2, 0)); }
        }
        CK(hipStreamEndCapture(s[0], &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        float best = 1e9; int bad = 0, herr[4] = {0, 0, 0, 0};
        for (int it = 0; it < 5; ++it) {
            CK(hipMemsetAsync(x[0], 0, XB, s[0])); CK(hipMemsetAsync(x[1], 0, XB, s[0]));
            CK(hipMemsetAsync(cnt, 0, ncnt * 4, s[0])); CK(hipMemsetAsync(err, 0, 16, s[0]));
            CK(hipEventRecord(e0, s[0]));
            CK(hipGraphLaunch(ge, s[0]));
            CK(hipEventRecord(e1, s[0])); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
            CK(hipMemcpy(hx.data(), x[(3 * n) & 1], XB, hipMemcpyDeviceToHost));
            CK(hipMemcpy(herr, err, 16, hipMemcpyDeviceToHost));
            bad = 0;
            if (mode != 2) for (size_t q = 0; q < hx.size(); ++q) bad += hx[q] != (float)(3 * n);
            if (herr[0]) break;
        }
        printf("%s: %d layers, %.2f us per layer (%.2f TB/s of weights), wrong outputs %d, spin timeout %d\n", names[mode], n, best * 1e3 / n,
               (double)(small_b + big_b + mid_b) / (best * 1e-3 / n) / 1e12, bad, herr[0]);
        if (mode == 1 || mode == 0 || mode >= 4) {   // stamps of workgroup 0 of the last layers: wait / read / compute shares (100 MHz ticks)
            std::vector<unsigned long long> st(3 * n * 16);
            CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
            for (int st_k = 0; st_k < 3; ++st_k) {
                double w = 0, r = 0, c = 0; int m = 0;
                for (int i = n / 2; i < n; ++i) {
                    const unsigned long long* p = &st[(size_t)(3 * i + st_k) * 16];
                    if (!p[3]) continue;
                    w += (double)(p[1] - p[0]); r += (double)(p[2] - p[1]); c += (double)(p[3] - p[2]); ++m;
                }
                if (m) printf("    stage %d workgroup 0: weights-issued -> dependency met %.2f us, activation read %.2f us, compute + store + signal %.2f us\n",
                              st_k, w / m / 100.0, r / m / 100.0, c / m / 100.0);
            }
            // layer period from consecutive small kernels' start stamps
            double per = 0; int m = 0;
            for (int i = n / 2 + 1; i < n; ++i) {
                const unsigned long long a = st[(size_t)(3 * (i - 1)) * 16], b = st[(size_t)(3 * i) * 16];
                if (a && b) { per += (double)(b - a); ++m; }
            }
            if (m) printf("    layer period by in-kernel clock: %.2f us\n", per / m / 100.0);
        }
        fflush(stdout);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
