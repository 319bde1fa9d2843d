// Experiment (round 3): a TRUE persistent skeleton of a decoder layer's three weight-streaming stages — one launch for all layers,
// 256 workgroups (one per CU, 7 weight waves + 1 chain wave), the next layer's weights of a stage requested into REGISTERS as soon
// as that stage has finished the current layer (so they are resident when its seam opens a layer later), every seam an in-kernel
// all-to-all hand-off — against the same stages as three launches per layer (tools/overlap_layer.hip, mode 0: 20.5 us per layer;
// dependency-free HBM ceiling 18.1 us).  It answers, with in-kernel stamps, what a seam costs when it is NOT a kernel boundary and
// whether the weight stream hides under the chain.
//
//   stages  : small (4.7 MB: o_proj-sized), big (55 MB: gate/up), mid (27.5 MB: down_proj) of Qwen2-VL-2B at batch 8; each stage's
//             bytes are spread evenly over the 256 x 7 weight waves (3 / 29 / 16 pieces of 1 KiB per wave: 5.4 / 52 / 28.7 MB per layer);
//   seam    : the producer's chain wave stores its 96-byte slice of the 24 KB activation with sc1 (write-through) stores, drains
//             them (vmcnt(0)), and 8 of its lanes add to the 8 replicas of the stage's arrival counter (one 128-byte line each);
//             the consumer's chain wave polls ONE replica (blockIdx.x & 7) with relaxed sc1 loads + s_sleep, then reads all 24 KB with
//             16-byte sc1 loads into LDS (the only wave with no prefetch in flight: loads retire in issue order, a wave that has
//             just requested 31 KB of weights would wait for all of them before it sees the activation), workgroup barrier;
//             (MI355X_MICROARCH.md, visibility: valid forms, replicated counter row);
//   modes   : 0 = no weights at all (the bare seam chain), 1 = burst (a stage's next-layer weights right after the stage),
//             2 = spread (the big stage's next-layer weights in three parts, one after each stage of the layer).
// Every activation element is incremented once per stage: after n layers it must read 3 n.  Every spin is bounded (err = 1).
// Build: hipcc -O3 --offload-arch=gfx950 persistent_layer.hip -o /tmp/persistent_layer
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int XB = 8 * 1536 * 2, X16 = XB / 16;    // 24 KB of activation = 1536 16-byte pieces
constexpr int NWG = 256, WW = 7;                    // workgroups, weight waves per workgroup (wave 7 = the chain wave)
constexpr int NS = 3, NB = 29, NM = 16;             // 1 KiB pieces per weight wave and layer: small / big / mid
constexpr int SPIN_MAX = 1 << 20;

__device__ __forceinline__ void keep(unsigned v) { asm volatile("" ::"v"(v)); }

template <int N>
__device__ __forceinline__ void issue(u32x4 (&buf)[N], const char* p, int from, int to) {
#pragma unroll
    for (int u = 0; u < N; ++u)
        if (u >= from && u < to) buf[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + (size_t)u * 1024));
}
template <int N>
__device__ __forceinline__ unsigned fold(const u32x4 (&buf)[N]) {
    unsigned a = 0;
#pragma unroll
    for (int u = 0; u < N; ++u) a ^= buf[u][0] ^ buf[u][3];
    return a;
}

__global__ void __launch_bounds__(512) layer_kernel(const char* __restrict__ Ws, const char* __restrict__ Wb, const char* __restrict__ Wm,
                                                    size_t small_b, size_t big_b, size_t mid_b, int L, int n_layers, char* x0, char* x1,
                                                    unsigned* cnt, int mode, int* err, unsigned long long* stamps) {
    // LDS: the 24 KB activation + the MID stage's weights (7 waves x 16 KiB = 112 KiB, filled by LDS-DMA: the big stage's 31 KiB per
    // wave live in registers, which leaves none for mid next to the chain wave's 24 x 16 bytes of activation in flight)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xs = smem;
    char* const wl = smem + XB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool chain = wave == WW;
    const size_t widx = (size_t)blockIdx.x * WW + (chain ? 0 : wave);
    u32x4 ws[NS], wb[NB];
    auto ps = [&](int l) { return Ws + (size_t)(l % L) * small_b + widx * (NS * 1024) + lane * 16; };
    auto pb = [&](int l) { return Wb + (size_t)(l % L) * big_b + widx * (NB * 1024) + lane * 16; };
    auto pm = [&](int l) { return Wm + (size_t)(l % L) * mid_b + widx * (NM * 1024) + lane * 16; };
    auto issue_mid = [&](int l) {
        const char* p = pm(l);
#pragma unroll
        for (int u = 0; u < NM; ++u)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + (size_t)u * 1024),
                                             (__attribute__((address_space(3))) void*)(wl + (wave * NM + u) * 1024), 16, 0, 2);
    };
    auto fold_mid = [&]() {
        unsigned a = 0;
#pragma unroll
        for (int u = 0; u < NM; ++u) a ^= *reinterpret_cast<const unsigned*>(wl + (wave * NM + u) * 1024 + lane * 16);
        return a;
    };
    if (!chain && mode) {      // layer 0's weights
        issue(ws, ps(0), 0, NS);
        issue(wb, pb(0), 0, NB);
        issue_mid(0);
    }
    // stamps of workgroup 0's chain wave, accumulated in LDS (not in registers: the weight arrays need them):
    // [stage][wait, read, rest] in 100 MHz ticks, then first / last
    unsigned long long* const tacc = reinterpret_cast<unsigned long long*>(smem + XB + WW * NM * 1024);
    if (tid < 16) tacc[tid] = 0;
    __syncthreads();
    for (int l = 0; l < n_layers; ++l) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int k = 3 * l + s;
            const char* xin = (k & 1) ? x1 : x0;
            char* xout = (k & 1) ? x0 : x1;
            unsigned long long t0 = 0, t1 = 0, t2 = 0;
            if (chain) {
                if (lane == 0) t0 = __builtin_amdgcn_s_memrealtime();
                if (k > 0 && lane == 0) {      // the seam: ONE lane polls ONE replica of the previous stage's counter
                    const unsigned* c = cnt + ((size_t)(k - 1) * 8 + (blockIdx.x & 7)) * 32;
                    int spins = 0;
                    while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)NWG) {
                        if (++spins > SPIN_MAX) { err[0] = 1; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) t1 = __builtin_amdgcn_s_memrealtime();
                const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(xin), 0, XB, 0x00020000);
                u32x4 v[X16 / 64];
#pragma unroll
                for (int i = 0; i < X16 / 64; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (i * 64 + lane) * 16, 0, 16);
#pragma unroll
                for (int i = 0; i < X16 / 64; ++i) *reinterpret_cast<u32x4*>(xs + (i * 64 + lane) * 16) = v[i];
                if (lane == 0) t2 = __builtin_amdgcn_s_memrealtime();
            }
            __syncthreads();      // the activation is in LDS
            if (!chain) {         // "compute": this stage's weights must have landed (they are this wave's OLDEST outstanding loads)
                if (mode) {
                    if (s == 0) keep(fold(ws));
                    else if (s == 1) keep(fold(wb));
                    else {      // mid: the LDS-DMA of this stage's weights are the wave's oldest outstanding loads
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS + 10) : "memory");   // (what mode 2 has issued since: at most small + a third of big)
                        keep(fold_mid());
                    }
                }
                keep(*reinterpret_cast<const unsigned*>(xs + (tid & 255) * 16));
            } else {              // the workgroup's slice of the output: 6 pieces of 16 bytes
                const int per = X16 / NWG, i = blockIdx.x * per + lane;
                if (lane < per) {
                    f32x4 v = *reinterpret_cast<const f32x4*>(xs + i * 16);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += 1.0f;
                    const auto rdst = __builtin_amdgcn_make_buffer_rsrc(xout, 0, XB, 0x00020000);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rdst, i * 16, 0, 16);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();      // every weight wave has consumed this stage's registers; the slice is stored and drained
            if (chain) {
                if (lane < 8) __hip_atomic_fetch_add(cnt + ((size_t)k * 8 + lane) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (blockIdx.x == 0 && lane == 0) {
                    const unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
                    if (l >= n_layers / 2) { tacc[s * 3 + 0] += t1 - t0; tacc[s * 3 + 1] += t2 - t1; tacc[s * 3 + 2] += t3 - t2; }
                    if (l == n_layers / 2 && s == 0) tacc[10] = t0;
                    if (l == n_layers - 1 && s == 2) tacc[11] = t3;
                }
            } else if (mode == 1) {   // burst: the next layer's weights of THIS stage, whose registers are free now
                if (l + 1 < n_layers) {
                    if (s == 0) issue(ws, ps(l + 1), 0, NS);
                    else if (s == 1) issue(wb, pb(l + 1), 0, NB);
                    else issue_mid(l + 1);
                }
            } else if (mode == 2) {   // spread: the big stage's 31 pieces in three parts, one behind each seam of the layer
                if (s == 0) {
                    if (l + 1 < n_layers) issue(ws, ps(l + 1), 0, NS);
                    if (l > 0) issue(wb, pb(l), 20, NB);          // big(l)'s last third, one stage before it is consumed
                } else if (s == 1) {
                    if (l + 1 < n_layers) issue(wb, pb(l + 1), 0, 10);
                } else if (l + 1 < n_layers) {
                    issue(wb, pb(l + 1), 10, 20);
                    issue_mid(l + 1);
                }
            }
        }
    }
    if (chain && blockIdx.x == 0 && lane == 0 && stamps) {
        for (int j = 0; j < 9; ++j) stamps[j] = tacc[j];
        stamps[9] = tacc[11] - tacc[10];
    }
}

int main() {
    const int L = 28, reps = 4, n = L * reps;
    const size_t small_b = (size_t)NWG * WW * NS * 1024, big_b = (size_t)NWG * WW * NB * 1024, mid_b = (size_t)NWG * WW * NM * 1024;
    char *Ws, *Wb, *Wm;
    CK(hipMalloc(&Ws, small_b * L)); CK(hipMalloc(&Wb, big_b * L)); CK(hipMalloc(&Wm, mid_b * L));
    CK(hipMemset(Ws, 0x5a, small_b * L)); CK(hipMemset(Wb, 0x5a, big_b * L)); CK(hipMemset(Wm, 0x5a, mid_b * L));
    char* x[2]; CK(hipMalloc(&x[0], XB)); CK(hipMalloc(&x[1], XB));
    const size_t cnt_bytes = (size_t)3 * n * 8 * 128;
    unsigned* cnt; CK(hipMalloc(&cnt, cnt_bytes));
    int* err; CK(hipMalloc(&err, 16));
    unsigned long long* stamps; CK(hipMalloc(&stamps, 16 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&layer_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, XB + WW * NM * 1024 + 128));
    std::vector<float> hx(XB / 4);
    int cus = 0; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    if (cus < NWG) { printf("needs %d CUs, found %d\n", NWG, cus); return 1; }
    printf("bytes per layer: small %.1f MB, big %.1f MB, mid %.1f MB\n", small_b / 1e6, big_b / 1e6, mid_b / 1e6);
    const char* names[3] = {"persistent, NO weights (bare seam chain)             ", "persistent, weights burst after their stage         ",
                            "persistent, big spread over the layer's three seams  "};
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9; int bad = 0, herr[4] = {0, 0, 0, 0};
        unsigned long long hs[16] = {};
        for (int it = 0; it < 4; ++it) {
            CK(hipMemset(x[0], 0, XB)); CK(hipMemset(x[1], 0, XB)); CK(hipMemset(cnt, 0, cnt_bytes)); CK(hipMemset(err, 0, 16));
            CK(hipMemset(stamps, 0, 16 * 8));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            layer_kernel<<<NWG, 512, XB + WW * NM * 1024 + 128>>>(Ws, Wb, Wm, small_b, big_b, mid_b, L, n, x[0], x[1], cnt, mode, err, stamps);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) { best = ms; CK(hipMemcpy(hs, stamps, sizeof(hs), hipMemcpyDeviceToHost)); }
            CK(hipMemcpy(hx.data(), x[(3 * n) & 1], XB, hipMemcpyDeviceToHost));
            CK(hipMemcpy(herr, err, 16, hipMemcpyDeviceToHost));
            bad = 0;
            for (size_t q = 0; q < hx.size(); ++q) bad += hx[q] != (float)(3 * n);
            if (herr[0]) break;
        }
        const double bytes = mode ? (double)(small_b + big_b + mid_b) : 0.0;
        printf("%s: %d layers, %.2f us per layer (%.2f TB/s of weights), wrong outputs %d, spin timeout %d\n", names[mode], n, best * 1e3 / n,
               bytes / (best * 1e-3 / n) / 1e12, bad, herr[0]);
        const double m = n - n / 2;
        const char* sn[3] = {"small", "big  ", "mid  "};
        for (int s = 0; s < 3; ++s)
            printf("    %s stage, workgroup 0: seam wait %.2f us, 24 KB activation read (one wave, sc1) %.2f us, barrier + compute + slice + signal %.2f us\n",
                   sn[s], hs[s * 3] / m / 100.0, hs[s * 3 + 1] / m / 100.0, hs[s * 3 + 2] / m / 100.0);
        printf("    layer period by in-kernel clock (second half of the run): %.2f us\n", hs[9] / m / 100.0);
        fflush(stdout);
    }
    return 0;
}
