// Micro-benchmark (round 4): does the SHAPE of the decode attention's K / V^T loads cost HBM bandwidth?
// The decode attention (kr_decode.hip, attn_decode2_kernel) reads a 32-key unit as
//   K   : 8 wave-instructions of 16 key rows x 64 B (row stride 256 B): two instructions share every 128-B line,
//   V^T : 8 wave-instructions of 16 channel rows x 64 B (row stride 128 B) = one HALF of every line of a 16 KiB block; the other
//         half belongs to the neighbouring unit = the next wave of the same workgroup.
// It streams 124 MB at 4.7 TB/s (7B, 32 rows) where the weight kernels reach 6.1-6.4 TB/s with whole-line loads.  Modes:
//   0  whole lines: 1 KiB contiguous per wave-instruction (the weight kernels' shape)
//   1  V^T shape: 16 rows x 64 B at stride 128 B, even / odd waves take the two halves of the same lines
//   2  V^T shape, the other half never read (useful bytes = half of the lines touched)
//   3  K shape: 16 rows x 64 B at stride 256 B, four instructions cover the rows
//   4  V^T in 32-key blocks: the same 8 KiB per unit, contiguous (what a [S/32][128][32] layout would give)
// Every wave reads `units` units of 8 KiB, 8 loads in flight, unit u of wave w at a stride that spreads waves over the buffer.
// Build: hipcc -O3 --offload-arch=gfx950 halfline_read.hip -o /tmp/halfline_read ; run: /tmp/halfline_read
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int MODE>
__global__ void __launch_bounds__(256) rd(const char* __restrict__ p, size_t bytes, int units, unsigned* sink) {
    const int lane = threadIdx.x & 63, fr = lane & 15, fg = lane >> 4;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * 4;
    u32x4 acc = {0, 0, 0, 0};
    for (int u = 0; u < units; ++u) {
        // unit index: waves interleave over the buffer as the attention's parts do (unit = part + n_part * k)
        const size_t unit = (size_t)u * nwaves + wave;
        u32x4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            size_t off;
            if (MODE == 0 || MODE == 4) off = unit * 8192 + (size_t)i * 1024 + lane * 16;
            else if (MODE == 1) off = (unit >> 1) * 16384 + (size_t)(i * 16 + fr) * 128 + (unit & 1) * 64 + fg * 16;
            else if (MODE == 2) off = unit * 16384 + (size_t)(i * 16 + fr) * 128 + fg * 16;
            else off = unit * 8192 + (size_t)((i >> 2) * 16 + fr) * 256 + (i & 3) * 64 + fg * 16;
            off %= bytes - 16;
            off &= ~(size_t)15;
            v[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + off));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc ^= v[i];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

template <int MODE>
static void run(const char* name, const char* buf, size_t bytes, unsigned* sink) {
    const int blocks = 2048, units = 64;      // 8192 waves x 64 units x 8 KiB = 4 GiB useful per launch
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    rd<MODE><<<blocks, 256>>>(buf, bytes, units, sink);
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0));
        rd<MODE><<<blocks, 256>>>(buf, bytes, units, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    const double useful = (double)blocks * 4 * units * 8192;
    printf("mode %d  %-58s %8.3f ms  %7.1f GB/s useful\n", MODE, name, best, useful / 1e9 / (best * 1e-3));
}

int main() {
    const size_t bytes = (size_t)6 << 30;
    char* buf; unsigned* sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, bytes));
    run<0>("whole lines (1 KiB contiguous per instruction)", buf, bytes, sink);
    run<1>("V^T shape: half lines, both halves by neighbouring waves", buf, bytes, sink);
    run<2>("V^T shape: half lines, the other half never read", buf, bytes, sink);
    run<3>("K shape: 16 rows x 64 B, four instructions per 256-B row", buf, bytes, sink);
    run<4>("V^T in 32-key blocks (contiguous 8 KiB units)", buf, bytes, sink);
    return 0;
}
