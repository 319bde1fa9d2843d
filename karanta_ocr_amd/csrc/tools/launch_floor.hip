// Micro-benchmark: per-kernel cost of a dependent chain of tiny kernels (stream vs hipGraph).
// Build: hipcc -O3 --offload-arch=gfx950 launch_floor.hip -o /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void tiny(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void stream_k(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

int main(int argc, char** argv) {
    int n = 2000;
    hipStream_t s; CK(hipStreamCreate(&s));
    int* d; CK(hipMalloc(&d, 4)); CK(hipMemset(d, 0, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int blocks : {1, 256, 2048}) {
        for (int i = 0; i < 100; ++i) tiny<<<blocks, 256, 0, s>>>(d);
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < n; ++i) tiny<<<blocks, 256, 0, s>>>(d);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("stream  chain blocks=%4d : %.2f us/kernel\n", blocks, ms * 1e3 / n);
        // graph of 200 kernels
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 200; ++i) tiny<<<blocks, 256, 0, s>>>(d);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("graph   chain blocks=%4d : %.2f us/kernel\n", blocks, ms * 1e3 / 2000);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    // streaming copy for the achievable HBM rate
    size_t bytes = (size_t)1 << 30; float4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes));
    for (int it = 0; it < 3; ++it) {
        CK(hipEventRecord(e0, s));
        stream_k<<<256 * 8, 256, 0, s>>>(a, b, bytes / 16);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("copy 1 GiB: %.3f ms -> %.2f TB/s (read+write)\n", ms, 2.0 * bytes / ms / 1e9);
    }
    return 0;
}
