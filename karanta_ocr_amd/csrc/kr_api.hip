// Library plumbing: error string, device info, events, graph capture.
#include <stdarg.h>
#include <string.h>

#include "kr_common.h"

static thread_local char g_err[512] = "";

void kr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

// major * 100 + minor; the major changes with every incompatible change of include/karanta_hip.h (callers check it: _lib.py)
int kr_version(void) { return KR_ABI_VERSION; }

const char* kr_last_error(void) { return g_err; }

int kr_device_info(int device, char* name64, int* compute_units, size_t* total_mem) {
    hipDeviceProp_t p;
    KR_CHECK_HIP(hipGetDeviceProperties(&p, device));
    if (name64) {
        strncpy(name64, p.gcnArchName, 63);
        name64[63] = 0;
    }
    if (compute_units) *compute_units = p.multiProcessorCount;
    if (total_mem) *total_mem = p.totalGlobalMem;
    return KR_OK;
}

int kr_set_device(int device) {
    KR_CHECK_HIP(hipSetDevice(device));
    return KR_OK;
}

int kr_stream_synchronize(kr_stream s) {
    KR_CHECK_HIP(hipStreamSynchronize(kr_hs(s)));
    return KR_OK;
}

// A stream whose kernels run on a SUBSET of the compute units (hipExtStreamCreateWithCUMask): the serving path puts the
// admissions' ViT + prefill launches — long-running 256x256 GEMM / attention workgroups that otherwise fill every CU — on
// `cus_enabled` of the device's CUs, so that the decode graph's short HBM-bound launches on the main stream always find free
// CUs instead of queueing behind them (round 1 measured no gain from a second stream without a mask, for that reason).
// Bit i of the mask = CU i in the runtime's numbering (on a multi-XCD part consecutive bits go round the XCDs, so the first
// n bits take n / 8 CUs of every XCD).
int kr_stream_create_cu_mask(kr_stream* out, int cus_enabled) {
    KR_CHECK_ARG(out && cus_enabled > 0, "kr_stream_create_cu_mask: bad args");
    int dev = 0, cus = 0;
    KR_CHECK_HIP(hipGetDevice(&dev));
    KR_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    KR_CHECK_ARG(cus > 0 && cus_enabled <= cus, "kr_stream_create_cu_mask: %d of %d CUs", cus_enabled, cus);
    const int words = (cus + 31) / 32;
    uint32_t mask[32] = {0};
    KR_CHECK_ARG(words <= 32, "kr_stream_create_cu_mask: %d CUs", cus);
    for (int i = 0; i < cus_enabled; ++i) mask[i >> 5] |= 1u << (i & 31);
    hipStream_t st = nullptr;
    KR_CHECK_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask));
    *out = reinterpret_cast<kr_stream>(st);
    return KR_OK;
}
int kr_stream_create_cu_range(kr_stream* out, int first_cu, int n_cus) {
    KR_CHECK_ARG(out && first_cu >= 0 && n_cus > 0, "kr_stream_create_cu_range: bad args");
    int dev = 0, cus = 0;
    KR_CHECK_HIP(hipGetDevice(&dev));
    KR_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    KR_CHECK_ARG(cus > 0 && first_cu + n_cus <= cus, "kr_stream_create_cu_range: CUs [%d, %d) of %d", first_cu, first_cu + n_cus, cus);
    const int words = (cus + 31) / 32;
    uint32_t mask[32] = {0};
    KR_CHECK_ARG(words <= 32, "kr_stream_create_cu_range: %d CUs", cus);
    for (int i = first_cu; i < first_cu + n_cus; ++i) mask[i >> 5] |= 1u << (i & 31);
    hipStream_t st = nullptr;
    KR_CHECK_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask));
    *out = reinterpret_cast<kr_stream>(st);
    return KR_OK;
}
int kr_stream_destroy(kr_stream s) {
    KR_CHECK_ARG(s, "kr_stream_destroy: the default stream");
    KR_CHECK_HIP(hipStreamDestroy(kr_hs(s)));
    return KR_OK;
}

int kr_event_create(void** ev) {
    KR_CHECK_ARG(ev, "kr_event_create: null");
    hipEvent_t e;
    KR_CHECK_HIP(hipEventCreate(&e));
    *ev = e;
    return KR_OK;
}
int kr_event_destroy(void* ev) {
    KR_CHECK_HIP(hipEventDestroy((hipEvent_t)ev));
    return KR_OK;
}
int kr_event_record(void* ev, kr_stream s) {
    KR_CHECK_HIP(hipEventRecord((hipEvent_t)ev, kr_hs(s)));
    return KR_OK;
}
int kr_stream_wait_event(kr_stream s, void* ev) {
    KR_CHECK_HIP(hipStreamWaitEvent(kr_hs(s), (hipEvent_t)ev, 0));
    return KR_OK;
}
int kr_event_synchronize(void* ev) {
    KR_CHECK_HIP(hipEventSynchronize((hipEvent_t)ev));
    return KR_OK;
}
int kr_event_elapsed_ms(void* start, void* stop, float* ms) {
    KR_CHECK_ARG(ms, "kr_event_elapsed_ms: null");
    KR_CHECK_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return KR_OK;
}

int kr_graph_begin_capture(kr_stream s) {
    KR_CHECK_HIP(hipStreamBeginCapture(kr_hs(s), hipStreamCaptureModeThreadLocal));
    return KR_OK;
}
int kr_graph_end_capture(kr_stream s, void** graph_exec) {
    KR_CHECK_ARG(graph_exec, "kr_graph_end_capture: null");
    hipGraph_t g = nullptr;
    KR_CHECK_HIP(hipStreamEndCapture(kr_hs(s), &g));
    hipGraphExec_t ge = nullptr;
    hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    if (e != hipSuccess) {
        kr_set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e));
        return KR_ERR_HIP;
    }
    *graph_exec = ge;
    return KR_OK;
}
int kr_graph_launch(void* graph_exec, kr_stream s) {
    KR_CHECK_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, kr_hs(s)));
    return KR_OK;
}
int kr_graph_destroy(void* graph_exec) {
    KR_CHECK_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return KR_OK;
}

}  // extern "C"
