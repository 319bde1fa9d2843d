// One-time weight broadcast over RCCL (xGMI).  The only collective on the path: steady-state
// inference is data parallel with no cross-GPU traffic.
#include <rccl/rccl.h>
#include <string.h>

#include "kr_common.h"

static_assert(sizeof(ncclUniqueId) == KR_UNIQUE_ID_BYTES, "ncclUniqueId size");

#define KR_CHECK_RCCL(expr)                                                                   \
    do {                                                                                      \
        ncclResult_t _r = (expr);                                                             \
        if (_r != ncclSuccess) {                                                              \
            kr_set_error("%s failed: %s (%s:%d)", #expr, ncclGetErrorString(_r), __FILE__, __LINE__); \
            return KR_ERR_RCCL;                                                               \
        }                                                                                     \
    } while (0)

extern "C" {

int kr_comm_unique_id(uint8_t* id128) {
    KR_CHECK_ARG(id128, "kr_comm_unique_id: null");
    ncclUniqueId id;
    KR_CHECK_RCCL(ncclGetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return KR_OK;
}

int kr_comm_init(void** comm, int n_ranks, int rank, const uint8_t* id128) {
    KR_CHECK_ARG(comm && id128 && n_ranks > 0 && rank >= 0 && rank < n_ranks, "kr_comm_init: bad args");
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c;
    KR_CHECK_RCCL(ncclCommInitRank(&c, n_ranks, id, rank));
    *comm = c;
    return KR_OK;
}

int kr_comm_count(void* comm, int* n_ranks) {
    KR_CHECK_ARG(comm && n_ranks, "kr_comm_count: null");
    KR_CHECK_RCCL(ncclCommCount((ncclComm_t)comm, n_ranks));
    return KR_OK;
}

// The RCCL build this library is bound to (ncclGetVersion: major * 10000 + minor * 100 + patch).  bench.py prints it next to the
// broadcast's GB/s: which algorithm RCCL picks for the ncclBroadcast (ring / tree, which xGMI links) is its decision, and a rate far
// below one link's 153 GB/s on the first multi-GPU run is to be read against the version and NCCL_DEBUG=INFO output of that run.
int kr_rccl_version(int* version) {
    KR_CHECK_ARG(version, "kr_rccl_version: null");
    KR_CHECK_RCCL(ncclGetVersion(version));
    return KR_OK;
}

int kr_comm_destroy(void* comm) {
    if (comm) KR_CHECK_RCCL(ncclCommDestroy((ncclComm_t)comm));
    return KR_OK;
}

// Packed weight arena broadcast.  RCCL's broadcast is a pipelined ring/tree over the xGMI links;
// the arena is sent in <= 1 GiB pieces so each call's count fits any internal 32-bit chunking and
// the pieces pipeline on the stream.
int kr_bcast_weights(void* comm, void* buf, size_t bytes, int root, kr_stream s) {
    KR_CHECK_ARG(comm && (buf || bytes == 0), "kr_bcast_weights: null");
    const size_t piece = (size_t)1 << 30;
    for (size_t off = 0; off < bytes; off += piece) {
        const size_t n = bytes - off < piece ? bytes - off : piece;
        KR_CHECK_RCCL(ncclBroadcast((char*)buf + off, (char*)buf + off, n, ncclChar, root, (ncclComm_t)comm, kr_hs(s)));
    }
    return KR_OK;
}

}  // extern "C"
