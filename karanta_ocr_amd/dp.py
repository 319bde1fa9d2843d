"""Data parallelism over the GPUs of one node: one engine process per GPU, pages sharded, weights
broadcast once over RCCL, no steady-state collective.

Reference topology being replaced: one vLLM server per GPU
(/root/reference/scripts/start_multiple_vllm_servers.sh:283, :444-453), one Celery queue per server
``gpu_queue_{port}`` with least-queue-length routing (/root/reference/bulk_processing/utils/gpu_router.py:
5-30), workers bound to a port by hostname (bulk_processing/workers/vllm_client.py:359-386).
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, List, Optional, Sequence


def shard_pages(n_pages: int, world: int, rank: int) -> range:
    """Contiguous, balanced shard of page indices for `rank` (sizes differ by at most one)."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} not in [0, {world})")
    base, extra = divmod(n_pages, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


class LeastLoadedRouter:
    """``GPURouter.get_best_queue`` semantics (gpu_router.py:10-20): the queue with the fewest
    outstanding items wins, first one on ties; queue names keep the reference's ``gpu_queue_{port}``."""

    def __init__(self, ports: Sequence[int], queue_len: Optional[Callable[[str], int]] = None):
        self.gpu_queues = [f"gpu_queue_{p}" for p in ports]
        self._outstanding: Dict[str, int] = {q: 0 for q in self.gpu_queues}
        self._len = queue_len or (lambda q: self._outstanding[q])

    def get_best_queue(self) -> str:
        best, best_len = None, float("inf")
        for q in self.gpu_queues:
            n = self._len(q)
            if n < best_len:
                best, best_len = q, n
        return best

    def submit(self, queue: str) -> None:
        self._outstanding[queue] += 1

    def done(self, queue: str) -> None:
        self._outstanding[queue] -= 1


def broadcast_weights(weights_arena, rank: int, world: int, stream: int = 0, root: int = 0) -> float:
    """One-time broadcast of the packed weight arena (a torch uint8 tensor) from `root`.

    CUDA arena: RCCL through the C-ABI (``kr_comm_*`` + ``kr_bcast_weights``), the 128-byte unique id
    travelling over torch.distributed's object broadcast.  CPU arena (tests, gloo): torch.distributed
    broadcast.  Returns the seconds spent in the broadcast itself."""
    import time

    import torch
    import torch.distributed as dist

    if world == 1:
        return 0.0
    if not weights_arena.is_cuda:
        t0 = time.perf_counter()
        dist.broadcast(weights_arena, src=root)
        return time.perf_counter() - t0
    from ._lib import lib, ptr

    L = lib()
    uid = (C.c_uint8 * 128)()
    if rank == root:
        L.kr_comm_unique_id(uid)
    obj = [bytes(uid)]
    dist.broadcast_object_list(obj, src=root, device=torch.device("cpu"))
    uid = (C.c_uint8 * 128).from_buffer_copy(obj[0])
    comm = C.c_void_p()
    err = None
    try:
        L.kr_comm_init(C.byref(comm), world, rank, uid)
    except Exception as e:  # keep every rank on the same sequence of host collectives
        err = e
    ok = torch.tensor([0.0 if err else 1.0])
    torch.cuda.synchronize()
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)  # CPU tensor: gloo; doubles as the pre-broadcast barrier
    if ok.item() < 1.0:
        if comm.value:
            L.kr_comm_destroy(comm)
        raise RuntimeError(f"kr_comm_init failed on at least one rank ({err})")
    t0 = time.perf_counter()
    L.kr_bcast_weights(comm, ptr(weights_arena), weights_arena.numel(), root, stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    L.kr_comm_destroy(comm)
    return dt
