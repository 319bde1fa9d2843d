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


class BroadcastError(RuntimeError):
    """The one-time weight broadcast failed on at least one rank; raised on EVERY rank (nobody is left in a collective)."""


def _agree(ok: bool, what: str, detail) -> None:
    """All ranks learn whether `what` worked everywhere (one MIN all-reduce on the host backend) and raise together."""
    import torch
    import torch.distributed as dist

    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)   # CPU tensor: gloo
    if flag.item() < 1.0:
        raise BroadcastError(f"{what} failed on at least one rank" + (f" (this rank: {detail})" if detail else ""))


def broadcast_weights(weights_arena, rank: int, world: int, stream: int = 0, root: int = 0, info: Optional[dict] = None) -> float:
    """One-time broadcast of the packed weight arena (a torch uint8 tensor) from `root`.

    CUDA arena: RCCL through the C-ABI (``kr_comm_*`` + ``kr_bcast_weights``), the 128-byte unique id
    travelling over torch.distributed's object broadcast.  CPU arena (tests, gloo): torch.distributed
    broadcast.  Returns the seconds spent in the broadcast itself; ``info`` (optional dict) receives
    ``rccl_ranks`` = what ncclCommCount reports for the communicator.

    Every step that can fail on one rank alone (unique id on the root, communicator init, the broadcast) is
    followed by an agreement over the host backend, so a failure raises :class:`BroadcastError` on ALL ranks instead
    of leaving the others blocked in a collective.  There is no fallback: a caller that cannot broadcast must stop."""
    import time

    import torch
    import torch.distributed as dist

    if world == 1:
        if info is not None:
            info["rccl_ranks"] = 1
        return 0.0
    if not weights_arena.is_cuda:
        t0 = time.perf_counter()
        dist.broadcast(weights_arena, src=root)
        if info is not None:
            info["rccl_ranks"] = None     # host backend (tests): RCCL was not involved
        return time.perf_counter() - t0
    from ._lib import lib, ptr

    L = lib()
    uid = (C.c_uint8 * 128)()
    err = None
    if rank == root:
        try:
            L.kr_comm_unique_id(uid)
        except Exception as e:      # the others are about to wait for the id: send a sentinel instead of leaving
            err = e
    obj = [None if err else bytes(uid)]
    dist.broadcast_object_list(obj, src=root, device=torch.device("cpu"))
    if obj[0] is None:
        raise BroadcastError(f"kr_comm_unique_id failed on the root rank ({err})")
    uid = (C.c_uint8 * 128).from_buffer_copy(obj[0])
    comm = C.c_void_p()
    try:
        L.kr_comm_init(C.byref(comm), world, rank, uid)
    except Exception as e:  # keep every rank on the same sequence of host collectives
        err = e
    torch.cuda.synchronize()
    try:
        _agree(err is None, "kr_comm_init", err)      # doubles as the pre-broadcast barrier
        n = C.c_int(0)
        t0 = time.perf_counter()
        try:        # anything that can fail on this rank alone stays inside: the agreement below must be reached
            L.kr_comm_count(comm, C.byref(n))
            if info is not None:
                info["rccl_ranks"] = int(n.value)
            t0 = time.perf_counter()
            L.kr_bcast_weights(comm, ptr(weights_arena), weights_arena.numel(), root, stream)
            torch.cuda.synchronize()
        except Exception as e:
            err = e
        dt = time.perf_counter() - t0
        _agree(err is None and n.value == world, "kr_bcast_weights", err or f"ncclCommCount = {n.value}, world = {world}")
    finally:
        if comm.value:
            try:
                L.kr_comm_destroy(comm)
            except Exception:
                pass
    return dt


# ---------------------------------------------------------------------------------------------------------------
# serving group: the N servers of one node (launch.py) form a torch.distributed group ONLY for the start-up weight
# broadcast; after it every server is on its own (no steady-state collective)
# ---------------------------------------------------------------------------------------------------------------
def serving_group_env(environ=None):
    """(rank, world) of this server inside a launch.py group: KARANTA_DP_RANK / KARANTA_DP_WORLD (+ MASTER_ADDR /
    MASTER_PORT for the rendezvous).  A server started by hand has neither: (0, 1)."""
    import os
    env = os.environ if environ is None else environ
    world = int(env.get("KARANTA_DP_WORLD", "1") or 1)
    rank = int(env.get("KARANTA_DP_RANK", "0") or 0)
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"KARANTA_DP_RANK={rank} KARANTA_DP_WORLD={world}")
    return rank, world


def load_or_receive_weights(arena_owner, rank: int, world: int, load_on_root: Callable[[], None], stream: int = 0,
                            log: Callable[[str], None] = lambda _m: None, timeout_s: float = 1800.0) -> dict:
    """Start-up of one server of a group: rank 0 reads the checkpoint into its arena (``load_on_root()``), the others
    only allocate theirs, then the arena travels once over RCCL / xGMI (``broadcast_weights``; gloo for CPU arenas in
    the tests).  This replaces N independent reads of the same checkpoint by N vLLM processes
    (/root/reference/scripts/start_multiple_vllm_servers.sh:283-294).  ``arena_owner`` has ``.arena`` (torch uint8
    tensor or None), ``.allocate()`` and ``.nbytes`` (engine.DeviceWeights).  Returns {"bcast_s", "rccl_ranks",
    "bytes"}.  The process group is created for the broadcast and destroyed after it."""
    import datetime
    import os
    import time

    import torch.distributed as dist

    info: dict = {"bcast_s": 0.0, "rccl_ranks": 1, "bytes": int(arena_owner.nbytes)}
    if world == 1:
        load_on_root()
        return info
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    own_group = not dist.is_initialized()
    if own_group:
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s))
    try:
        err = None
        t0 = time.perf_counter()
        try:
            if rank == 0:
                load_on_root()
            elif arena_owner.arena is None:
                arena_owner.allocate()
        except Exception as e:        # a checkpoint that does not load: every rank stops, nobody waits for the arena
            err = e
        _agree(err is None, "loading the checkpoint" if rank == 0 else "allocating the weight arena", err)
        log(f"rank {rank}/{world}: {'checkpoint read' if rank == 0 else 'arena allocated'} in {time.perf_counter() - t0:.1f}s; "
            f"broadcasting {arena_owner.nbytes / 1e9:.2f} GB")
        info["bcast_s"] = broadcast_weights(arena_owner.arena, rank, world, stream=stream, info=info)
        log(f"rank {rank}/{world}: weights {'sent' if rank == 0 else 'received'} in {info['bcast_s'] * 1e3:.0f} ms "
            f"(rccl_ranks={info['rccl_ranks']})")
    finally:
        if own_group:
            dist.destroy_process_group()
    return info
