"""Multi-GPU path on CPU: page sharding, least-loaded routing, and the weight broadcast through
torch.distributed (gloo, world_size 2) — the N > 1 host logic of bench.py / dp.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from karanta_ocr_amd.dp import LeastLoadedRouter, broadcast_weights, shard_pages


def test_shard_pages_partition():
    for n in (0, 1, 7, 8, 10000):
        for world in (1, 2, 3, 8):
            shards = [shard_pages(n, world, r) for r in range(world)]
            flat = [i for s in shards for i in s]
            assert flat == list(range(n))
            assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
    with pytest.raises(ValueError):
        shard_pages(4, 2, 2)


def test_least_loaded_router_matches_reference_semantics():
    r = LeastLoadedRouter([8000, 8001, 8002])
    assert r.gpu_queues == ["gpu_queue_8000", "gpu_queue_8001", "gpu_queue_8002"]
    assert r.get_best_queue() == "gpu_queue_8000"           # first queue wins ties
    r.submit("gpu_queue_8000"); r.submit("gpu_queue_8001")
    assert r.get_best_queue() == "gpu_queue_8002"
    r.submit("gpu_queue_8002"); r.done("gpu_queue_8001")
    assert r.get_best_queue() == "gpu_queue_8001"
    lens = {"gpu_queue_1": 5, "gpu_queue_2": 2}
    assert LeastLoadedRouter([1, 2], queue_len=lens.__getitem__).get_best_queue() == "gpu_queue_2"


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from karanta_ocr_amd.config import CONFIGS
        from karanta_ocr_amd.engine import DeviceWeights
        from karanta_ocr_amd.weights import random_weights
        import unittest.mock as mock

        cfg = CONFIGS["tiny"]
        dw = DeviceWeights(cfg, torch.device("cpu"))
        dw.arena = torch.zeros(dw.nbytes, dtype=torch.uint8)
        if rank == 0:
            with mock.patch("torch.cuda.synchronize"):
                dw.load(random_weights(cfg, 5))
        dt = broadcast_weights(dw.arena, rank, world)
        # every rank now holds rank 0's arena; pages are sharded, nothing else is exchanged
        digest = int(dw.arena.to(torch.int64).sum().item())
        pages = list(shard_pages(11, world, rank))
        out = [None] * world
        dist.all_gather_object(out, (digest, pages, dt >= 0))
        if rank == 0:
            q.put(out)
    finally:
        dist.destroy_process_group()


def test_weight_broadcast_and_sharding_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    out = q.get(timeout=120)
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (d0, p0, ok0), (d1, p1, ok1) = out
    assert d0 == d1 and d0 > 0 and ok0 and ok1
    assert p0 + p1 == list(range(11))
