"""`python bench.py --gpus N` starts its own ranks: fresh child processes with RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_*, rank 0's JSON line relayed, non-zero exit if any rank fails.  Exercised on CPU through the
bench's --dry-run control-plane rehearsal (gloo, world_size 2) — no engine, no GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    e.update(env)
    return subprocess.run([sys.executable, BENCH, *args], env=e, capture_output=True, text=True, timeout=300)


def test_bench_self_launches_two_ranks_and_relays_rank0_json():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "0", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                         # exactly ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["dry_run"] is True and out["value"] is None     # cannot be mistaken for a measurement
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["pages_per_rank"] == [8, 8]
    assert out["ms_per_step"] >= 20.0 * 0.9                    # the slower rank (2 x 10 ms per step) sets the time: MAX over ranks


def test_bench_single_rank_dry_run_needs_no_launcher():
    r = _run(["--steps", "1", "--dry-run"])
    assert r.returncode == 0 and json.loads(r.stdout.strip())["n_gpus"] == 1


def test_a_failing_rank_fails_the_run():
    r = _run(["--gpus", "2", "--steps", "1", "--dry-run"], KARANTA_BENCH_DRY_FAIL_RANK="1")
    assert r.returncode != 0 and "INVALID" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.strip().startswith("{")]   # no JSON line from a broken run


def test_world_size_mismatch_is_refused():
    r = _run(["--gpus", "2", "--dry-run"], WORLD_SIZE="1", RANK="0")
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_parity_block_compares_a_teacher_forced_engine_run_with_the_oracle_run():
    """bench.py's `parity` entry (BASELINE.md's parity statement on every driver record): the engine is re-run TEACHER-FORCED
    with the oracle's tokens, every step's logits compare, argmax equality is required on the decisive steps only.  A fake
    engine stands in for the GPU: logits = the oracle's plus a known perturbation."""
    import types

    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    rng = np.random.default_rng(3)
    steps, V = 9, 400
    o_log = rng.standard_normal((steps, V)).astype(np.float32)
    o_log[np.arange(steps), rng.integers(0, V, steps)] += 6.0          # a clear winner at every step
    o_log[4] = 0.0
    o_log[4, 7], o_log[4, 9] = 5.0, 4.9999                             # ... except a near-tie at step 4
    o_tok = o_log.argmax(-1)
    seen = {}

    class FakeEngine:
        def generate(self, pages, n, ignore_eos, return_logits, force_tokens):
            seen.update(n=n, forced=np.asarray(force_tokens))
            lg = o_log + 0.01
            lg[4, 9] += 0.05                                           # the near-tie flips: allowed
            return types.SimpleNamespace(logits=lg[None], tokens=[lg.argmax(-1)])

    out = bench.parity_block(FakeEngine(), object(), {"tokens": o_tok, "logits": o_log})
    assert seen["n"] == steps and seen["forced"].shape == (1, steps - 1) and (seen["forced"][0] == o_tok[:-1]).all()
    assert out["pass"] is True and out["argmax_equal"] == "8/9" and out["decisive_steps"] == 8 and out["decisive_argmax_equal"] == "8/8"
    assert 0.05 < out["max_abs_dlogit"] < 0.07 and out["tol_rel"] == bench.PARITY_TOL_REL

    class WrongEngine(FakeEngine):
        def generate(self, *a, **k):
            r = FakeEngine.generate(self, *a, **k)
            r.logits[0, 2] = -r.logits[0, 2]                           # a decisive step goes wrong
            r.tokens = [r.logits[0].argmax(-1)]
            return r

    assert bench.parity_block(WrongEngine(), object(), {"tokens": o_tok, "logits": o_log})["pass"] is False


def test_committed_pmc_traffic_is_quoted_only_on_the_kernel_source_it_was_measured_on(tmp_path, monkeypatch):
    """`roofline.traffic` comes from a committed rocprofv3 --pmc pass; the file carries the hash of the
    decode kernels' source and bench.py drops the figure (with the reason) when the tree's kernels differ."""
    sys.path.insert(0, ROOT)
    import bench
    got, src = bench.pmc_traffic()
    newest = next(n for n in ("r04_pmc_traffic.json", "r03_pmc_traffic.json") if os.path.exists(os.path.join(ROOT, "profiles", n)))
    committed = json.load(open(os.path.join(ROOT, "profiles", newest)))
    if committed["kernel_source_sha16"] == bench.kernel_source_sha16():
        assert got and 0.9 < got / 55221248 < 1.2 and src == "profiles/" + newest   # ~ the algorithmic bytes
        assert os.path.exists(os.path.join(ROOT, committed["raw_csv"])), "the raw counter CSV is kept beside the summary"
    else:
        assert got is None and "another kr_decode.hip" in src
    monkeypatch.setattr(bench, "kernel_source_sha16", lambda: "0" * 16)           # the kernels changed since the pass
    got, why = bench.pmc_traffic()
    assert got is None and "not quoted" in why


class _FakeArenaEngine:
    """What bench.distribute_weights touches of an engine: the weight arena, its size, the stream, load_weights."""

    def __init__(self, fill):
        import torch
        self.w = type("W", (), {})()
        self.w.arena = torch.full((1 << 12,), fill, dtype=torch.uint8)
        self.w.nbytes = self.w.arena.numel()
        self.s = 0
        self.loaded = None

    def load_weights(self, w):
        self.loaded = w
        self.w.arena.fill_(int(w["fill"]))


def test_failed_rccl_broadcast_is_reported_and_every_rank_generates_the_seeded_weights():
    """N > 1 bench: a broadcast that raises (on every rank: dp.BroadcastError) leaves rccl_error in the record, the other ranks
    generate the same seeded weights, the digests are still compared; --strict-rccl re-raises; a good broadcast reports its ranks."""
    import pytest
    sys.path.insert(0, ROOT)
    import bench
    from karanta_ocr_amd.dp import BroadcastError

    def failing(arena, rank, world, stream=0, info=None):
        raise BroadcastError("kr_comm_init failed on at least one rank")

    eng = _FakeArenaEngine(fill=0)          # a non-root rank: zeros until the weights arrive
    root_digest = []

    def gather(x):                          # the root holds the seeded weights (fill 7)
        root = _FakeArenaEngine(fill=7)
        root_digest.append(int(root.w.arena[:: max(1, root.w.arena.numel() // (1 << 22))].to(__import__("torch").int64).sum().item()))
        return [root_digest[-1], x]

    s, ranks, err = bench.distribute_weights(eng, 1, 2, failing, gather, lambda: {"fill": 7}, strict=False)
    assert s is None and ranks is None and "kr_comm_init" in err and eng.loaded == {"fill": 7}
    with pytest.raises(BroadcastError):
        bench.distribute_weights(_FakeArenaEngine(0), 1, 2, failing, gather, lambda: {"fill": 7}, strict=True)
    with pytest.raises(SystemExit, match="digests differ"):     # a rank that ends up with other bytes stops the run
        bench.distribute_weights(_FakeArenaEngine(0), 1, 2, failing, gather, lambda: {"fill": 9}, strict=False)

    def good(arena, rank, world, stream=0, info=None):
        arena.fill_(7)
        info["rccl_ranks"] = world
        return 0.25

    eng2 = _FakeArenaEngine(fill=0)
    s, ranks, err = bench.distribute_weights(eng2, 1, 2, good, gather, lambda: {"fill": 7}, strict=False)
    assert (s, ranks, err) == (0.25, 2, None) and eng2.loaded is None
