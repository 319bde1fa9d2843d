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
