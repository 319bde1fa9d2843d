"""`python -m karanta_ocr_amd.bench_corpus --gpus N` rehearsed on the CPU: BASELINE config 4 as ONE command — N servers through
launch.py (here tests/fake_group_server.py stands in for `karanta_ocr_amd.cli`: same command line, same serving-group start-up,
the weight arena broadcast over gloo), the submit loop of /root/reference/bulk_processing/main.py:30-60 with least-loaded
routing (/root/reference/bulk_processing/utils/gpu_router.py:10-20), worker threads bound to a port
(/root/reference/bulk_processing/scripts/start_multiple_celery_workers.sh:254-297), one merged JSON line."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_node_mode_two_fake_servers(tmp_path):
    env = dict(os.environ, KARANTA_TEST_OUT=str(tmp_path), PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "karanta_ocr_amd.bench_corpus", "--gpus", "2", "--model-dir", "/models/karantaocr", "--model", "tiny",
           "--pages", "14", "--workers", "3", "--page", "56", "--distinct", "2", "--t-min", "2", "--t-max", "4", "--no-preflight",
           "--timeout", "120", "--log-dir", str(tmp_path / "logs"),
           "--server-cmd", f"{sys.executable} {os.path.join(ROOT, 'tests', 'fake_group_server.py')}"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["n_errors"] == 0 and out["server_exit_codes"] == [0, 0]
    assert sum(out["per_gpu_requests"]) == 14 and min(out["per_gpu_requests"]) >= 3          # least-loaded routing uses both servers
    assert 14 * 2 <= out["completion_tokens"] <= 14 * 3 and out["pages_per_s"] > 0
    assert out["latency_s"]["p50"] <= out["latency_s"]["p95"] <= out["latency_s"]["p99"] <= out["latency_s"]["max"]
    assert abs(sum(out["per_gpu_pages_per_s"]) - out["pages_per_s"]) < 0.05 * out["pages_per_s"] + 0.01
    assert "rccl_ranks" in out and "RCCL broadcast" in out["weights"]
    # both servers were one serving group (rank, world, the GPU they were pinned to, the same arena) and got the reference's flags
    rows = sorted(open(tmp_path / f).read().split() for f in os.listdir(tmp_path) if f.startswith("server_"))
    assert [r_[:3] for r_ in rows] == [["0", "2", "0"], ["1", "2", "1"]] and rows[0][3] == rows[1][3]
    assert "--max-num-seqs" in rows[0] and "--greedy" in rows[0] and rows[0][4] == "/models/karantaocr"


def test_run_corpus_routes_to_the_least_loaded_port():
    """run_corpus against two in-process servers of different speed: the slower one gets fewer requests, every request is
    answered once, latencies are measured from submission."""
    import threading
    import time
    from types import SimpleNamespace
    from karanta_ocr_amd import bench_corpus as BC
    from karanta_ocr_amd import serving as S
    from karanta_ocr_amd.config import CONFIGS

    cfg = CONFIGS["tiny"]

    def make(delay):
        class Engine:
            B = 2

            def generate(self, pages, max_new_tokens, **kw):
                time.sleep(delay)
                toks = [np.asarray(list(b"ok") + [cfg.eos_token_ids[0]], np.int64) for _ in pages]
                return SimpleNamespace(tokens=toks, finish_reasons=["stop"] * len(pages), prompt_tokens=[len(p.input_ids) for p in pages])
        Engine.cfg = cfg
        return S.LocalServer(Engine(), S.ChatFrontend(cfg, S.ByteTokenizer(cfg)), log=lambda *a: None)

    fast, slow = make(0.002), make(0.05)
    S.register_local_server(18001, fast)
    S.register_local_server(18002, slow)
    try:
        req = lambda i: dict(messages=[{"role": "user", "content": f"page {i}"}], max_tokens=4, temperature=0.0)
        wall, lat, toks, errors, served = BC.run_corpus(60, req, [18001, 18002], 2)
    finally:
        S.unregister_local_server(18001)
        S.unregister_local_server(18002)
        fast.close(), slow.close()
    assert not errors and sum(served.values()) == 60 and served[18001] > served[18002] > 0
    assert all(2 <= t <= 3 for t in toks) and min(lat) > 0 and max(lat) < wall + 1e-3
