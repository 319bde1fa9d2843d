"""Multi-GPU serving launcher on CPU: two servers with fake engines form a serving group over gloo (rank 0 "reads the
checkpoint", the arena is broadcast), answer /health, are routed by least outstanding work and shut down cleanly.
Mirrors /root/reference/scripts/start_multiple_vllm_servers.sh (:147-186 checks, :240-268 health polling, :271-310
one server per GPU with a log file each, :444-453 summary)."""
import json
import os
import socket
import sys
import time
import urllib.request

import pytest

from karanta_ocr_amd import launch

STUB = [sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_group_server.py")]


def _env(tmp_path, **extra):
    return dict(os.environ, KARANTA_TEST_OUT=str(tmp_path), **extra)


def test_two_servers_share_one_checkpoint_read_and_serve(tmp_path):
    ports = [launch.free_port(), launch.free_port()]
    logs = []
    group = launch.start_servers("/models/m", [0, 1], ports, extra=["--max-model-len", "4096"], log_dir=str(tmp_path / "logs"),
                                 timeout_s=120, poll_s=0.2, server_cmd=STUB, log=logs.append, preflight=False, env=_env(tmp_path))
    try:
        assert all(group.alive()) and all(launch.health(p) for p in ports)
        rows = [open(tmp_path / f"server_{p}.txt").read().split() for p in ports]
        assert [r[0] for r in rows] == ["0", "1"] and {r[1] for r in rows} == {"2"}
        assert [r[2] for r in rows] == ["0", "1"]                       # HIP_VISIBLE_DEVICES = the GPU id of its rank
        assert rows[0][3] == rows[1][3] and int(rows[0][3]) > 0           # rank 1 RECEIVED rank 0's arena
        assert rows[0][4] == "/models/m" and "--max-model-len 4096" in " ".join(rows[0][5:])
        # the OpenAI surface answers on both ports
        for p in ports:
            with urllib.request.urlopen(f"http://127.0.0.1:{p}/v1/models", timeout=5) as r:
                assert json.load(r)["data"][0]["id"]
        summary = json.load(open(tmp_path / "logs" / "server_summary.json"))
        assert [s["port"] for s in summary["servers"]] == ports and "RCCL broadcast" in summary["weights"]
        assert os.path.basename(group.logs[0]) == f"vllm_gpu_0_port_{ports[0]}.log"
        # routing: least outstanding work, first queue on ties, the reference's queue names
        r = group.router()
        assert r.get_best_queue() == f"gpu_queue_{ports[0]}"
        r.submit(r.get_best_queue())
        assert r.get_best_queue() == f"gpu_queue_{ports[1]}"
    finally:
        codes = group.stop()
    assert codes == [0, 0] and not any(group.alive())
    assert not any(launch.health(p, timeout=0.5) for p in ports)
    assert any("is ready" in l for l in logs)


def test_a_dying_server_tears_the_group_down(tmp_path):
    ports = [launch.free_port(), launch.free_port()]
    with pytest.raises(RuntimeError, match="exited with code 3"):
        launch.start_servers("/models/m", [0, 1], ports, log_dir=str(tmp_path / "logs"), timeout_s=60, poll_s=0.2,
                             server_cmd=STUB, log=lambda _m: None, preflight=False, env=_env(tmp_path, STUB_FAIL_RANK="1"))
    time.sleep(0.2)
    assert not any(launch.health(p, timeout=0.5) for p in ports)        # rank 0 was stopped too


def test_launcher_argument_checks(tmp_path):
    with pytest.raises(ValueError, match="must match"):
        launch.start_servers("/m", [0, 1], [8000], preflight=False, log=lambda _m: None)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); s.listen(1)
    busy = s.getsockname()[1]
    try:
        assert not launch.port_is_free(busy)
        with pytest.raises(RuntimeError, match="already in use"):
            launch.start_servers("/m", [0], [busy], preflight=False, log=lambda _m: None, log_dir=str(tmp_path))
    finally:
        s.close()
    assert launch.main(["--gpus", "0,x", "--ports", "1", "--model", "/m"]) == 2
    assert launch.main(["--gpus", "0", "--ports", "1", "--model", "/m", "--tensor-parallel-size", "2"]) == 2


def test_gpu_preflight_parses_rocm_smi(monkeypatch):
    from types import SimpleNamespace
    monkeypatch.setattr(launch.shutil, "which", lambda name: "/opt/rocm/bin/rocm-smi" if name == "rocm-smi" else None)
    out = json.dumps({"card0": {"GPU use (%)": "3"}, "card1": {"GPU use (%)": "97"}})
    notes = []
    use = launch.gpu_preflight([0, 1], notes.append, run=lambda *a, **k: SimpleNamespace(stdout=out))
    assert use == {0: 3.0, 1: 97.0} and any("heavily utilised" in n for n in notes)
    with pytest.raises(RuntimeError, match="not found"):
        launch.gpu_preflight([0, 5], notes.append, run=lambda *a, **k: SimpleNamespace(stdout=out))
    monkeypatch.setattr(launch.shutil, "which", lambda name: None)
    assert launch.gpu_preflight([0], notes.append) == {0: None}          # no tool: a warning, not an error
