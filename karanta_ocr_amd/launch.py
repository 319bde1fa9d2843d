"""One engine server per GPU of a node, started together: the ROCm counterpart of the reference's multi-server scripts.

    python -m karanta_ocr_amd.launch --gpus 0,1,2,3 --ports 8000,8001,8002,8003 --model /models/karantaocr-2b \
        [--max-model-len M] [--max-num-seqs S] [--timeout 300] [--log-dir ./vllm_logs] [-- extra server flags]

What the reference does (/root/reference/scripts/start_multiple_vllm_servers.sh): checks the GPUs with ``nvidia-smi``
(:147-173) and the ports (:176-186), starts ``CUDA_VISIBLE_DEVICES=i python -m vllm.entrypoints.openai.api_server
--model M --port P --dtype bfloat16 ...`` once per GPU with a log file each (:271-310), polls ``/health`` until every
server answers or a timeout passes (:240-268), and leaves a summary plus a cleanup script (:444-453);
``bulk_processing/scripts/start_multiple_celery_workers.sh:254-297`` then binds workers to those ports and
``bulk_processing/utils/gpu_router.py`` sends each task to the shortest ``gpu_queue_{port}``.

Here (same command-line shape, MI355X semantics):

* pre-flight with ``rocm-smi`` / ``amd-smi`` when one is installed (a missing tool is a warning: the servers themselves
  fail loudly without a GPU), free-port check by binding;
* one ``python -m karanta_ocr_amd.cli serve`` per GPU with ``HIP_VISIBLE_DEVICES=i`` (``ROCR_VISIBLE_DEVICES`` is left
  alone), all of them one *serving group*: **rank 0 reads the checkpoint once and the packed weight arena is broadcast
  to the other GPUs over RCCL / xGMI** (dp.load_or_receive_weights -> kr_bcast_weights) instead of N processes reading
  the same files; after start-up the servers share nothing (no steady-state collective);
* ``/health`` polling with the reference's timeout semantics; if one server dies or times out, the group is torn down
  (half a group cannot have received its weights);
* SIGINT / SIGTERM: every child is terminated by PID and reaped — the reference's generated cleanup script, built in;
* :func:`router` gives the :class:`dp.LeastLoadedRouter` over the group's ports (``gpu_queue_{port}`` names).
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import signal
import socket
import subprocess
import sys
import time
import urllib.request
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence

from .dp import LeastLoadedRouter

DEFAULT_SERVER_CMD = [sys.executable, "-m", "karanta_ocr_amd.cli"]


def _ints(csv: str, what: str) -> List[int]:
    try:
        out = [int(x) for x in csv.split(",") if x.strip() != ""]
    except ValueError:
        raise ValueError(f"{what} must be a comma-separated list of integers, got {csv!r}") from None
    if not out:
        raise ValueError(f"{what} is empty")
    if len(set(out)) != len(out):
        raise ValueError(f"{what} has duplicates: {csv}")
    return out


def port_is_free(port: int, host: str = "127.0.0.1") -> bool:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        try:
            s.bind((host, port))
        except OSError:
            return False
    return True


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def gpu_preflight(gpus: Sequence[int], log: Callable[[str], None], run=subprocess.run) -> Dict[int, Optional[float]]:
    """GPU id -> utilisation in percent (None when unknown).  ``rocm-smi --showuse --json`` first, ``amd-smi metric
    --usage --json`` second; a GPU the tool does not list is an error, a busy one (> 90 %) a warning, no tool at all a
    warning (start_multiple_vllm_servers.sh:147-173 does the same with nvidia-smi, where a missing tool is fatal —
    here the servers refuse to start without a device anyway)."""
    use: Dict[int, Optional[float]] = {g: None for g in gpus}
    listed: Optional[set] = None
    if shutil.which("rocm-smi"):
        try:
            r = run(["rocm-smi", "--showuse", "--json"], capture_output=True, text=True, timeout=30)
            cards = json.loads(r.stdout or "{}")
            listed = set()
            for name, row in cards.items():
                if not name.startswith("card"):
                    continue
                idx = int(name[4:])
                listed.add(idx)
                for k, v in row.items():
                    if "GPU use" in k and idx in use:
                        try:
                            use[idx] = float(v)
                        except (TypeError, ValueError):
                            pass
        except Exception as e:      # a broken tool is not a reason to refuse the launch
            log(f"[WARNING] rocm-smi failed ({e}); skipping the GPU pre-flight")
            listed = None
    elif shutil.which("amd-smi"):
        try:
            r = run(["amd-smi", "metric", "--usage", "--json"], capture_output=True, text=True, timeout=30)
            rows = json.loads(r.stdout or "[]")
            rows = rows.get("gpu_data", rows) if isinstance(rows, dict) else rows
            listed = set()
            for row in rows:
                idx = int(row.get("gpu", -1))
                listed.add(idx)
                u = (row.get("usage") or {}).get("gfx_activity")
                u = u.get("value") if isinstance(u, dict) else u
                if idx in use and isinstance(u, (int, float)):
                    use[idx] = float(u)
        except Exception as e:
            log(f"[WARNING] amd-smi failed ({e}); skipping the GPU pre-flight")
            listed = None
    else:
        log("[WARNING] neither rocm-smi nor amd-smi found; skipping the GPU pre-flight")
    if listed is not None:
        missing = [g for g in gpus if g not in listed]
        if missing:
            raise RuntimeError(f"GPU(s) {missing} not found (the SMI tool lists {sorted(listed)})")
    for g, u in use.items():
        if u is not None and u > 90:
            log(f"[WARNING] GPU {g} is heavily utilised ({u:.0f} %)")
        else:
            log(f"[INFO] GPU {g} is available" + (f" (utilisation {u:.0f} %)" if u is not None else ""))
    return use


def health(port: int, host: str = "127.0.0.1", timeout: float = 2.0) -> bool:
    try:
        with urllib.request.urlopen(f"http://{host}:{port}/health", timeout=timeout) as r:
            return r.status == 200
    except Exception:
        return False


@dataclass
class ServerGroup:
    """The running servers of one node: ports, GPU ids, child processes (by rank) and their log files."""
    gpus: List[int]
    ports: List[int]
    procs: List[subprocess.Popen] = field(default_factory=list)
    logs: List[str] = field(default_factory=list)

    def router(self) -> LeastLoadedRouter:
        return router(self.ports)

    def alive(self) -> List[bool]:
        return [p.poll() is None for p in self.procs]

    def stop(self, grace_s: float = 10.0) -> List[Optional[int]]:
        """SIGTERM to every child (exact PIDs), SIGKILL after the grace period; returns the exit codes."""
        for p in self.procs:
            if p.poll() is None:
                try:
                    p.send_signal(signal.SIGTERM)
                except ProcessLookupError:
                    pass
        deadline = time.time() + grace_s
        for p in self.procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        return [p.returncode for p in self.procs]


def router(ports: Sequence[int], queue_len=None) -> LeastLoadedRouter:
    """Least-outstanding-work routing over the group's ports (gpu_router.py:10-20 semantics, ``gpu_queue_{port}``)."""
    return LeastLoadedRouter(list(ports), queue_len)


def start_servers(model: str, gpus: Sequence[int], ports: Sequence[int], extra: Sequence[str] = (), log_dir: str = "./vllm_logs",
                  timeout_s: float = 300.0, poll_s: float = 1.0, server_cmd: Optional[Sequence[str]] = None,
                  log: Callable[[str], None] = print, preflight: bool = True, broadcast: bool = True,
                  env: Optional[Dict[str, str]] = None, should_stop: Optional[Callable[[], bool]] = None) -> ServerGroup:
    """Start the group and return once every server answers ``/health`` (raises, with everything stopped, otherwise).
    should_stop: polled once per health round; when it turns true (SIGINT / SIGTERM during start-up) the children —
    which run in their own sessions, so a terminal's Ctrl-C does not reach them — are stopped and InterruptedError raised."""
    gpus, ports = list(gpus), list(ports)
    if len(gpus) != len(ports):
        raise ValueError(f"Number of GPUs ({len(gpus)}) must match number of ports ({len(ports)})")
    if not model:
        raise ValueError("Model name is required")
    if preflight:
        gpu_preflight(gpus, log)
    busy = [p for p in ports if not port_is_free(p)]
    if busy:
        raise RuntimeError(f"Port(s) {busy} already in use")
    os.makedirs(log_dir, exist_ok=True)
    n = len(gpus)
    master_port = free_port()
    group = ServerGroup(gpus, ports)
    cmd0 = list(server_cmd or DEFAULT_SERVER_CMD)
    base_env = dict(os.environ if env is None else env)
    # the RCCL weight broadcast shares device memory between the servers: dmabuf IPC (the legacy mode fails in hipIpcGetMemHandle)
    base_env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for rank, (g, port) in enumerate(zip(gpus, ports)):
        e = dict(base_env, HIP_VISIBLE_DEVICES=str(g), OMP_NUM_THREADS=base_env.get("OMP_NUM_THREADS", "1"))
        if broadcast and n > 1:
            e.update(KARANTA_DP_RANK=str(rank), KARANTA_DP_WORLD=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(master_port))
        else:
            e.pop("KARANTA_DP_RANK", None)
            e.pop("KARANTA_DP_WORLD", None)
        path = os.path.join(log_dir, f"vllm_gpu_{g}_port_{port}.log")      # the reference's log file names
        cmd = cmd0 + ["serve", model, "--port", str(port), *extra]
        log(f"[INFO] Launching server on GPU {g}, port {port}: HIP_VISIBLE_DEVICES={g} {' '.join(cmd)}  (log: {path})")
        with open(path, "ab") as fh:
            group.procs.append(subprocess.Popen(cmd, env=e, stdout=fh, stderr=subprocess.STDOUT, start_new_session=True))
        group.logs.append(path)
    t0, ready, last_note = time.time(), [False] * n, 0.0
    try:
        while not all(ready):
            for i, p in enumerate(group.procs):
                if p.poll() is not None:
                    raise RuntimeError(f"server on GPU {gpus[i]} (port {ports[i]}) exited with code {p.returncode}; see {group.logs[i]}")
                if not ready[i] and health(ports[i]):
                    ready[i] = True
                    log(f"[SUCCESS] server on GPU {gpus[i]} (port {ports[i]}) is ready after {time.time() - t0:.0f}s")
            if all(ready):
                break
            if should_stop is not None and should_stop():
                raise InterruptedError("stop requested while the servers were starting")
            if time.time() - t0 > timeout_s:
                late = [f"GPU {gpus[i]}:{ports[i]}" for i in range(n) if not ready[i]]
                raise TimeoutError(f"Timeout waiting for {', '.join(late)} after {timeout_s:.0f}s")
            if time.time() - last_note >= 10:
                last_note = time.time()
                log(f"[INFO] waiting for {n - sum(ready)} server(s)... ({time.time() - t0:.0f}s elapsed)")
            time.sleep(poll_s)
    except BaseException:
        group.stop()
        raise
    summary = {"model": model, "servers": [{"gpu": g, "port": p, "pid": pr.pid, "log": lg, "url": f"http://localhost:{p}/v1"}
                                           for g, p, pr, lg in zip(gpus, ports, group.procs, group.logs)],
               "weights": "rank 0 read the checkpoint; RCCL broadcast to the others" if (broadcast and n > 1) else "each server read the checkpoint"}
    with open(os.path.join(log_dir, "server_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    return group


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="karanta_ocr_amd.launch", description=__doc__.split("\n\n")[0])
    ap.add_argument("--gpus", required=True, help='comma-separated GPU ids, e.g. "0,1,2,3"')
    ap.add_argument("--ports", required=True, help='comma-separated ports, one per GPU, e.g. "8000,8001,8002,8003"')
    ap.add_argument("--model", required=True, help="model directory (config.json, *.safetensors, tokenizer.json)")
    ap.add_argument("--max-model-len", type=int, default=None)
    ap.add_argument("--max-num-seqs", type=int, default=None)
    ap.add_argument("--served-model-name", default=None)
    ap.add_argument("--dtype", default="bfloat16")
    ap.add_argument("--trust-remote-code", action="store_true")
    ap.add_argument("--tensor-parallel-size", type=int, default=1)
    ap.add_argument("--timeout", type=float, default=300.0, help="health check timeout in seconds")
    ap.add_argument("--log-dir", default="./vllm_logs")
    ap.add_argument("--no-broadcast", action="store_true", help="every server reads the checkpoint itself (no RCCL broadcast)")
    ap.add_argument("--no-preflight", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    return ap


def main(argv: Optional[List[str]] = None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    extra: List[str] = []
    if "--" in argv:
        k = argv.index("--")
        argv, extra = argv[:k], argv[k + 1:]
    args = build_parser().parse_args(argv)
    log = lambda m: print(m, file=sys.stderr, flush=True)
    if args.tensor_parallel_size != 1:
        log("[ERROR] one process serves one GPU: tensor parallel size must be 1")
        return 2
    try:
        gpus, ports = _ints(args.gpus, "--gpus"), _ints(args.ports, "--ports")
    except ValueError as e:
        log(f"[ERROR] {e}")
        return 2
    passthrough = ["--dtype", args.dtype]
    if args.max_model_len:
        passthrough += ["--max-model-len", str(args.max_model_len)]
    if args.max_num_seqs:
        passthrough += ["--max-num-seqs", str(args.max_num_seqs)]
    if args.served_model_name:
        passthrough += ["--served-model-name", args.served_model_name]
    if args.trust_remote_code:
        passthrough += ["--trust-remote-code"]
    stop = {"now": False}

    def on_signal(*_):
        stop["now"] = True

    for sig in (signal.SIGTERM, signal.SIGINT):
        try:
            signal.signal(sig, on_signal)
        except ValueError:
            pass
    try:
        group = start_servers(args.model, gpus, ports, passthrough + extra, args.log_dir, args.timeout, log=log,
                              preflight=not args.no_preflight, broadcast=not args.no_broadcast,
                              should_stop=lambda: stop["now"])
    except InterruptedError as e:
        log(f"[INFO] {e}: servers stopped")
        return 130
    except Exception as e:
        log(f"[ERROR] {e}")
        return 1
    log("=== server launch summary ===")
    for g, p in zip(gpus, ports):
        log(f"[SUCCESS]   GPU {g}: http://localhost:{p}/v1   (queue gpu_queue_{p})")
    log("all servers are ready; SIGINT / SIGTERM stops them")
    rc = 0
    while not stop["now"]:
        dead = [i for i, ok in enumerate(group.alive()) if not ok]
        if dead:
            for i in dead:
                log(f"[ERROR] server on GPU {gpus[i]} (port {ports[i]}) exited with code {group.procs[i].returncode}; see {group.logs[i]}")
            rc = 1
            break
        time.sleep(1.0)
    group.stop()
    return rc


if __name__ == "__main__":
    sys.exit(main())
