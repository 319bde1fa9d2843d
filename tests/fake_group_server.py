"""Stand-in for `python -m karanta_ocr_amd.cli` in the launcher's CPU test: same command line shape (`serve MODEL
--port P ...`), same start-up sequence (serving group from the environment -> rank 0 "reads the checkpoint" -> weight
arena broadcast -> HTTP server with /health), a fake engine instead of the GPU one.  Not a test module."""
import os
import signal
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class Arena:
    nbytes = 1 << 16

    def __init__(self):
        self.arena = None

    def allocate(self):
        import torch
        self.arena = torch.zeros(self.nbytes, dtype=torch.uint8)


def main(argv):
    import numpy as np
    import torch

    from karanta_ocr_amd import serving as S
    from karanta_ocr_amd.config import CONFIGS
    from karanta_ocr_amd.dp import load_or_receive_weights, serving_group_env

    assert argv[0] == "serve"
    model, port = argv[1], int(argv[argv.index("--port") + 1])
    rank, world = serving_group_env()
    if os.environ.get("STUB_FAIL_RANK") == str(rank):
        print(f"rank {rank}: failing on purpose", flush=True)
        return 3
    out_dir = os.environ["KARANTA_TEST_OUT"]
    owner = Arena()

    def load():
        owner.allocate()
        owner.arena.copy_(torch.from_numpy(np.random.default_rng(11).integers(0, 256, owner.nbytes, dtype=np.uint8)))

    info = load_or_receive_weights(owner, rank, world, load, log=lambda m: print(m, flush=True), timeout_s=60)
    with open(os.path.join(out_dir, f"server_{port}.txt"), "w") as f:
        f.write(f"{rank} {world} {os.environ.get('HIP_VISIBLE_DEVICES')} {int(owner.arena.to(torch.int64).sum())} {model} "
                f"{' '.join(argv[2:])}\n")

    cfg = CONFIGS["tiny"]

    class Engine:
        B = 2

        def generate(self, pages, max_new_tokens, **kw):
            from types import SimpleNamespace
            toks = [np.asarray(list(b"OK") + [cfg.eos_token_ids[0]], np.int64) for _ in pages]
            return SimpleNamespace(tokens=toks, finish_reasons=["stop"] * len(pages), prompt_tokens=[len(p.input_ids) for p in pages])

    Engine.cfg = cfg
    srv = S.LocalServer(Engine(), S.ChatFrontend(cfg, S.ByteTokenizer(cfg)), log=lambda *a: print(*a, flush=True))
    httpd = S.serve_http(srv, port=port, host="127.0.0.1")
    stop = threading.Event()
    signal.signal(signal.SIGTERM, lambda *_: stop.set())
    signal.signal(signal.SIGINT, lambda *_: stop.set())
    stop.wait()
    httpd.shutdown()
    srv.close()
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
