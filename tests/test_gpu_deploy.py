"""The deployment path executed end to end on the GPU: `python -m karanta_ocr_amd.cli serve <model dir> --port N` with the REAL
server factory (cli.make_server: weights.load_checkpoint -> Engine -> serving.HFTokenizer -> the checkpoint's own chat template ->
LocalServer -> HTTP), on a synthetic hub-layout checkpoint directory (tools/synthetic_checkpoint.py), answering the request the
reference sends (/root/reference/karanta/pipeline.py:115-171 build_page_query, :317-319 POST; create_vision_message,
/root/reference/karanta/data/utils.py:283-297) — and the text it returns is the oracle's greedy ids detokenised by the same
tokenizer.  This is what `vllm serve` is to the reference (/root/reference/karanta/pipeline.py:707-742)."""
import json
import os
import socket
import threading
import urllib.request

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from karanta_ocr_amd import cli  # noqa: E402
from karanta_ocr_amd import image_processing as IP  # noqa: E402
from karanta_ocr_amd import serving as S  # noqa: E402
from karanta_ocr_amd.config import CONFIGS  # noqa: E402
from karanta_ocr_amd.tools import synthetic_checkpoint as SC  # noqa: E402
from oracle import qwen2vl_oracle as O  # noqa: E402
from tests.test_deploy_cpu import reference_request  # noqa: E402
from oracle.tolerances import LOGIT_TOL_REL, TOKEN_MARGIN_FACTOR  # noqa: E402


def _free_port():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _post(port, path, body=None, timeout=120):
    req = urllib.request.Request(f"http://127.0.0.1:{port}{path}", data=None if body is None else json.dumps(body).encode(),
                                 headers={"Content-Type": "application/json"})
    with urllib.request.urlopen(req, timeout=timeout) as r:
        return r.status, json.loads(r.read() or b"{}")


def _decisive_case(cfg, d, page, n_new, fp8, layout):
    """Write checkpoints for successive seeds until the oracle's greedy run on the reference request has a top-2 margin above
    twice the logit tolerance at every step (a random-init model has near-ties; an HTTP client cannot teacher-force past them)."""
    for seed in range(40):
        meaning = SC.write_checkpoint(d, cfg, seed, layout, fp8, max_pixels=200704)
        tok = S.HFTokenizer(os.path.join(d, "tokenizer.json"), cfg)
        front = S.ChatFrontend(cfg, tok, min_pixels=3136, max_pixels=200704, chat_template=S.load_chat_template(d))
        pr = front.parse(reference_request(page, max_tokens=n_new))
        o_tok, o_log = O.generate_greedy(cfg, meaning, pr.input_ids[None], pr.pixel_values, pr.grids, n_new, policy="bf16",
                                         return_logits=True)
        toks = [int(t) for t in o_tok[0]]
        n = next((i + 1 for i, t in enumerate(toks) if t in cfg.eos_token_ids), n_new)       # steps up to and including EOS
        tol = LOGIT_TOL_REL * float(np.abs(o_log[0, 0]).max())
        top2 = np.sort(o_log[0, :n], axis=-1)[:, -2:]
        if (top2[:, 1] - top2[:, 0] > TOKEN_MARGIN_FACTOR * tol).all():
            return seed, pr, toks[:n], tok
    pytest.fail("no seed with a decisive greedy run in 40 tries")


@pytest.mark.parametrize("name,layout,fp8,slots", [("tiny", "v4", False, 2), ("tiny-2.5", "v5", False, 2), ("tiny-w512", "v4", True, 20)])
def test_cli_serve_with_the_real_server_factory_answers_the_reference_request(tmp_path, name, layout, fp8, slots):
    cfg = CONFIGS[name]
    d = str(tmp_path / "karantaocr-ckpt")
    page = IP.synthetic_page(17, 140, 196)
    n_new = 10
    seed, pr, want_ids, tok = _decisive_case(cfg, d, page, n_new, fp8, layout)
    want_text = tok.decode([t for t in want_ids if t not in cfg.eos_token_ids])
    stopped = want_ids[-1] in cfg.eos_token_ids
    port = _free_port()
    ready, box, logs = threading.Event(), {}, []
    on_ready = lambda httpd, srv, stop: (box.update(stop=stop, srv=srv), ready.set())

    def run():
        try:
            # the pipeline's own command line (karanta/pipeline.py:707-734), this engine's sizes
            box["rc"] = cli.main(["serve", d, "--port", str(port), "--host", "127.0.0.1", "--disable-log-requests", "--uvicorn-log-level",
                                  "warning", "--served-model-name", "karantaocr", "--tensor-parallel-size", "1", "--data-parallel-size", "1",
                                  "--limit-mm-per-prompt", '{"video": 0}', "--max-model-len", "1024", "--max-num-seqs", str(slots),
                                  "--max-num-batched-tokens", "2048", "--max-tokens-cap", "64"], on_ready=on_ready)
        except BaseException as e:      # noqa: BLE001  (a failure in the server thread must fail the test, not hang it)
            box["error"] = e
            ready.set()
    th = threading.Thread(target=run, daemon=True)
    th.start()
    assert ready.wait(300), "server did not come up"
    assert "error" not in box, f"cli.main raised: {box.get('error')!r}"
    try:
        srv = box["srv"]
        eng = srv.engine
        # the factory sized the engine by the admission budget and read the checkpoint's preprocessor bounds
        assert eng.max_tokens == 2048 and eng.max_patches == 4 * 2048 + 64 * slots and eng.B == slots
        assert srv.frontend.max_pixels == 200704 and srv.frontend._template is not None
        assert isinstance(srv.frontend.tok, S.HFTokenizer) and eng.fp8 == fp8
        st, models = _post(port, "/v1/models")
        assert st == 200 and models["data"][0]["id"] == "karantaocr"
        assert _post(port, "/health")[0] == 200
        st, body = _post(port, "/v1/chat/completions", reference_request(page, max_tokens=n_new))
        assert st == 200
        ch = body["choices"][0]
        assert ch["message"]["content"] == want_text, f"seed {seed}: got {ch['message']['content']!r}, oracle {want_text!r}"
        assert ch["finish_reason"] == ("stop" if stopped else "length")
        assert body["usage"]["prompt_tokens"] == len(pr.input_ids) and body["usage"]["completion_tokens"] == len(want_ids)
        assert body["model"] == "karantaocr"
        # a second, concurrent pair of requests (one of them guided by the pipeline's pattern) comes back too
        out = [None, None]
        rx = r"[a-e]{2,6}"
        reqs = [reference_request(page, max_tokens=n_new), reference_request(IP.synthetic_page(18, 112, 112), max_tokens=8, guided_regex=rx)]
        ts = [threading.Thread(target=lambda i=i: out.__setitem__(i, _post(port, "/v1/chat/completions", reqs[i]))) for i in range(2)]
        [t.start() for t in ts]
        [t.join(120) for t in ts]
        assert out[0][0] == 200 and out[0][1]["choices"][0]["message"]["content"] == want_text
        import re
        assert out[1][0] == 200 and re.fullmatch(rx, out[1][1]["choices"][0]["message"]["content"])
    finally:
        box["stop"].set()
        th.join(60)
    assert box.get("rc") == 0


@pytest.mark.parametrize("name,fp8", [("tiny", False), ("tiny-w512", True)])
def test_checkpoint_parity_tool_on_a_synthetic_directory(tmp_path, name, fp8):
    """`python -m oracle.checkpoint_parity <model dir>`: what a maintainer with real weights runs — engine
    and full-depth oracle on the same checkpoint directory, per-step logit error, argmax equality and the two decoded strings
    (call sequence of /root/reference/karanta/training/test_trained_model.py:76-99)."""
    from PIL import Image
    from oracle import checkpoint_parity as CP
    d = str(tmp_path / "ckpt")
    SC.write_checkpoint(d, CONFIGS[name], 3, "v4", fp8, max_pixels=200704)
    img = str(tmp_path / "scan.png")
    Image.fromarray(IP.synthetic_page(5, 168, 224)).save(img)
    rep = CP.run(d, [img], 1, 8, "bf16", 100352, CP.REFERENCE_PROMPT, log=lambda *a: None)
    assert len(rep["pages"]) == 2 and rep["pass"], rep
    for p in rep["pages"]:
        assert p["max_rel_err"] < p["tol_rel"] and p["steps"] == 8 and isinstance(p["engine_text"], str)
        assert p["prompt_tokens"] > p["image_tokens"] > 0
    assert CP.main([d, "--page", img, "--steps", "4", "--policy", "fp32", "--max-pixels", "100352", "--json", str(tmp_path / "r.json")]) in (0, 1)
    assert json.load(open(tmp_path / "r.json"))["pages"][0]["page"] == img
