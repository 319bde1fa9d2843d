"""OpenAI-chat-completions surface of the MI355X engine — the boundary the reference talks to.

The reference never calls a model in-process: ``process_page`` POSTs ``build_page_query``'s dict to
``{server}/chat/completions`` (/root/reference/karanta/pipeline.py:115-171, :317-344) and the bulk
workers go through ``VLLMClient.generate`` (/root/reference/bulk_processing/workers/vllm_client.py:
155-227).  This module turns exactly that request dict into engine work and the engine's tokens
back into exactly the response JSON those callers parse:

    choices[0].message.content, choices[0].finish_reason in {"stop","length"},
    usage.{prompt_tokens, completion_tokens, total_tokens}, model

Status codes follow what ``process_page`` distinguishes (pipeline.py:321-332): 200, 400 (bad
request: caller skips the attempt), 500 (internal error: caller retries).

``temperature`` is honoured (0 or absent: greedy — the reference's ``build_page_query`` default, pipeline.py:170;
``process_page`` sends 0.1 on its first attempt, :281,:301): Gumbel-max sampling from softmax(logits / T)
(kr_gumbel_argmax), reproducible through the OpenAI ``seed`` field, a random seed per request otherwise.
``top_p`` / ``top_k`` are not applied.  ``guided_regex`` / ``response_format`` / ``logprobs``: guided.py, engine.

Prompt text: the checkpoint's own ``chat_template`` when the model directory ships one (``chat_template.json`` /
``chat_template.jinja`` / ``tokenizer_config.json`` — what vLLM applies, pipeline.py:707-734), rendered with jinja2; the
hand-coded Qwen2-VL template otherwise.
"""
from __future__ import annotations

import json
import queue
import threading
import time
import uuid
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import image_processing as IP
from .config import ModelConfig

DEFAULT_SYSTEM = "You are a helpful assistant."


def load_chat_template(model_dir: str) -> Optional[str]:
    """The checkpoint's chat template, as `transformers` / vLLM look it up: ``chat_template.jinja``, ``chat_template.json``
    (``{"chat_template": ...}``: the processor's file on Qwen2-VL checkpoints), then ``tokenizer_config.json``'s
    ``chat_template`` (a string, or a list of ``{"name", "template"}`` whose "default" entry is taken).  None when the
    directory has none: the caller falls back to the hand-coded Qwen2-VL template."""
    import os
    try:
        with open(os.path.join(model_dir, "chat_template.jinja"), encoding="utf-8") as f:
            return f.read()
    except OSError:
        pass
    for name in ("chat_template.json", "tokenizer_config.json"):
        try:
            with open(os.path.join(model_dir, name), encoding="utf-8") as f:
                t = json.load(f).get("chat_template")
        except (OSError, ValueError, AttributeError):
            continue
        if isinstance(t, list):
            named = {e.get("name"): e.get("template") for e in t if isinstance(e, dict)}
            t = named.get("default") or next(iter(named.values()), None)
        if isinstance(t, str) and t.strip():
            return t
    return None


# ----------------------------------------------------------------------------- tokenizers
class ByteTokenizer:
    """Tokenizer of last resort (tests, random-init models): UTF-8 bytes as ids 0..255.
    Special tokens keep the model's own ids.  Real checkpoints use :class:`HFTokenizer`."""

    def __init__(self, cfg: ModelConfig, im_start: Optional[int] = None, im_end: Optional[int] = None):
        self.cfg = cfg
        self.im_start = im_start if im_start is not None else min(cfg.text.vocab_size - 1, 151644)
        self.im_end = im_end if im_end is not None else cfg.eos_token_ids[0]
        self.newline = 10

    def encode(self, text: str) -> List[int]:
        return list(text.encode("utf-8"))

    def decode(self, ids: Sequence[int]) -> str:
        special = set(self.cfg.eos_token_ids) | {self.im_start, self.cfg.pad_token_id}
        return bytes(int(i) for i in ids if 0 <= int(i) < 256 and int(i) not in special).decode("utf-8", "replace")

    def token_bytes(self) -> List[bytes]:
        """Byte string of every token id (guided decoding matches patterns against these); b"" = never allowed."""
        special = set(self.cfg.eos_token_ids) | {self.im_start, self.cfg.pad_token_id}
        V = self.cfg.text.vocab_size
        return [bytes([i]) if i < 256 and i not in special else b"" for i in range(V)]


class HFTokenizer:
    """``tokenizer.json`` of a real Qwen2-VL checkpoint through the `tokenizers` library."""

    def __init__(self, path: str, cfg: ModelConfig):
        from tokenizers import Tokenizer

        self.tk = Tokenizer.from_file(path)
        self.cfg = cfg
        self.im_start = self.tk.token_to_id("<|im_start|>")
        self.im_end = self.tk.token_to_id("<|im_end|>")
        self.newline = self.tk.encode("\n", add_special_tokens=False).ids[0]

    def encode(self, text: str) -> List[int]:
        return self.tk.encode(text, add_special_tokens=False).ids

    def decode(self, ids: Sequence[int]) -> str:
        return self.tk.decode([int(i) for i in ids], skip_special_tokens=True)

    def token_bytes(self) -> List[bytes]:
        from .guided import vocab_bytes_from_hf
        return vocab_bytes_from_hf(self.tk, self.cfg.text.vocab_size)


# ----------------------------------------------------------------------------- request parsing
class BadRequest(ValueError):
    """Maps to HTTP 400 (the reference skips the attempt, pipeline.py:321-324)."""


@dataclass
class ParsedRequest:
    input_ids: np.ndarray
    pixel_values: Optional[np.ndarray]
    grids: List[Tuple[int, int, int]]
    max_tokens: int
    model: str
    temperature: float = 0.0
    seed: Optional[int] = None      # None: the server draws one per request
    images: Optional[List[np.ndarray]] = None   # device_images front end: decoded HWC uint8 pages instead of pixel_values
    guide: Any = None               # guided.Guide compiled from guided_regex / response_format (None: unconstrained)
    logprobs: Optional[int] = None  # None: not asked; k: log-prob of every token + top_logprobs = k alternatives


class ChatFrontend:
    """messages -> Qwen2-VL chat-template token ids + image patches.

    Template (Qwen2-VL): ``<|im_start|>system\\n{sys}<|im_end|>\\n<|im_start|>user\\n{content}<|im_end|>\\n
    <|im_start|>assistant\\n``; an ``image_url`` part becomes ``<|vision_start|><|image_pad|>*T<|vision_end|>``
    in place, parts in message order (the reference puts the text first, the image second:
    /root/reference/karanta/data/utils.py:283-297)."""

    def __init__(self, cfg: ModelConfig, tokenizer, min_pixels: int = IP.MIN_PIXELS,
                 max_pixels: int = IP.MAX_PIXELS_CLASS_DEFAULT, max_model_len: int = 16384, device_images: bool = False,
                 chat_template: Optional[str] = None, upload_device: Optional[str] = None):
        self.cfg, self.tok = cfg, tokenizer
        # device_images + upload_device: the decoded page goes to HBM HERE, in the request's own thread and on a stream of its
        # own, instead of at admission on the scheduler thread (a blocking 3 MB copy per page with the GPU waiting behind it)
        self._upload_device, self._upload_stream = None, None
        if upload_device is not None and device_images:
            import torch
            if torch.cuda.is_available():
                self._upload_device = torch.device(upload_device)
                self._upload_stream = torch.cuda.Stream(device=self._upload_device)
        # the checkpoint's own jinja template (load_chat_template); None: the hand-coded Qwen2-VL turns below
        self._template = None
        if chat_template:
            import jinja2
            from jinja2.sandbox import ImmutableSandboxedEnvironment

            def raise_exception(msg):
                raise jinja2.exceptions.TemplateError(msg)

            def tojson(x, ensure_ascii=False, indent=None, separators=None, sort_keys=False):
                import json as _json
                return _json.dumps(x, ensure_ascii=ensure_ascii, indent=indent, separators=separators, sort_keys=sort_keys)

            def strftime_now(fmt):
                import datetime
                return datetime.datetime.now().strftime(fmt)

            # the environment transformers renders chat templates in (utils/chat_template_utils.py): sandboxed, trim / lstrip blocks,
            # loop controls ({% break %} / {% continue %}), tojson without HTML escaping, raise_exception, strftime_now
            env = ImmutableSandboxedEnvironment(trim_blocks=True, lstrip_blocks=True, extensions=["jinja2.ext.loopcontrols"])
            env.filters["tojson"] = tojson
            env.globals["raise_exception"] = raise_exception
            env.globals["strftime_now"] = strftime_now
            try:
                self._template = env.from_string(chat_template)
            except jinja2.exceptions.TemplateSyntaxError as e:
                # a template this environment cannot compile must not keep the server from starting: say so and serve with
                # the hand-coded Qwen2-VL turns
                import sys
                print(f"[karanta] chat template of the checkpoint does not compile ({e}); falling back to the hand-coded "
                      f"Qwen2-VL template", file=sys.stderr, flush=True)
                self._template = None
        self.min_pixels, self.max_pixels = min_pixels, max_pixels
        # True: only decode the image here; the engine resizes / normalises / patchifies it on the GPU
        # (Engine.patches_from_images, bit-identical to the host path) and 3 bytes per pixel cross PCIe
        self.device_images = bool(device_images)
        self.max_model_len = max_model_len  # reference --max_model_len default (pipeline.py:1225-1230)
        self._guide_cache: Dict[str, Any] = {}   # regex -> guided.Guide (the pipeline sends one pattern for every page)

    def _upload(self, u8: np.ndarray):
        """HWC uint8 page -> a tensor resident in HBM (Engine.patches_from_images takes it as it is).  The copy blocks this thread
        and the upload stream only; the request holds the tensor until it has been answered."""
        import torch
        with torch.cuda.stream(self._upload_stream):
            u8 = np.ascontiguousarray(u8)
            if not u8.flags.writeable:        # PIL-backed arrays are read-only views
                u8 = u8.copy()
            return torch.from_numpy(u8).to(self._upload_device)

    def _guide_for(self, req: Dict[str, Any]):
        """guided_regex (pipeline.py:304-307) / response_format (vllm_client.py:196) -> byte DFA, or None."""
        from .guided import GuideError, compile_regex, regex_for_request
        try:
            rx = regex_for_request(req.get("guided_regex"), req.get("response_format"))
            if rx is None:
                return None
            g = self._guide_cache.get(rx)
            if g is None:
                g = compile_regex(rx)
                if len(self._guide_cache) >= 64:
                    self._guide_cache.pop(next(iter(self._guide_cache)))
                self._guide_cache[rx] = g
            return g
        except GuideError as e:
            raise BadRequest(f"guided decoding: {e}") from e

    def _turn(self, role: str, body: List[int]) -> List[int]:
        t = self.tok
        return [t.im_start] + t.encode(role) + [t.newline] + body + [t.im_end, t.newline]

    def _special_ids(self) -> Dict[str, int]:
        return {"<|im_start|>": self.tok.im_start, "<|im_end|>": self.tok.im_end,
                "<|vision_start|>": self.cfg.vision_start_token_id, "<|vision_end|>": self.cfg.vision_end_token_id,
                "<|image_pad|>": self.cfg.image_token_id}

    def _ids_from_template(self, messages: List[dict], n_tok: List[int]) -> List[int]:
        """Render the checkpoint's template (`add_generation_prompt=True`, as vLLM does for a chat completion) and
        tokenise it: the special-token strings map to their ids here (so the byte tokenizer of the tests and the HF
        tokenizer agree), the text between them goes through the tokenizer, and the k-th ``<|image_pad|>`` becomes the
        k-th image's T placeholders — what the HF processor does to the rendered text
        (/root/reference/karanta/training/test_trained_model.py:77-87)."""
        import re

        import jinja2
        msgs = []
        for m in messages:
            c = m.get("content")
            if isinstance(c, list):   # OpenAI parts -> the shape HF templates test for ('image' in content / type == 'image')
                c = [({"type": "image", "image": ""} if p.get("type") in ("image_url", "image") else p) for p in c]
            msgs.append({**m, "content": c})
        try:
            text = self._template.render(messages=msgs, add_generation_prompt=True, add_vision_id=False, bos_token="",
                                         eos_token="<|im_end|>", pad_token="<|endoftext|>", unk_token="")
        except jinja2.exceptions.TemplateError as e:
            raise BadRequest(f"chat template: {e}") from e
        except Exception as e:  # noqa: BLE001  — a template bug on this request's shape is the client's 400, not a 500
            raise BadRequest(f"chat template: {type(e).__name__}: {e}") from e
        sp = self._special_ids()
        ids: List[int] = []
        k = 0
        for piece in re.split("(" + "|".join(re.escape(x) for x in sp) + ")", text):
            if not piece:
                continue
            if piece == "<|image_pad|>":
                if k >= len(n_tok):
                    raise BadRequest("chat template: more image placeholders than images")
                ids += [sp[piece]] * n_tok[k]
                k += 1
            elif piece in sp:
                ids.append(sp[piece])
            else:
                ids += self.tok.encode(piece)
        if k != len(n_tok):
            raise BadRequest(f"chat template rendered {k} image placeholders for {len(n_tok)} images")
        return ids

    def parse(self, req: Dict[str, Any]) -> ParsedRequest:
        if not isinstance(req, dict) or not isinstance(req.get("messages"), list) or not req["messages"]:
            raise BadRequest("`messages` must be a non-empty list")
        mt = req.get("max_tokens", req.get("max_completion_tokens"))
        max_tokens = 100 if mt is None else int(mt)
        if max_tokens < 1:
            raise BadRequest("max_tokens must be >= 1")
        ids: List[int] = []
        pvs, grids, images, n_toks = [], [], [], []
        hand = self._template is None        # the hand-coded Qwen2-VL turns; with the checkpoint's own template only the images are
        #                                      collected here and the template decides roles, order and text (no double tokenisation)
        if hand and req["messages"][0].get("role") != "system":
            ids += self._turn("system", self.tok.encode(DEFAULT_SYSTEM))
        for msg in req["messages"]:
            role, content = msg.get("role"), msg.get("content")
            if hand and role not in ("system", "user", "assistant"):
                raise BadRequest(f"unsupported role {role!r}")
            body: List[int] = []
            parts = [{"type": "text", "text": content}] if isinstance(content, str) else content
            if not isinstance(parts, list):
                raise BadRequest("message content must be a string or a list of parts")
            for part in parts:
                kind = part.get("type")
                if kind == "text":
                    if hand:
                        body += self.tok.encode(part.get("text", ""))
                elif kind in ("image_url", "image"):
                    url = part["image_url"]["url"] if kind == "image_url" else part["image"]
                    try:
                        img = IP.decode_data_url(url)
                    except Exception as e:  # undecodable image -> 400, like vLLM
                        raise BadRequest(f"cannot decode image: {e}") from e
                    if self.device_images:
                        u8 = np.asarray(img.convert("RGB"), dtype=np.uint8)
                        unit = self.cfg.vision.patch_size * self.cfg.vision.spatial_merge_size
                        try:
                            rh, rw = IP.smart_resize(u8.shape[0], u8.shape[1], unit, self.min_pixels, self.max_pixels)
                        except ValueError as e:
                            raise BadRequest(str(e)) from e
                        grid = (1, rh // self.cfg.vision.patch_size, rw // self.cfg.vision.patch_size)
                        images.append(self._upload(u8) if self._upload_device is not None else u8)
                    else:
                        pv, grid = IP.image_to_patches(img, self.min_pixels, self.max_pixels)
                        pvs.append(pv)
                    grids.append(grid)
                    n_tok = grid[0] * grid[1] * grid[2] // (self.cfg.vision.spatial_merge_size ** 2)
                    n_toks.append(n_tok)
                    body += [self.cfg.vision_start_token_id] + [self.cfg.image_token_id] * n_tok + [self.cfg.vision_end_token_id]
                else:
                    raise BadRequest(f"unsupported content part {kind!r}")
            if hand:
                ids += self._turn(role, body)
        if hand:
            ids += [self.tok.im_start] + self.tok.encode("assistant") + [self.tok.newline]
        else:
            ids = self._ids_from_template(req["messages"], n_toks)
        if len(ids) + max_tokens > self.max_model_len:
            raise BadRequest(f"prompt ({len(ids)} tokens) + max_tokens ({max_tokens}) exceeds max_model_len {self.max_model_len}")
        try:
            temperature = float(req.get("temperature") or 0.0)
            seed = None if req.get("seed") is None else int(req["seed"]) & 0xFFFFFFFF
        except (TypeError, ValueError) as e:
            raise BadRequest(f"temperature / seed: {e}") from e
        if not (0.0 <= temperature <= 100.0):   # NaN fails both comparisons
            raise BadRequest("temperature must be in [0, 100]")
        logprobs = None
        if req.get("logprobs"):
            try:
                logprobs = int(req.get("top_logprobs") or 0)
            except (TypeError, ValueError) as e:
                raise BadRequest(f"top_logprobs: {e}") from e
            if not 0 <= logprobs <= 20:
                raise BadRequest("top_logprobs must be in 0..20")
        return ParsedRequest(np.asarray(ids, np.int64), np.concatenate(pvs, 0) if pvs else None, grids, max_tokens,
                             str(req.get("model", "karantaocr")), temperature, seed, images or None,
                             self._guide_for(req), logprobs)


# ----------------------------------------------------------------------------- in-process server
class LocalServer:
    """Scheduler in front of one engine (one GPU).  Thread-safe: callers block in :meth:`chat_completions`
    while a worker thread runs the engine.

    ``continuous=False``: static batching — whatever is waiting (up to the engine's max batch) goes into one
    ``generate`` call and leaves together.  ``continuous=True``: the slot scheduler (scheduler.SlotScheduler) —
    a request leaves as soon as it hits EOS or its ``max_tokens`` and the next waiting request is prefilled into
    its slot while the others keep decoding; ``max_tokens_cap`` bounds any request's ``max_tokens`` and
    ``chunk`` is the number of decode steps between two looks at the device's finished flags (with launch_ahead the next chunk is
    queued before the look: scheduler.SlotScheduler)."""

    def __init__(self, engine, frontend: ChatFrontend, served_model_name: str = "karantaocr",
                 batch_wait_s: float = 0.005, log=print, continuous: bool = False, max_tokens_cap: int = 4096,
                 chunk: int = 2, honor_temperature: bool = True, max_logprobs: Optional[int] = None, admit_min: int = 1,
                 admit_max_wait: int = 16, overlap_admissions: bool = False, launch_ahead: bool = True):
        self.engine, self.frontend, self.name = engine, frontend, served_model_name
        self.honor_temperature = bool(honor_temperature)   # False: every request is served greedy
        # guided decoding needs the tokenizer's byte strings on the device; engines without set_vocab (test fakes)
        # answer guided requests with 400
        self.guided = hasattr(engine, "set_vocab") and hasattr(frontend.tok, "token_bytes")
        if self.guided:
            engine.set_vocab(frontend.tok.token_bytes())
        # continuous mode: the decode graph records log-probabilities only when this is set (top_logprobs bound,
        # vLLM's --max-logprobs); static mode decides per batch
        self.max_logprobs = None if max_logprobs is None else int(max_logprobs)
        self.batch_wait_s, self.log = batch_wait_s, log
        self.continuous, self.max_tokens_cap, self.chunk = bool(continuous), int(max_tokens_cap), int(chunk)
        self.admit_min, self.admit_max_wait = int(admit_min), int(admit_max_wait)   # SlotScheduler's admission batching
        # continuous mode: ViT + prefill of an admission on a second (CU-masked: Engine(admission_cus=...)) stream while the
        # other slots keep decoding (SlotScheduler(overlap=True))
        self.overlap_admissions = bool(overlap_admissions)
        self.launch_ahead = bool(launch_ahead)
        self._q: "queue.Queue" = queue.Queue()
        self._running = 0
        self._stop = False
        self.pages_done = 0
        self._t_start = time.time()
        self.latencies: List[float] = []
        self._thread = threading.Thread(target=self._loop_continuous if self.continuous else self._loop, daemon=True)
        self._thread.start()
        # the reference watches the server's output for one of these lines (pipeline.py:790-800)
        self.log("Starting vLLM API server (karanta MI355X engine)")

    # -- public surface -------------------------------------------------------
    def health(self) -> Tuple[int, dict]:
        return 200, {"status": "ok", "engine": "karanta-mi355x", "running": self._running, "waiting": self._q.qsize()}

    def metrics(self) -> Tuple[int, dict]:
        """Pages served and request latency percentiles (the reference computes neither)."""
        lat = sorted(self.latencies[-10000:])
        pct = lambda q: float(lat[min(len(lat) - 1, int(q * len(lat)))]) if lat else None
        up = max(1e-9, time.time() - self._t_start)
        return 200, {"pages_done": self.pages_done, "uptime_s": round(up, 3), "pages_per_s": round(self.pages_done / up, 4),
                     "running": self._running, "waiting": self._q.qsize(),
                     "latency_s": {"p50": pct(0.50), "p95": pct(0.95), "p99": pct(0.99), "n": len(lat)},
                     "scheduler": self.scheduler_stats()}

    def scheduler_stats(self) -> Optional[dict]:
        """Continuous mode: where the scheduler thread's wall time went and how full the decode slots were."""
        sch = getattr(self, "_sch", None)
        if sch is None:
            return None
        return {"decode_steps": sch.steps, "slot_occupancy": round(sch.slot_steps_busy / max(1, sch.steps * sch.n_slots), 4),
                "admissions": sch.admissions, "pages_admitted": sch.pages_admitted,
                "host_phase_s": {k: round(v, 3) for k, v in sch.phase_s.items()}}

    def models(self) -> Tuple[int, dict]:
        return 200, {"object": "list", "data": [{"id": self.name, "object": "model", "owned_by": "karanta"}]}

    def chat_completions(self, req: Dict[str, Any]) -> Tuple[int, dict]:
        t0 = time.time()
        try:
            parsed = self.frontend.parse(req)
        except BadRequest as e:
            return 400, {"error": {"message": str(e), "type": "BadRequestError", "code": 400}}
        except Exception as e:  # malformed beyond recognition
            return 400, {"error": {"message": f"malformed request: {e}", "type": "BadRequestError", "code": 400}}
        if parsed.guide is not None and not self.guided:
            return 400, {"error": {"message": "guided decoding is not available on this server", "type": "BadRequestError", "code": 400}}
        if parsed.logprobs is not None and self.continuous and (self.max_logprobs is None or parsed.logprobs > self.max_logprobs):
            return 400, {"error": {"message": f"logprobs: this server records at most {self.max_logprobs} top_logprobs "
                                              "(start it with --max-logprobs)", "type": "BadRequestError", "code": 400}}
        room = self.engine.seq_room() if hasattr(self.engine, "seq_room") else None
        over = self.chunk * (2 if self.launch_ahead and not self.overlap_admissions else 1)   # steps past its limit (scheduler.over)
        need = len(parsed.input_ids) + min(int(parsed.max_tokens), self.max_tokens_cap) + (over if self.continuous else 0)
        if room is not None and need > room:
            # the request's OWN prompt + max_tokens against one sequence's cache rows: a client error for this request
            # only (the reference skips the attempt on 400, pipeline.py:321-332) — never an engine exception that
            # would fail the requests admitted with it
            return 400, {"error": {"message": f"prompt ({len(parsed.input_ids)} tokens) + max_tokens ({parsed.max_tokens}) "
                                              f"exceeds this server's sequence capacity ({room})",
                                   "type": "BadRequestError", "code": 400}}
        slot: Dict[str, Any] = {"req": parsed, "done": threading.Event()}
        self._q.put(slot)
        slot["done"].wait()
        if "error" in slot:
            code = int(slot.get("status", 500))
            kind = "BadRequestError" if code == 400 else "InternalServerError"
            return code, {"error": {"message": slot["error"], "type": kind, "code": code}}
        toks, reason = slot["tokens"], slot["reason"]
        text = self.frontend.tok.decode(toks)
        self.latencies.append(time.time() - t0)
        if len(self.latencies) > 20000:          # bounded: /metrics reads the last 10000
            del self.latencies[:10000]
        n_in, n_out = int(len(parsed.input_ids)), int(len(toks))
        choice: Dict[str, Any] = {"index": 0, "message": {"role": "assistant", "content": text}, "finish_reason": reason}
        if parsed.logprobs is not None:
            choice["logprobs"] = self._logprobs_json(toks, slot.get("logprobs"), parsed.logprobs)
        return 200, {
            "id": "chatcmpl-" + uuid.uuid4().hex, "object": "chat.completion", "created": int(time.time()),
            "model": req.get("model", self.name),
            "choices": [choice],
            "usage": {"prompt_tokens": n_in, "completion_tokens": n_out, "total_tokens": n_in + n_out},
        }

    def _logprobs_json(self, toks, lp, k: int) -> Optional[dict]:
        """OpenAI chat `logprobs` object: per generated token its text, log-prob, bytes and the top-k alternatives."""
        if lp is None:
            return None
        tok = self.frontend.tok
        piece = lambda i: tok.decode([int(i)])
        item = lambda i, v: {"token": piece(i), "logprob": float(v), "bytes": list(piece(i).encode("utf-8"))}
        n = min(len(toks), len(lp["token"]))
        return {"content": [dict(item(toks[i], lp["token"][i]),
                                 top_logprobs=[item(lp["top_ids"][i][j], lp["top"][i][j]) for j in range(min(k, lp["top"].shape[1]))])
                            for i in range(n)]}

    def close(self):
        self._stop = True
        self._q.put(None)
        self._thread.join(timeout=5)

    # -- scheduler --------------------------------------------------------------
    def _loop(self):
        from .engine import PageRequest

        while not self._stop:
            first = self._q.get()
            if first is None:
                break
            batch = [first]
            deadline = time.time() + self.batch_wait_s
            while len(batch) < self.engine.B:
                try:
                    nxt = self._q.get(timeout=max(0.0, deadline - time.time()))
                except queue.Empty:
                    break
                if nxt is None:
                    self._stop = True
                    break
                batch.append(nxt)
            self._running = len(batch)
            # same shape as vLLM's periodic stats line that the reference scrapes (pipeline.py:782-800)
            self.log(f"Running: {self._running} reqs, Waiting: {self._q.qsize()} reqs")
            for group in self._static_groups(batch):
                try:
                    pages = [self._page(s["req"]) for s in group]
                    res = self.engine.generate(pages, max(s["req"].max_tokens for s in group))
                    for i, (s, toks, reason) in enumerate(zip(group, res.tokens, res.finish_reasons)):
                        if getattr(res, "logprobs", None) is not None:
                            s["logprobs"] = res.logprobs[i]
                        self._finish(s, toks, reason)
                except Exception as e:  # engine failure -> 500 for every request of the group
                    for s in group:
                        s["error"] = f"{type(e).__name__}: {e}"
            self.pages_done += len(batch)
            self._running = 0
            for s in batch:
                s["done"].set()


    def _static_groups(self, batch):
        """A static batch decodes every sequence for the LONGEST max_tokens of the batch, so a long prompt with a small
        limit can overflow its cache rows next to a short prompt with a large one although both fit alone (each was
        checked at the door).  Split such a batch into groups that fit together; normally one group."""
        room = self.engine.seq_room() if hasattr(self.engine, "seq_room") else None
        if room is None:
            return [batch]
        groups: List[list] = []
        for s in sorted(batch, key=lambda s: -int(s["req"].max_tokens)):
            for g in groups:
                mt = max(int(x["req"].max_tokens) for x in g + [s])
                if all(len(x["req"].input_ids) + mt <= room for x in g + [s]):
                    g.append(s)
                    break
            else:
                groups.append([s])
        return groups

    def _page(self, r: ParsedRequest):
        from .engine import PageRequest
        import random
        page = PageRequest(r.input_ids, r.pixel_values, r.grids, images=getattr(r, "images", None))
        if getattr(r, "guide", None) is not None:
            page.guide = r.guide
        if getattr(r, "logprobs", None) is not None:
            page.logprobs = r.logprobs
        if self.honor_temperature and r.temperature > 0:
            page.temperature = r.temperature
            page.seed = r.seed if r.seed is not None else random.getrandbits(32)
        return page

    def _finish(self, s: Dict[str, Any], toks, reason: str):
        mt = s["req"].max_tokens
        if len(toks) > mt:
            toks, reason = toks[:mt], "length"
        eos = set(int(e) for e in self.engine.cfg.eos_token_ids)
        if reason == "stop" and len(toks) and int(toks[-1]) in eos:
            toks = toks[:-1]  # the EOS token is not part of the message content
        s["tokens"], s["reason"] = toks, reason

    def _loop_continuous(self):
        from .engine import PageRequest
        from .scheduler import SlotRequest, SlotScheduler

        try:
            sch = SlotScheduler(self.engine, self.max_tokens_cap, self.chunk, sampling=self.honor_temperature,
                                guided=self.guided, logprobs=self.max_logprobs, admit_min=self.admit_min,
                                admit_max_wait=self.admit_max_wait, overlap=self.overlap_admissions,
                                launch_ahead=self.launch_ahead)
        except Exception as e:  # cannot enter slot mode: every request gets a 500
            sch, boot_error = None, f"{type(e).__name__}: {e}"
        self._sch = sch
        last = (-1, -1)
        while not self._stop:
            # block only when there is nothing to decode; otherwise take what has arrived and keep stepping
            try:
                first = self._q.get(block=sch is None or sch.idle, timeout=None)
            except queue.Empty:
                first = False
            got = [] if first is False else [first]
            while True:
                try:
                    got.append(self._q.get_nowait())
                except queue.Empty:
                    break
            for s in got:
                if s is None:
                    self._stop = True
                elif sch is None:
                    s["error"] = boot_error
                    s["done"].set()
                else:
                    r = s["req"]
                    sch.submit(SlotRequest(self._page(r), max(1, int(r.max_tokens)), tag=s))
            if self._stop or sch is None:
                continue
            try:
                results = sch.step()
            except Exception as e:  # engine failure mid-flight: everything in a slot or waiting fails, slot mode restarts
                msg = f"{type(e).__name__}: {e}"
                for r in list(sch.active.values()) + list(sch.waiting):
                    r.tag["error"] = msg
                    r.tag["done"].set()
                self.pages_done += len(sch.active) + len(sch.waiting)
                try:
                    sch = SlotScheduler(self.engine, self.max_tokens_cap, self.chunk, sampling=self.honor_temperature,
                                        guided=self.guided, logprobs=self.max_logprobs, admit_min=self.admit_min,
                                        admit_max_wait=self.admit_max_wait, overlap=self.overlap_admissions,
                                launch_ahead=self.launch_ahead)
                except Exception as e2:
                    sch, boot_error = None, f"{type(e2).__name__}: {e2}"
                self._sch = sch
                continue
            self._running = sch.running
            now = (sch.running, len(sch.waiting) + self._q.qsize())
            if now != last:  # vLLM's stats line, printed when it changes (pipeline.py:782-800 scrapes it)
                self.log(f"Running: {now[0]} reqs, Waiting: {now[1]} reqs")
                last = now
            for res in results:
                s = res.tag
                if res.error:
                    s["error"] = res.error
                    s["status"] = getattr(res, "status", 500)
                else:
                    if res.logprobs is not None:
                        s["logprobs"] = res.logprobs
                    self._finish(s, res.tokens, res.finish_reason)
                self.pages_done += 1
                s["done"].set()
        # shutting down: nobody is left waiting forever
        if sch is not None:
            for r in list(sch.active.values()) + list(sch.waiting):
                r.tag["error"] = "server shutting down"
                r.tag["done"].set()


# registry used by the in-process VLLMClient (clients.py): port -> LocalServer
_LOCAL_SERVERS: Dict[Tuple[str, int], LocalServer] = {}


def register_local_server(port: int, server: LocalServer, host: str = "localhost") -> None:
    _LOCAL_SERVERS[(host, int(port))] = server


def unregister_local_server(port: int, host: str = "localhost") -> None:
    _LOCAL_SERVERS.pop((host, int(port)), None)


def local_server(host: str, port: int) -> Optional[LocalServer]:
    return _LOCAL_SERVERS.get((host, int(port))) or (_LOCAL_SERVERS.get(("localhost", int(port))) if host in ("127.0.0.1", "localhost") else None)


# ----------------------------------------------------------------------------- HTTP shim
def serve_http(server: LocalServer, port: int, host: str = "127.0.0.1"):
    """OpenAI-compatible HTTP front (``POST /v1/chat/completions``, ``GET /v1/models``, ``GET /health``)
    so the unmodified reference (``karanta.pipeline --server http://host:port/v1``, the Celery workers'
    ``VLLMClient``) can point at this engine.  Returns the running ``ThreadingHTTPServer``."""
    from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer

    class Handler(BaseHTTPRequestHandler):
        protocol_version = "HTTP/1.1"

        def log_message(self, *a):  # --disable-log-requests
            pass

        def _send(self, status: int, body: dict):
            data = json.dumps(body).encode()
            self.send_response(status)
            self.send_header("Content-Type", "application/json")
            self.send_header("Content-Length", str(len(data)))
            self.send_header("Connection", "close")
            self.end_headers()
            self.wfile.write(data)
            self.close_connection = True

        def do_GET(self):
            path = self.path.split("?")[0].rstrip("/")
            if path == "/health":
                self._send(*server.health())
            elif path in ("/v1/models", "/models"):
                self._send(*server.models())
            elif path == "/metrics":
                self._send(*server.metrics())
            else:
                self._send(404, {"error": {"message": "not found", "code": 404}})

        def do_POST(self):
            path = self.path.split("?")[0].rstrip("/")
            if path not in ("/v1/chat/completions", "/chat/completions"):
                return self._send(404, {"error": {"message": "not found", "code": 404}})
            try:
                n = int(self.headers.get("Content-Length", "0"))
                req = json.loads(self.rfile.read(n))
            except Exception as e:
                return self._send(400, {"error": {"message": f"invalid JSON: {e}", "type": "BadRequestError", "code": 400}})
            self._send(*server.chat_completions(req))

    httpd = ThreadingHTTPServer((host, port), Handler)
    httpd.daemon_threads = True
    threading.Thread(target=httpd.serve_forever, daemon=True).start()
    server.log("The server is fired up and ready to roll!")
    return httpd
