"""Client side of the drop-in boundary: the bulk workers' ``VLLMClient`` family with the MI355X engine behind it.

The reference's Celery workers reach their model server through ``bulk_processing/workers/vllm_client.py``
(an OpenAI-SDK wrapper).  A worker switches to this engine by importing the same names from here; what is kept is the
*surface* — class and method names, argument names and defaults, the result-dict keys, when ``VLLMClientError`` is
raised — and nothing of the implementation:

====================================  =======================================================================
reference (vllm_client.py)            kept here
====================================  =======================================================================
``VLLMClient(...)`` :28-74            same seven arguments and defaults; ``base_url`` / ``health_url`` attributes
``health_check(force)`` :76-110       True / False, a positive answer is remembered for 60 s
``get_server_info(force_refresh)``    ``{"models", "base_url", "port", "host", "last_updated", + /health JSON}``
``generate(...)`` :155-227            unhealthy server or no model -> ``VLLMClientError``; ``max_retries + 1`` tries,
                                      pause ``retry_delay * 2**k`` after try k, then ``VLLMClientError``
result dict :229-266                  ``text, finish_reason, model, usage{3}, metadata{generation_time, server_url,
                                      generation_params (without messages), timestamp}``
``batch_generate`` :268-296           one result per prompt, failures as ``{"error", "metadata": {batch_index, failed}}``
``VLLMClientManager`` :304-386        one cached client per port; ``worker_port_{port}_{i}@host`` -> port
``get_vllm_client_for_worker``        module-level manager
====================================  =======================================================================

Transport (new): a :class:`karanta_ocr_amd.serving.LocalServer` registered for the client's port is called in-process
(no HTTP, no SDK); otherwise the three endpoints are reached with the standard library's HTTP client, i.e. the shim of
``serving.serve_http`` / ``python -m karanta_ocr_amd.cli`` or any OpenAI-compatible server.
"""
from __future__ import annotations

import json
import logging
import re
import time
import urllib.error
import urllib.request
from typing import Any, Dict, List, Optional, Tuple

from .serving import local_server

logger = logging.getLogger(__name__)

_HEALTH_TTL_S = 60.0                                   # how long a positive health answer is trusted
_WORKER_PORT = re.compile(r"(?:^|_)port_(\d+)(?:_|@|$)")  # worker_port_8000_1@hostname -> 8000


class VLLMClientError(Exception):
    """Raised for every failure the caller cannot retry its way out of (same name as the reference's)."""


class _Endpoint:
    """GET / POST against one server: in-process when a LocalServer is registered for (host, port), HTTP otherwise.
    Always answers ``(status, json_body)``; only connection-level failures raise."""

    _ROUTES = {"/health": "health", "/v1/models": "models"}

    def __init__(self, host: str, port: int, api_key: str, timeout: float):
        self.host, self.port, self.api_key, self.timeout = host, port, api_key, timeout

    def request(self, path: str, payload: Optional[dict] = None, timeout: Optional[float] = None) -> Tuple[int, dict]:
        srv = local_server(self.host, self.port)
        if srv is not None:
            if payload is None:
                return getattr(srv, self._ROUTES[path])()
            return srv.chat_completions(payload)
        req = urllib.request.Request(f"http://{self.host}:{self.port}{path}",
                                     data=None if payload is None else json.dumps(payload).encode(),
                                     method="GET" if payload is None else "POST",
                                     headers={"Content-Type": "application/json",
                                              **({"Authorization": f"Bearer {self.api_key}"} if self.api_key else {})})
        try:
            with urllib.request.urlopen(req, timeout=self.timeout if timeout is None else timeout) as resp:
                return resp.status, self._json(resp.read())
        except urllib.error.HTTPError as e:      # 4xx / 5xx still carry the server's JSON error body
            return e.code, self._json(e.read())

    @staticmethod
    def _json(raw: bytes) -> dict:
        if not raw:
            return {}
        try:
            doc = json.loads(raw)
            return doc if isinstance(doc, dict) else {"data": doc}
        except ValueError:
            return {"error": {"message": raw.decode("utf-8", "replace")}}


class VLLMClient:
    def __init__(self, port: int, host: str = "localhost", api_key: str = "EMPTY", timeout: float = 300.0,
                 max_retries: int = 3, retry_delay: float = 1.0, health_check_timeout: float = 30.0):
        self.port, self.host, self.api_key = port, host, api_key
        self.timeout, self.health_check_timeout = timeout, health_check_timeout
        self.max_retries, self.retry_delay = max_retries, retry_delay
        root = f"http://{host}:{port}"
        self.base_url, self.health_url = root + "/v1", root + "/health"
        self._endpoint = _Endpoint(host, port, api_key, timeout)
        self._info: Optional[Dict[str, Any]] = None
        self._healthy_until = 0.0

    def __repr__(self) -> str:
        return f"VLLMClient(host={self.host}, port={self.port}, base_url={self.base_url})"

    # ------------------------------------------------------------------ probes
    def health_check(self, force: bool = False) -> bool:
        now = time.time()
        if not force and now < self._healthy_until:
            return True
        try:
            status, _ = self._endpoint.request("/health", timeout=self.health_check_timeout)
        except Exception as exc:                      # refused connection, time-out, DNS ...
            logger.error("health probe of %s failed: %s", self.health_url, exc)
            return False
        if status != 200:
            logger.warning("health probe of %s answered %s", self.health_url, status)
            return False
        self._healthy_until = now + _HEALTH_TTL_S
        return True

    def get_server_info(self, force_refresh: bool = False) -> Dict[str, Any]:
        if self._info is not None and not force_refresh:
            return self._info
        try:
            status, listing = self._endpoint.request("/v1/models")
            if status != 200:
                raise RuntimeError(f"/v1/models answered {status}")
            info: Dict[str, Any] = {"models": [entry["id"] for entry in listing.get("data", [])], "base_url": self.base_url,
                                    "port": self.port, "host": self.host, "last_updated": time.time()}
        except Exception as exc:
            raise VLLMClientError(f"Failed to get server info: {exc}") from exc
        try:                                          # whatever /health reports beyond its status, best effort
            status, extra = self._endpoint.request("/health", timeout=5)
            if status == 200:
                info.update(extra)
        except (OSError, ValueError, RuntimeError):
            pass
        self._info = info
        return info

    # ------------------------------------------------------------------ generation
    def generate(self, messages: List[Dict[str, str]], model: Optional[str] = None, max_tokens: int = 100,
                 temperature: float = 0.7, response_format: Optional[Dict[str, Any]] = None, **kwargs) -> Dict[str, Any]:
        if self.health_check() is False:
            raise VLLMClientError(f"{self.base_url}: the VLLM server is not healthy (GET {self.health_url} did not answer 200)")
        if not model:
            served = self.get_server_info().get("models")
            if not served:
                raise VLLMClientError(f"{self.base_url} serves no model")
            model = served[0]
        request = dict(kwargs, model=model, messages=messages, max_tokens=max_tokens, temperature=temperature,
                       response_format=response_format)
        t_first = time.time()
        tries = self.max_retries + 1
        failure: Optional[Exception] = None
        for k in range(tries):
            if k:
                time.sleep(self.retry_delay * 2 ** (k - 1))
            try:
                status, body = self._endpoint.request("/v1/chat/completions", request)
                if status == 200:
                    return self._process_response(body, t_first, request)
                failure = RuntimeError(f"Error code: {status} - {body.get('error', body)}")
            except VLLMClientError:
                raise
            except Exception as exc:
                failure = exc
            logger.warning("chat completion try %d of %d on %s failed: %s", k + 1, tries, self.base_url, failure)
        raise VLLMClientError(f"Generation failed after {tries} attempts. Last error: {failure}")

    def _process_response(self, response: dict, start_time: float, generation_params: Dict[str, Any]) -> Dict[str, Any]:
        choices = response.get("choices") or []
        if not choices:
            raise VLLMClientError(f"{self.base_url} answered without choices")
        first, usage, now = choices[0], response.get("usage") or {}, time.time()
        echoed = dict(generation_params)
        echoed.pop("messages", None)                  # the prompt (a page image) is not echoed back
        return {
            "text": (first.get("message") or {}).get("content"),
            "finish_reason": first.get("finish_reason"),
            "model": response.get("model"),
            "usage": {key: int(usage.get(key) or 0) for key in ("prompt_tokens", "completion_tokens", "total_tokens")},
            "metadata": {"generation_time": now - start_time, "server_url": self.base_url, "generation_params": echoed,
                         "timestamp": now},
        }

    def batch_generate(self, prompts: List[Any], **generation_kwargs) -> List[Dict[str, Any]]:
        """One :meth:`generate` per element, in order; a failure becomes that element's result.  Elements are message
        lists (what the reference passes through, vllm_client.py:286); a bare string is sent as one user message."""
        out: List[Dict[str, Any]] = []
        for index, item in enumerate(prompts):
            conversation = [{"role": "user", "content": item}] if isinstance(item, str) else item
            try:
                result = self.generate(conversation, **generation_kwargs)
            except Exception as exc:
                logger.error("batch item %d failed: %s", index, exc)
                out.append({"error": str(exc), "metadata": {"batch_index": index, "failed": True}})
                continue
            result["metadata"]["batch_index"] = index
            out.append(result)
        return out


class VLLMClientManager:
    """Clients by port, created on first use; ``server_config`` maps a port to its host (default ``localhost``)."""

    def __init__(self, server_config: Dict[int, str] = None):
        self.server_config: Dict[int, str] = dict(server_config or {})
        self.clients = {}   # port -> VLLMClient

    def get_client(self, port: int, **client_kwargs) -> VLLMClient:
        client = self.clients.get(port)
        if client is None:
            client = self.clients[port] = VLLMClient(port=port, host=self.server_config.get(port, "localhost"), **client_kwargs)
        return client

    def health_check_all(self) -> Dict[int, bool]:
        report: Dict[int, bool] = {}
        for port in sorted(self.clients):
            try:
                report[port] = bool(self.clients[port].health_check(force=True))
            except (VLLMClientError, OSError, ValueError):      # a probe must not take the whole report down
                report[port] = False
        return report

    def get_client_from_worker_name(self, worker_name: str, **client_kwargs) -> VLLMClient:
        """Celery worker host names carry their server's port: ``worker_port_{port}_{worker_index}@hostname``
        (bulk_processing/scripts/start_multiple_celery_workers.sh)."""
        found = _WORKER_PORT.search(worker_name.split("@", 1)[0]) if isinstance(worker_name, str) else None
        if found is None:
            raise VLLMClientError(f"Invalid worker name format: {worker_name}. "
                                  "Expected format: worker_port_{port}_{worker_index}@hostname")
        return self.get_client(int(found.group(1)), **client_kwargs)


client_manager = VLLMClientManager()


def get_vllm_client_for_worker(worker_name: str, **kwargs) -> VLLMClient:
    manager = client_manager
    return manager.get_client_from_worker_name(worker_name, **kwargs)


# ----------------------------------------------------------------------------- llm_clients surface
# The reference's data tooling talks to models through ``BaseLLM.completion(prompt, structured_object)``
# (/root/reference/karanta/llm_clients/base.py:62-72), an async method in the concrete clients returning
# ``ModelCompletion(generation, model)`` objects (base.py:11-32; litellm_client.py:38-45: ``temperature`` default 1.0,
# ``max_tokens`` default 512, ``structured_object`` sent as ``response_format``, the content ``json.loads``-ed when a
# structure was asked for, ``ValueError("Error decoding response: ...")`` otherwise).  Same shape here, served by the
# MI355X engine; a structured request is enforced on the device (guided decoding), so the decode step cannot fail on
# a complete answer.
import asyncio  # noqa: E402
from dataclasses import asdict, dataclass  # noqa: E402


@dataclass
class ModelCompletion:
    generation: Any          # dict / list when a structure was requested, else the text
    model: str

    def to_json(self) -> str:
        return json.dumps(asdict(self), ensure_ascii=False)

    def to_dict(self) -> dict:
        return asdict(self)


class KarantaLLM:
    """``BaseLLM``-shaped client of one engine server (in-process LocalServer registered for the port, or HTTP)."""

    def __init__(self, model_name: str = "karantaocr", port: int = 8000, host: str = "localhost", **client_kwargs):
        self.model_name = model_name
        self._client = VLLMClient(port=port, host=host, **client_kwargs)

    async def completion(self, prompt, structured_object: Optional[Any] = None, **generation_kwargs: Any) -> List[ModelCompletion]:
        assert isinstance(prompt, list) and prompt, "Prompt must be a non-empty list"
        assert isinstance(prompt[0], (dict, list)), "Prompt must be a list of dictionaries or a list of lists of dictionaries"
        temperature = generation_kwargs.get("temperature", 1.0)
        max_tokens = generation_kwargs.get("max_tokens", 512)
        conversations = prompt if isinstance(prompt[0], list) else [prompt]
        loop = asyncio.get_running_loop()

        def one(messages):
            r = self._client.generate(messages, model=self.model_name, max_tokens=max_tokens, temperature=temperature,
                                      response_format=structured_object)
            if not structured_object:
                return ModelCompletion(generation=r["text"], model=self.model_name)
            try:
                return ModelCompletion(generation=json.loads(r["text"]), model=self.model_name)
            except json.JSONDecodeError:
                raise ValueError(f"Error decoding response: {r['text']}")

        # the conversations of a batch go out together: the server batches them (continuous or static)
        return list(await asyncio.gather(*[loop.run_in_executor(None, one, m) for m in conversations]))
