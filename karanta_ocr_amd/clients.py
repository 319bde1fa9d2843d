"""Drop-in for the reference's ``bulk_processing/workers/vllm_client.py``: same class names, method
names, argument meaning, result-dict schema and error behaviour — with the MI355X engine behind it.

Reference surface mirrored (file:line in /root/reference/bulk_processing/workers/vllm_client.py):
``VLLMClientError`` :14 · ``VLLMClient.__init__`` :28-74 · ``health_check`` :76-110 (60 s cache) ·
``get_server_info`` :112-153 · ``generate`` :155-227 (health check, default model = first served model,
``max_retries + 1`` attempts with ``retry_delay * 2**attempt`` back-off, then ``VLLMClientError``) ·
``_process_response`` :229-266 (result schema) · ``batch_generate`` :268-296 · ``VLLMClientManager``
:304-386 (``worker_port_{port}_{i}@host`` parsing) · ``get_vllm_client_for_worker`` :393-404.

Transport: when a :class:`karanta_ocr_amd.serving.LocalServer` is registered for the client's
port the call is in-process (no HTTP, no OpenAI SDK); otherwise the same three endpoints are reached
over plain HTTP (stdlib), i.e. the HTTP shim of ``serving.serve_http`` or any OpenAI-compatible server.
"""
from __future__ import annotations

import json
import logging
import time
import urllib.error
import urllib.request
from typing import Any, Dict, List, Optional

from .serving import local_server

logger = logging.getLogger(__name__)


class VLLMClientError(Exception):
    """Custom exception for VLLM client errors"""


class VLLMClient:
    def __init__(self, port: int, host: str = "localhost", api_key: str = "EMPTY", timeout: float = 300.0,
                 max_retries: int = 3, retry_delay: float = 1.0, health_check_timeout: float = 30.0):
        self.port = port
        self.host = host
        self.api_key = api_key
        self.timeout = timeout
        self.max_retries = max_retries
        self.retry_delay = retry_delay
        self.health_check_timeout = health_check_timeout
        self.base_url = f"http://{self.host}:{self.port}/v1"
        self.health_url = f"http://{self.host}:{self.port}/health"
        self._server_info = None
        self._last_health_check = 0
        self._health_check_interval = 60  # seconds, as the reference
        logger.info(f"Initialized VLLM client for {self.base_url}")

    # ------------------------------------------------------------------ transport
    def _call(self, method: str, path: str, body: Optional[dict] = None, timeout: Optional[float] = None):
        srv = local_server(self.host, self.port)
        if srv is not None:
            if path == "/health":
                return srv.health()
            if path == "/v1/models":
                return srv.models()
            return srv.chat_completions(body)
        url = f"http://{self.host}:{self.port}{path}"
        data = json.dumps(body).encode() if body is not None else None
        req = urllib.request.Request(url, data=data, method=method)
        req.add_header("Content-Type", "application/json")
        if self.api_key:
            req.add_header("Authorization", f"Bearer {self.api_key}")
        try:
            with urllib.request.urlopen(req, timeout=timeout or self.timeout) as r:
                raw = r.read()
                return r.status, (json.loads(raw) if raw else {})
        except urllib.error.HTTPError as e:
            raw = e.read()
            try:
                return e.code, json.loads(raw)
            except Exception:
                return e.code, {"error": {"message": raw.decode("utf-8", "replace")}}

    # ------------------------------------------------------------------ reference API
    def health_check(self, force: bool = False) -> bool:
        current_time = time.time()
        if not force and (current_time - self._last_health_check) < self._health_check_interval:
            return True
        try:
            status, _ = self._call("GET", "/health", timeout=self.health_check_timeout)
            is_healthy = status == 200
            self._last_health_check = current_time
            if not is_healthy:
                logger.warning(f"VLLM server health check failed: {status}")
            return is_healthy
        except Exception as e:  # connection refused etc. (requests.RequestException in the reference)
            logger.error(f"VLLM server health check failed: {e}")
            return False

    def get_server_info(self, force_refresh: bool = False) -> Dict[str, Any]:
        if self._server_info is None or force_refresh:
            try:
                status, body = self._call("GET", "/v1/models")
                if status != 200:
                    raise RuntimeError(f"GET /v1/models -> {status}")
                models = [m["id"] for m in body.get("data", [])]
                info = {"models": models, "base_url": self.base_url, "port": self.port, "host": self.host,
                        "last_updated": time.time()}
                try:
                    hs, hb = self._call("GET", "/health", timeout=5)
                    if hs == 200 and isinstance(hb, dict):
                        info.update(hb)
                except Exception:
                    pass
                self._server_info = info
                logger.info(f"Retrieved server info: {len(models)} models available")
            except Exception as e:
                logger.error(f"Failed to get server info: {e}")
                raise VLLMClientError(f"Failed to get server info: {e}")
        return self._server_info

    def generate(self, messages: List[Dict[str, str]], model: Optional[str] = None, max_tokens: int = 100,
                 temperature: float = 0.7, response_format: Optional[Dict[str, Any]] = None, **kwargs) -> Dict[str, Any]:
        if not self.health_check():
            raise VLLMClientError(f"VLLM server at {self.base_url} is not healthy")
        if model is None:
            server_info = self.get_server_info()
            if not server_info.get("models"):
                raise VLLMClientError("No models available on server")
            model = server_info["models"][0]
            logger.debug(f"Using default model: {model}")
        generation_params = {"model": model, "messages": messages, "max_tokens": max_tokens, "temperature": temperature,
                             "response_format": response_format, **kwargs}
        last_exception = None
        start_time = time.time()
        for attempt in range(self.max_retries + 1):
            try:
                logger.debug(f"Generation attempt {attempt + 1}/{self.max_retries + 1}")
                status, body = self._call("POST", "/v1/chat/completions", generation_params)
                if status != 200:
                    raise RuntimeError(f"Error code: {status} - {body.get('error', body)}")
                return self._process_response(body, start_time, generation_params)
            except Exception as e:
                last_exception = e
                logger.warning(f"Generation attempt {attempt + 1} failed: {e}")
                if attempt < self.max_retries:
                    time.sleep(self.retry_delay * (2 ** attempt))
                    continue
                break
        error_msg = f"Generation failed after {self.max_retries + 1} attempts. Last error: {last_exception}"
        logger.error(error_msg)
        raise VLLMClientError(error_msg)

    def _process_response(self, response: dict, start_time: float, generation_params: Dict[str, Any]) -> Dict[str, Any]:
        end_time = time.time()
        if not response.get("choices"):
            raise VLLMClientError("No choices returned from VLLM server")
        choice = response["choices"][0]
        usage = response.get("usage") or {}
        return {
            "text": choice["message"]["content"],
            "finish_reason": choice.get("finish_reason"),
            "model": response.get("model"),
            "usage": {"prompt_tokens": usage.get("prompt_tokens", 0), "completion_tokens": usage.get("completion_tokens", 0),
                      "total_tokens": usage.get("total_tokens", 0)},
            "metadata": {
                "generation_time": end_time - start_time,
                "server_url": self.base_url,
                "generation_params": {k: v for k, v in generation_params.items() if k not in ["messages"]},
                "timestamp": end_time,
            },
        }

    def batch_generate(self, prompts: List[Any], **generation_kwargs) -> List[Dict[str, Any]]:
        """One ``generate`` per prompt, errors captured per item.  (The reference passes each element
        straight through as ``messages``, vllm_client.py:286 — so elements are message lists; a bare
        string is wrapped into a single user message here instead of failing server-side.)"""
        results = []
        for i, prompt in enumerate(prompts):
            try:
                messages = [{"role": "user", "content": prompt}] if isinstance(prompt, str) else prompt
                result = self.generate(messages, **generation_kwargs)
                result["metadata"]["batch_index"] = i
                results.append(result)
            except Exception as e:
                logger.error(f"Failed to process prompt {i + 1}: {e}")
                results.append({"error": str(e), "metadata": {"batch_index": i, "failed": True}})
        return results

    def __repr__(self) -> str:
        return f"VLLMClient(host={self.host}, port={self.port}, base_url={self.base_url})"


class VLLMClientManager:
    def __init__(self, server_config: Dict[int, str] = None):
        self.server_config = server_config or {}
        self.clients: Dict[int, VLLMClient] = {}

    def get_client(self, port: int, **client_kwargs) -> VLLMClient:
        if port not in self.clients:
            host = self.server_config.get(port, "localhost")
            self.clients[port] = VLLMClient(port=port, host=host, **client_kwargs)
            logger.info(f"Created VLLM client for port {port}")
        return self.clients[port]

    def health_check_all(self) -> Dict[int, bool]:
        results = {}
        for port, client in self.clients.items():
            try:
                results[port] = client.health_check(force=True)
            except Exception as e:
                logger.error(f"Health check failed for port {port}: {e}")
                results[port] = False
        return results

    def get_client_from_worker_name(self, worker_name: str, **client_kwargs) -> VLLMClient:
        try:
            parts = worker_name.split("_")
            port_index = parts.index("port")
            port = int(parts[port_index + 1])
            return self.get_client(port, **client_kwargs)
        except (ValueError, IndexError):
            raise VLLMClientError(
                f"Invalid worker name format: {worker_name}. Expected format: worker_port_{{port}}_{{worker_index}}@hostname")


client_manager = VLLMClientManager()


def get_vllm_client_for_worker(worker_name: str, **kwargs) -> VLLMClient:
    return client_manager.get_client_from_worker_name(worker_name, **kwargs)


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
