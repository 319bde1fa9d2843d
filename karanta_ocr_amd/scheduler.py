"""Continuous batching over the engine's fixed decode slots.

The reference leaves scheduling to vLLM (one server per GPU, `scripts/start_multiple_vllm_servers.sh:283`; the
pipeline only watches its `Running: n reqs, Waiting: m reqs` lines, `karanta/pipeline.py:769-800`).  Here the decode
hipGraph always steps all `max_batch` slots; a sequence that hits EOS (device flag) or its `max_tokens` (host) frees
its slot, and the next waiting request is prefilled into that slot while the other sequences keep their state —
instead of the whole batch waiting for its longest member (`Engine.generate`, static batching).

Only the host logic lives here; it drives `Engine.begin_slots / admit / decode_steps / poll_slots / slot_tokens /
retire`, and any object with those six methods (the CPU tests use a fake) can stand in for the engine.
"""
from __future__ import annotations

import collections
import os
import time
from dataclasses import dataclass
from typing import Any, Deque, Dict, Iterable, List, Optional, Sequence

import numpy as np


@dataclass
class SlotRequest:
    page: Any                 # engine.PageRequest
    max_tokens: int
    tag: Any = None           # returned untouched with the result


@dataclass
class SlotResult:
    tag: Any
    tokens: np.ndarray        # generated ids, EOS included when finish_reason == "stop"
    finish_reason: str        # "stop" | "length"
    prompt_tokens: int
    error: Optional[str] = None
    request: Optional[SlotRequest] = None   # the request this answers
    status: int = 500                       # HTTP-style class of `error`: 400 = the request itself cannot be served
    logprobs: Optional[Dict[str, np.ndarray]] = None   # when the page asked for them (Engine.slot_logprobs)


class SlotScheduler:
    """`submit()` requests, call `step()` until `idle`; every step admits what fits, runs `chunk` decode steps and
    returns the requests that finished."""

    def __init__(self, engine, max_tokens_cap: int, chunk: int = 8, eos_token_ids: Optional[Sequence[int]] = None,
                 max_prompt_tokens: Optional[int] = None, max_patches: Optional[int] = None, sampling: bool = False,
                 overlap: bool = False, guided: bool = False, logprobs: Optional[int] = None, admit_min: int = 1,
                 admit_max_wait: int = 4, launch_ahead: bool = True):
        if max_tokens_cap < 1 or chunk < 1:
            raise ValueError("max_tokens_cap and chunk must be >= 1")
        self.engine = engine
        self.n_slots = int(engine.B)
        self.cap, self.chunk = int(max_tokens_cap), int(chunk)
        self.eos = set(int(e) for e in (eos_token_ids if eos_token_ids is not None else engine.cfg.eos_token_ids))
        self.max_prompt_tokens = max_prompt_tokens if max_prompt_tokens is not None else getattr(engine, "max_tokens", None)
        self.max_patches = max_patches if max_patches is not None else getattr(engine, "max_patches", None)
        self.waiting: Deque[SlotRequest] = collections.deque()
        self.active: Dict[int, SlotRequest] = {}     # slot -> request
        self.prompt_len: Dict[int, int] = {}
        # overlap: ViT + prefill of an admission run on a second stream while the other slots keep decoding
        # (Engine.admit_begin / admit_ready / admit_end); one admission in flight at a time
        self.overlap = bool(overlap) and all(hasattr(engine, m) for m in ("admit_begin", "admit_ready", "admit_end"))
        self._inflight = None                        # (handle, requests, slots)
        # launch-ahead (engines with snapshot_slots / read_snapshot): the NEXT decode chunk is queued before the host waits for the
        # previous chunk's slot flags, so harvesting, the server loop and the next admission's host work run while the GPU decodes.
        # A finished slot is seen one chunk later (a slot may run up to 2 * chunk - 1 steps past its limit), so it goes with SHORT
        # chunks: measured on the corpus run (profiles/r04_corpus_launch_ahead.txt) chunk 8 loses 1 % to the r3 loop (slot occupancy
        # 0.86 -> 0.84), chunks of 2 gain 2 % (29.6-29.9 against 29.15 pages/s) — the server's defaults.
        self.launch_ahead = (bool(launch_ahead) and not self.overlap
                             and all(hasattr(engine, m) for m in ("snapshot_slots", "read_snapshot")))
        self.over = (2 if self.launch_ahead else 1) * int(chunk)
        self._snap = None                            # (sequence number, handle) of the chunk whose flags are read next
        self._snap_seq = 0                           # snapshots taken so far
        self._adm_seq: Dict[int, int] = {}           # slot -> snapshots taken when its request was admitted
        # Admission batching: while sequences are decoding, wait until `admit_min` slots are free (and as many requests
        # wait) before interrupting the decode graph with an admission — one ViT + prefill over several pages runs its
        # GEMMs at several times the rows of a single page — but never longer than `admit_max_wait` scheduler steps.
        self.admit_min, self.admit_max_wait = max(1, int(admit_min)), max(0, int(admit_max_wait))
        self._held = 0                               # scheduler steps the oldest admissible request has been held back
        self.steps = 0                               # decode steps run
        # host wall time by phase (seconds): admission launches, decode-chunk launches, the wait for the GPU in _harvest
        self.phase_s = {"admit": 0.0, "decode_launch": 0.0, "harvest": 0.0}
        self.admissions = 0                          # admission rounds that carried pages
        self.pages_admitted = 0
        self.slot_steps_busy = 0                     # sum over steps of occupied slots (utilisation numerator)
        # a slot may run up to chunk - 1 steps past its limit before the host looks: size the history for that
        # sampling: pages may carry temperature > 0 (the decode graph then includes the Gumbel-max pass)
        # guided: pages may carry a pattern (masked sampling pass + DFA advance in the graph); logprobs = k: every step
        # records log-probabilities (top-k alternatives) for the pages that ask
        self.logprobs = logprobs
        kw = {}
        if sampling:
            kw["sampling"] = True
        if guided:
            kw["guided"] = True
        if logprobs is not None:
            kw["logprobs"] = int(logprobs)
        engine.begin_slots(self.cap + self.over, **kw)
        # engines that can switch the sampling / guided passes of the decode graph off while no request in a slot needs them
        self._features = ((sampling or guided) and hasattr(engine, "set_step_features")
                          and os.environ.get("KARANTA_STEP_FEATURES", "1") == "1")
        self._features_now = None

    # ------------------------------------------------------------------ public
    def submit(self, req: SlotRequest) -> None:
        if req.max_tokens < 1:
            raise ValueError("max_tokens must be >= 1")
        self.waiting.append(req)

    @property
    def idle(self) -> bool:
        return not self.waiting and not self.active and self._inflight is None

    @property
    def running(self) -> int:
        return len(self.active)

    def step(self) -> List[SlotResult]:
        if self.launch_ahead:
            return self._step_ahead()
        if not self.overlap:
            t0 = time.perf_counter()
            done: List[SlotResult] = self._admit()
            t1 = time.perf_counter()
            self.phase_s["admit"] += t1 - t0
            if self.active:
                self._decode_chunk()
                t2 = time.perf_counter()
                done += self._harvest()
                self.phase_s["decode_launch"] += t2 - t1
                self.phase_s["harvest"] += time.perf_counter() - t2
            return done
        # overlapped: finish an admission that is through, queue the decode chunk, THEN spend host time launching the
        # next admission (the GPU has both to run), then look at the finished flags
        done = self._finish_admission(block=not self.active)
        if self.active:
            self._decode_chunk()
        if self._inflight is None:
            done += self._admit(begin_only=True)
        if self.active:
            done += self._harvest()
        return done

    def _step_ahead(self) -> List[SlotResult]:
        done: List[SlotResult] = []
        t0 = time.perf_counter()
        if self.active:
            if self._snap is None:                   # nothing queued yet: this chunk's flags are the first to be read
                self._decode_chunk()
                self._snap = self._take_snapshot()
            self._decode_chunk()                     # the GPU's work while the host does everything below
            nxt = self._take_snapshot()
            t1 = time.perf_counter()
            fin, gen = self.engine.read_snapshot(self._snap[1])
            done += self._harvest((fin, gen), self._snap[0])
            self._snap = nxt if self.active else None
            t2 = time.perf_counter()
            self.phase_s["decode_launch"] += t1 - t0
            self.phase_s["harvest"] += t2 - t1
            t0 = t2
        done += self._admit()                        # queued behind the chunk that is executing
        self.phase_s["admit"] += time.perf_counter() - t0
        return done

    def _take_snapshot(self):
        seq = self._snap_seq
        self._snap_seq += 1
        return seq, self.engine.snapshot_slots()

    def _decode_chunk(self):
        if self._features:      # the passes the steps carry follow the requests that are in the slots right now
            need_g = any(getattr(r.page, "guide", None) is not None for r in self.active.values())
            need_s = need_g or any(float(getattr(r.page, "temperature", 0.0) or 0.0) > 0 for r in self.active.values())
            if (need_s, need_g) != self._features_now:
                self.engine.set_step_features(need_s, need_g)
                self._features_now = (need_s, need_g)
        self.engine.decode_steps(self.chunk)
        self.steps += self.chunk
        self.slot_steps_busy += self.chunk * len(self.active)

    def _finish_admission(self, block: bool) -> List[SlotResult]:
        if self._inflight is None:
            return []
        handle, batch, slots = self._inflight
        if not block and not self.engine.admit_ready(handle):
            return []
        self._inflight = None
        try:
            lens = self.engine.admit_end(handle)
        except Exception as e:
            return [self._failure(r, f"{type(e).__name__}: {e}") for r in batch]
        for r, j, n in zip(batch, slots, lens):
            self.active[j] = r
            self.prompt_len[j] = int(n)
        return []

    def run(self, requests: Iterable[SlotRequest]) -> List[SlotResult]:
        """All requests to completion; results in submission order."""
        reqs = list(requests)
        order = {id(r): i for i, r in enumerate(reqs)}
        for r in reqs:
            self.submit(r)
        out: List[Optional[SlotResult]] = [None] * len(reqs)
        left = len(reqs)
        while left:
            for res in self.step():
                i = order.get(id(res.request))
                if i is not None and out[i] is None:
                    out[i] = res
                    left -= 1
        return out  # type: ignore[return-value]

    # ------------------------------------------------------------------ internals
    def _admit(self, begin_only: bool = False) -> List[SlotResult]:
        free = [j for j in range(self.n_slots) if j not in self.active]
        if self.active and self.waiting and free and self.admit_min > 1:
            ready = min(len(free), len(self.waiting))
            if ready < min(self.admit_min, self.n_slots) and self._held < self.admit_max_wait:
                self._held += 1
                return []
        self._held = 0
        batch: List[SlotRequest] = []
        tok_budget = self.max_prompt_tokens
        patch_budget = self.max_patches
        failed: List[SlotResult] = []
        room = self.engine.seq_room() if hasattr(self.engine, "seq_room") else None
        while self.waiting and len(batch) < len(free):
            r = self.waiting[0]
            n_tok = int(len(r.page.input_ids))
            if room is not None and n_tok + self._budget(r) > room:
                # its own prompt + max_tokens (+ the chunk overshoot) exceed a sequence's cache rows: this request fails
                # alone, as a client error; whatever is admitted with it is unaffected
                self.waiting.popleft()
                failed.append(self._failure(r, f"prompt ({n_tok} tokens) + max_tokens ({min(int(r.max_tokens), self.cap)}) + "
                                               f"{self.over} scheduler steps exceed the sequence capacity {room}", status=400))
                continue
            n_patch = sum(int(np.prod(g)) for g in getattr(r.page, "grids", None) or [])   # from the grids: pages may
            #                                                    carry uint8 images (GPU front end) instead of patches
            over_tok = tok_budget is not None and n_tok > tok_budget
            over_patch = patch_budget is not None and n_patch > patch_budget
            if over_tok or over_patch:
                if not batch and (self.max_prompt_tokens is not None and n_tok > self.max_prompt_tokens
                                  or self.max_patches is not None and n_patch > self.max_patches):
                    # can never fit: fail it instead of blocking the queue
                    self.waiting.popleft()
                    failed.append(self._failure(r, f"request does not fit the engine ({n_tok} prompt tokens, {n_patch} patches)",
                                                status=400))
                    continue
                break  # fits an emptier admission round
            self.waiting.popleft()
            batch.append(r)
            if tok_budget is not None:
                tok_budget -= n_tok
            if patch_budget is not None:
                patch_budget -= n_patch
        if batch and begin_only:
            slots = free[:len(batch)]
            try:
                self._inflight = (self.engine.admit_begin([r.page for r in batch], slots, **self._budget_kw(batch)), batch, slots)
            except Exception as e:
                return failed + [self._failure(r, f"{type(e).__name__}: {e}") for r in batch]
        elif batch:
            slots = free[:len(batch)]
            try:
                lens = self.engine.admit([r.page for r in batch], slots, **self._budget_kw(batch))
            except Exception as e:  # the admission as a whole failed: none of these requests entered a slot
                return failed + [self._failure(r, f"{type(e).__name__}: {e}") for r in batch]
            for r, j, n in zip(batch, slots, lens):
                self.active[j] = r
                self.prompt_len[j] = int(n)
                self._adm_seq[j] = self._snap_seq    # snapshots taken from now on see this request in the slot
            self.admissions += 1
            self.pages_admitted += len(batch)
        return failed

    def _budget(self, r: SlotRequest) -> int:
        """Cache rows a request may write after its prompt: its token limit plus the steps a slot can run past it
        before the host looks at the flags again."""
        return min(int(r.max_tokens), self.cap) + self.over

    def _budget_kw(self, batch) -> dict:
        # engines that check capacity per request take the budgets (test fakes without seq_room do not)
        return {"budgets": [self._budget(r) for r in batch]} if hasattr(self.engine, "seq_room") else {}

    def _failure(self, r: SlotRequest, msg: str, status: int = 500) -> SlotResult:
        return SlotResult(r.tag, np.zeros(0, np.int64), "length", 0, error=msg, request=r, status=status)

    def _harvest(self, flags=None, seq: Optional[int] = None) -> List[SlotResult]:
        """Requests that finished.  flags / seq: a snapshot's (finished, generated) and its sequence number — slots admitted after it
        was taken still show their previous occupant there and are skipped."""
        fin, gen = flags if flags is not None else self.engine.poll_slots()
        out: List[SlotResult] = []
        for j in sorted(self.active):
            if seq is not None and self._adm_seq.get(j, 0) > seq:
                continue
            r = self.active[j]
            limit = min(int(r.max_tokens), self.cap)
            if not (fin[j] or gen[j] >= limit):
                continue
            n = int(min(gen[j], limit))
            toks = np.asarray(self.engine.slot_tokens(j, n), np.int64)
            reason = "length"
            hit = np.flatnonzero(np.isin(toks, list(self.eos))) if self.eos else np.zeros(0, np.int64)
            if hit.size:
                toks, reason = toks[: int(hit[0]) + 1], "stop"
            if not fin[j]:
                self.engine.retire(j)
            lps = None
            k = getattr(r.page, "logprobs", None)
            if k is not None and self.logprobs is not None:
                n_lp = len(toks) - (1 if reason == "stop" else 0)     # the EOS step records nothing
                lps = self.engine.slot_logprobs(j, n_lp, int(k))
            out.append(SlotResult(r.tag, toks, reason, self.prompt_len.pop(j), request=r, logprobs=lps))
            del self.active[j]
        return out
