"""Slot scheduler (continuous batching) host logic against a fake engine that follows the Engine slot API."""
import numpy as np
import pytest

from karanta_ocr_amd.scheduler import SlotRequest, SlotScheduler


class Page:
    def __init__(self, ids, n_patches=0):
        self.input_ids = np.asarray(ids)
        self.pixel_values = np.zeros((n_patches, 4), np.float32) if n_patches else None
        self.grids = [(1, 1, n_patches)] if n_patches else []


class FakeEngine:
    """Deterministic stand-in: slot j's sequence for a prompt is script[prompt[0]] (list of token ids)."""
    class cfg:
        eos_token_ids = (99,)

    def __init__(self, n_slots, script, max_tokens=1000, max_patches=1000):
        self.B, self.script, self.max_tokens, self.max_patches = n_slots, script, max_tokens, max_patches
        self.log = []
        self.fail_admit = False

    def begin_slots(self, max_new, sampling=False):
        self.max_new = max_new
        self.seq = [None] * self.B
        self.gen = [0] * self.B
        self.fin = [True] * self.B
        self.hist = [[] for _ in range(self.B)]

    def admit(self, pages, slots):
        if self.fail_admit:
            raise RuntimeError("boom")
        assert len(set(slots)) == len(slots) and all(self.fin[j] for j in slots), "admitted into a busy slot"
        assert sum(len(p.input_ids) for p in pages) <= self.max_tokens
        self.log.append(("admit", tuple(slots)))
        for p, j in zip(pages, slots):
            self.seq[j] = list(self.script[int(p.input_ids[0])])
            self.hist[j], self.gen[j], self.fin[j] = [], 0, False
            self._emit(j)
        return [len(p.input_ids) for p in pages]

    def _emit(self, j):
        if self.fin[j]:
            return                                   # frozen
        tok = self.seq[j][self.gen[j]] if self.gen[j] < len(self.seq[j]) else 7
        assert self.gen[j] < self.max_new + 1, "history overflow"
        self.hist[j].append(tok)
        self.gen[j] += 1
        if tok == 99:
            self.fin[j] = True

    def decode_steps(self, n):
        self.log.append(("steps", n, sum(not f for f in self.fin)))
        for _ in range(n):
            for j in range(self.B):
                self._emit(j)

    def poll_slots(self):
        return np.asarray(self.fin), np.asarray(self.gen)

    def slot_tokens(self, j, n):
        return np.asarray(self.hist[j][:n])

    def retire(self, j):
        self.fin[j] = True


SCRIPT = {0: [1, 2, 3, 99], 1: [5] * 40, 2: [99], 3: [4, 4, 99, 8, 8], 4: [6] * 7 + [99], 5: [9] * 100}


def test_results_match_the_scripts_and_keep_submission_order():
    eng = FakeEngine(2, SCRIPT)
    sch = SlotScheduler(eng, max_tokens_cap=32, chunk=4)
    reqs = [SlotRequest(Page([k, 0, 0]), mt, tag=f"r{k}") for k, mt in [(0, 10), (1, 9), (2, 5), (3, 30), (4, 30), (5, 32)]]
    res = sch.run(reqs)
    assert [r.tag for r in res] == ["r0", "r1", "r2", "r3", "r4", "r5"]
    assert res[0].tokens.tolist() == [1, 2, 3, 99] and res[0].finish_reason == "stop"
    assert res[1].tokens.tolist() == [5] * 9 and res[1].finish_reason == "length"
    assert res[2].tokens.tolist() == [99] and res[2].finish_reason == "stop"          # EOS as the very first token
    assert res[3].tokens.tolist() == [4, 4, 99] and res[3].finish_reason == "stop"    # nothing after the EOS
    assert res[4].tokens.tolist() == [6] * 7 + [99]
    assert res[5].tokens.tolist() == [9] * 32 and res[5].finish_reason == "length"    # the scheduler's cap
    assert all(r.prompt_tokens == 3 and r.error is None for r in res)
    assert sch.idle and sch.running == 0


def test_slots_are_refilled_while_others_keep_decoding():
    eng = FakeEngine(2, SCRIPT)
    sch = SlotScheduler(eng, max_tokens_cap=64, chunk=4)
    for k, mt in [(5, 60), (0, 10), (3, 10), (4, 20)]:
        sch.submit(SlotRequest(Page([k]), mt, tag=k))
    done = []
    while not sch.idle:
        done += sch.step()
    admits = [e for e in eng.log if e[0] == "admit"]
    assert admits[0] == ("admit", (0, 1))                 # both slots filled at once
    assert all(a[1] == (1,) for a in admits[1:])          # the long request keeps slot 0; slot 1 turns over
    assert [r.tag for r in done] == [0, 3, 4, 5]          # completion order, not submission order
    # static batching would have idled slot 1 for the whole 60-token request
    assert sch.slot_steps_busy / (sch.steps * 2) > 0.6   # three short requests shared slot 1 while slot 0 ran the long one


def test_length_limit_between_polls_never_overruns_history():
    eng = FakeEngine(1, SCRIPT)
    sch = SlotScheduler(eng, max_tokens_cap=10, chunk=7)     # polls at 8, 15 generated tokens
    (r,) = sch.run([SlotRequest(Page([5]), 10)])
    assert r.tokens.tolist() == [9] * 10 and r.finish_reason == "length"
    assert eng.max_new == 17 and max(eng.gen) <= 17


def test_prompt_budget_splits_admissions_and_rejects_what_can_never_fit():
    eng = FakeEngine(3, SCRIPT, max_tokens=10)
    sch = SlotScheduler(eng, max_tokens_cap=8, chunk=2)
    reqs = [SlotRequest(Page([0] * 6), 4, "a"), SlotRequest(Page([0] * 6), 4, "b"), SlotRequest(Page([0] * 11), 4, "big"),
            SlotRequest(Page([2] * 3), 4, "c")]
    res = sch.run(reqs)
    assert [e[1] for e in eng.log if e[0] == "admit"][0] == (0,)       # 6 + 6 > 10: one per admission round
    big = res[2]
    assert big.error and "does not fit" in big.error and big.tokens.size == 0
    assert [r.error for r in res[:2]] == [None, None] and res[3].tokens.tolist() == [99]


def test_engine_failure_fails_only_that_admission():
    eng = FakeEngine(2, SCRIPT)
    sch = SlotScheduler(eng, max_tokens_cap=8, chunk=2)
    eng.fail_admit = True
    sch.submit(SlotRequest(Page([0]), 4, "x"))
    (r,) = sch.step()
    assert r.error.startswith("RuntimeError") and sch.idle
    eng.fail_admit = False
    (ok,) = sch.run([SlotRequest(Page([0]), 8, "y")])
    assert ok.error is None and ok.tokens.tolist() == [1, 2, 3, 99]


def test_argument_checks():
    eng = FakeEngine(1, SCRIPT)
    with pytest.raises(ValueError):
        SlotScheduler(eng, 0)
    sch = SlotScheduler(eng, 4)
    with pytest.raises(ValueError):
        sch.submit(SlotRequest(Page([0]), 0))


def test_patch_budget_counts_grids():
    eng = FakeEngine(2, SCRIPT, max_patches=10)
    sch = SlotScheduler(eng, max_tokens_cap=8, chunk=2)
    res = sch.run([SlotRequest(Page([0], n_patches=6), 4, "a"), SlotRequest(Page([0], n_patches=6), 4, "b"),
                   SlotRequest(Page([0], n_patches=11), 4, "big")])
    assert [e[1] for e in eng.log if e[0] == "admit"][0] == (0,)      # 6 + 6 > 10 patches: one page per admission
    assert res[0].error is None and res[1].error is None and "does not fit" in res[2].error


class FakeAsyncEngine(FakeEngine):
    """FakeEngine + the overlapped-admission API: an admission becomes ready after `latency` polls."""
    latency = 2

    def admit_begin(self, pages, slots):
        assert all(self.fin[j] for j in slots), "admission into a busy slot"
        self.log.append(("begin", tuple(slots)))
        return {"pages": pages, "slots": list(slots), "polls": 0}

    def admit_ready(self, h):
        h["polls"] += 1
        return h["polls"] > self.latency

    def admit_end(self, h):
        self.log.append(("end", tuple(h["slots"])))
        return FakeEngine.admit(self, h["pages"], h["slots"])


def test_overlapped_admission_same_results_and_decode_keeps_running():
    reqs = lambda: [SlotRequest(Page([k, 0]), mt, tag=f"r{k}") for k, mt in [(5, 40), (0, 10), (3, 10), (4, 20), (1, 9), (2, 5)]]
    plain = SlotScheduler(FakeEngine(2, SCRIPT), max_tokens_cap=64, chunk=3).run(reqs())
    eng = FakeAsyncEngine(2, SCRIPT)
    sch = SlotScheduler(eng, max_tokens_cap=64, chunk=3, overlap=True)
    assert sch.overlap
    over = sch.run(reqs())
    for a, b in zip(plain, over):
        assert a.tag == b.tag and a.tokens.tolist() == b.tokens.tolist() and a.finish_reason == b.finish_reason
    # decode chunks ran between the begin and the end of at least one admission (the other slot kept going)
    kinds = [e[0] for e in eng.log]
    i = kinds.index("begin", 1)
    assert "steps" in kinds[i:kinds.index("end", i)]
    assert sch.idle and sch._inflight is None
    # an engine without the async API silently falls back
    assert SlotScheduler(FakeEngine(2, SCRIPT), 8, overlap=True).overlap is False


def test_overlapped_admission_failure_paths():
    class Boom(FakeAsyncEngine):
        def admit_end(self, h):
            raise RuntimeError("late boom")
    sch = SlotScheduler(Boom(2, SCRIPT), max_tokens_cap=8, chunk=2, overlap=True)
    (r,) = sch.run([SlotRequest(Page([0]), 4, "x")])
    assert r.error.startswith("RuntimeError: late boom") and sch.idle


class RoomEngine(FakeEngine):
    """FakeEngine that, like the real one, bounds prompt + generated tokens per sequence (Engine.seq_room) and takes the
    per-request budgets with the admission."""
    room = 40

    def seq_room(self):
        return self.room

    def admit(self, pages, slots, budgets=None):
        assert budgets is not None and len(budgets) == len(pages)
        for p, b in zip(pages, budgets):
            assert len(p.input_ids) + b <= self.room, "the scheduler let an oversized request through"
        self.budgets = list(budgets)
        return super().admit(pages, slots)


def test_oversized_request_fails_alone_with_a_client_error():
    """the bound is the request's OWN prompt + max_tokens (+ chunk), checked before admission; the
    request that cannot fit gets a 400-class failure and the requests admitted in the same round are served."""
    eng = RoomEngine(2, SCRIPT)
    sch = SlotScheduler(eng, max_tokens_cap=32, chunk=4)
    ok_small = SlotRequest(Page([0] + [0] * 9), 10, tag="fits")                 # 10 + 10 + 4 <= 40
    too_long = SlotRequest(Page([1] + [0] * 29), 9, tag="prompt too long")      # 30 + 9 + 4 > 40
    long_small = SlotRequest(Page([2] + [0] * 29), 5, tag="long prompt, small limit")   # 30 + 5 + 4 <= 40
    res = sch.run([ok_small, too_long, long_small])
    assert res[0].error is None and res[0].tokens.tolist() == [1, 2, 3, 99]
    assert res[1].error is not None and res[1].status == 400 and "capacity" in res[1].error
    assert res[2].error is None and res[2].tokens.tolist() == [99]
    assert ("admit", (0, 1)) in eng.log                                        # the two that fit went in together
    assert eng.budgets == [10 + 4, 5 + 4]


def test_admission_batching_waits_for_several_free_slots_but_not_forever():
    """admit_min = 2: while something decodes, one free slot alone does not trigger an admission (requests entering
    together share one ViT + prefill launch sequence); after admit_max_wait scheduler steps the wait ends; results are
    what they are without batching."""
    script = {0: [1] * 30, 1: [2, 99], 2: [3] * 8 + [99], 3: [4, 99], 4: [5, 5, 99]}
    reqs = lambda: [SlotRequest(Page([k, 0]), 40, tag=k) for k in range(5)]
    plain = SlotScheduler(FakeEngine(2, script), max_tokens_cap=40, chunk=2).run(reqs())
    eng = FakeEngine(2, script)
    sch = SlotScheduler(eng, max_tokens_cap=40, chunk=2, admit_min=2, admit_max_wait=3)
    res = sch.run(reqs())
    for a, b in zip(plain, res):
        assert a.tokens.tolist() == b.tokens.tolist() and a.finish_reason == b.finish_reason
    admits = [e[1] for e in eng.log if e[0] == "admit"]
    assert admits[0] == (0, 1)                                   # nothing decoding: no waiting
    # request 1 finishes at once; slot 1 is free while slot 0 decodes its 30 tokens: the next admission is held back
    # for 3 scheduler steps, then goes alone (only one slot can be free)
    steps_between = [e for e in eng.log[eng.log.index(("admit", (0, 1))) + 1:]]
    first_admit_after = next(i for i, e in enumerate(steps_between) if e[0] == "admit")
    assert first_admit_after == 4, steps_between[:6]            # 1 harvest step + 3 held steps of decoding before it
    eng2 = FakeEngine(4, script)
    sch2 = SlotScheduler(eng2, max_tokens_cap=40, chunk=2, admit_min=2, admit_max_wait=50)
    sch2.run(reqs())
    assert all(len(e[1]) >= 2 or i == len([x for x in eng2.log if x[0] == "admit"]) - 1
               for i, e in enumerate([x for x in eng2.log if x[0] == "admit"]))


class AheadEngine(FakeEngine):
    """FakeEngine with the snapshot API: the scheduler queues one decode chunk ahead of the flags it reads."""

    def begin_slots(self, max_new, sampling=False):
        super().begin_slots(max_new, sampling)
        self.snaps = 0

    def snapshot_slots(self):
        self.snaps += 1
        return np.asarray(self.fin).copy(), np.asarray(self.gen).copy()

    def read_snapshot(self, snap):
        return snap

    def poll_slots(self):
        raise AssertionError("launch-ahead reads snapshots, it never drains the stream")


def test_launch_ahead_gives_the_same_results_one_chunk_later():
    reqs = lambda: [SlotRequest(Page([k, 0, 0]), mt, tag=f"r{k}") for k, mt in [(0, 10), (1, 9), (2, 5), (3, 30), (4, 30), (5, 32)]]
    base = SlotScheduler(FakeEngine(2, SCRIPT), max_tokens_cap=32, chunk=4)
    assert not base.launch_ahead and base.over == 4
    want = base.run(reqs())
    eng = AheadEngine(2, SCRIPT)
    sch = SlotScheduler(eng, max_tokens_cap=32, chunk=4, launch_ahead=True)
    assert sch.launch_ahead and sch.over == 8 and eng.max_new == 32 + 8
    got = sch.run(reqs())
    for a, b in zip(got, want):
        assert a.tag == b.tag and a.tokens.tolist() == b.tokens.tolist() and a.finish_reason == b.finish_reason and a.error is None
    assert sch.idle and sch.running == 0 and eng.snaps >= 2
    # a slot re-filled after a snapshot was taken is not harvested from that snapshot (it still shows the previous occupant's EOS)
    eng = AheadEngine(1, SCRIPT)
    sch = SlotScheduler(eng, max_tokens_cap=64, chunk=2, launch_ahead=True)
    res = sch.run([SlotRequest(Page([2]), 5, tag="a"), SlotRequest(Page([4]), 20, tag="b"), SlotRequest(Page([5]), 50, tag="c")])
    assert [r.tokens.tolist() for r in res] == [[99], [6] * 7 + [99], [9] * 50]
    assert max(len(h) for h in eng.hist) <= 50 + 4             # at most 2 chunks past a limit
    # the default where the engine can do it; opt-out
    assert SlotScheduler(AheadEngine(1, SCRIPT), max_tokens_cap=8, chunk=2).launch_ahead
    assert not SlotScheduler(AheadEngine(1, SCRIPT), max_tokens_cap=8, chunk=2, launch_ahead=False).launch_ahead
