"""Engine-vs-oracle parity at PRODUCTION WIDTHS, and the BASELINE.json configurations the tiny-model tests never reach.

VERDICT r1 "Missing #1 / Next #1": the engine picks other kernel instantiations at real widths than on the toy configs
(dec_wide_kernel<.,24> / dec_narrow_kernel<..,24,2,..> at K = 1536, the K = 3584 ones, the deferred split-K chain, the
256x256 GEMM tiles at M >= 4900, attn_varlen_kernel<80> over a 4900-token segment with its lazy-rescale branch, the
8-split decode attention at contexts of ~1400).  Here the bench models run at their full widths and full vocabulary with
the depth truncated to what the oracle finishes in seconds (tests/prodwidth.py), call sequence of
/root/reference/karanta/training/test_trained_model.py:76-99 (processor -> generate -> ids):

  (a) Qwen2-VL-2B widths, one 1024x1024 scan (70x70 patches, 1225 image tokens): ViT merged output, last-position
      prefill logits, 12 greedy tokens with a MINIMUM number of decisive comparisons, eager = graph, batch of 3 = solo;
  (b) Qwen2-VL-7B widths, batch of 4 (BASELINE config 3's per-GPU share), bf16 and fp8 weights (config 5's dtype);
  (a') the bench model itself — Qwen2-VL-2B at FULL depth (32 + 28, tied head) — against the oracle, teacher-forced for 16
      decode steps, and config 2's batch of 8 pages: batch = solo, eager = graph, slot scheduler;
  (b') Qwen2.5-VL-7B widths (the reference's default olmOCR-7B-0725 architecture): 8 windowed / full-attention ViT blocks
      + 2 decoder layers on a 1024x1024 scan; the W8A8 prefill against the oracle's w8a8 policy;
  (c) config 5 end to end: a 1700x2200 scan at max_pixels = 12 845 056 (158x122 = 19 276 patches) through the ViT, its
      4988-token prompt through the fp8 engine's prefill and 9 decode steps at contexts ~5000;
  (d) config 1 geometry: a 1056x1422 JPEG through VLLMClient.generate -> LocalServer -> engine (grid 82x60, 1230 image
      tokens), ids equal to a direct Engine.generate;
  (e) config 4 shape: 64 requests from 8 VLLMClient worker threads through the continuous server
      (/root/reference/bulk_processing/workers/inference_worker.py:324-339), every result equal to its solo run.

Tolerance (floating point, stated): engine (bf16 storage, fp32 accumulate) vs the oracle at the same dtype policy —
ViT merged output within 2 % of the reference's range, logits within TOL = 1 % of the logit range at the last prompt
position and 1.5 % at every decode step (measured on MI355X: 0.5-0.6 %); tokens equal at every step whose oracle top-2
margin exceeds 2 x TOL.  Two runs per model: free-running (the sequences must agree until a near-tie lets them part:
prodwidth.compare_generation) and TEACHER-FORCED with the oracle's tokens (Engine.generate(force_tokens=...)), where
every one of the steps compares and a minimum number of DECISIVE steps is asserted — a test that compared nothing
fails.  Measured errors and margins are written to gpurun_out/prodwidth_report.json.
"""
import base64
import io
import json
import os
import threading

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from karanta_ocr_amd import image_processing as IP  # noqa: E402
from karanta_ocr_amd.engine import Engine, PageRequest  # noqa: E402
from karanta_ocr_amd.weights import fp8_dequantized_weights, random_weights  # noqa: E402
from oracle import qwen2vl_oracle as O  # noqa: E402  (checker only)

from tests import prodwidth as PW  # noqa: E402

REPORT = {}
TOL_REL = 0.01        # of the logit range (max |logit| of the reference at the last prompt position)
MAXPIX_A = 1003520
MAXPIX_B = 12845056


def _record(key, **vals):
    REPORT[key] = {k: (float(v) if isinstance(v, (np.floating, float)) else v) for k, v in vals.items()}
    try:
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "prodwidth_report.json"), "w") as f:
            json.dump(REPORT, f, indent=1)
    except OSError:
        pass


@pytest.fixture(scope="module")
def m2b():
    """Qwen2-VL-2B at full widths, 4 ViT blocks + 4 decoder layers, full vocabulary, untied head; engine with 8 slots."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cfg = PW.truncated_config("Qwen2-VL-2B", 4, 4)
    w = random_weights(cfg, 7, as_bits=True)
    eng = Engine(cfg, max_batch=8, s_max=2048, max_patches=3 * 4960, max_prompt_tokens=3 * 1400, decode_splits=16)
    eng.load_weights(w)
    assert eng.wide_mode and eng.narrow_mode and eng.defer_down, "the production decode kernels must be the ones running"
    yield cfg, w, eng
    eng.close()


@pytest.fixture(scope="module")
def m7b():
    """Qwen2-VL-7B at full widths (d 3584, ff 18944, 28/4 heads, vocabulary 152064), 2 ViT blocks + 2 decoder layers."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cfg = PW.truncated_config("Qwen2-VL-7B", 2, 2)
    w = random_weights(cfg, 11, as_bits=True)
    return cfg, w, {}      # + a cache: the bf16 and fp8 runs share the page's ViT output (the tower is bf16 in both)


# ------------------------------------------------------------------------------------------------------------ (a)
def test_2b_width_vit_prefill_and_greedy_tokens_match_oracle(m2b):
    cfg, w, eng = m2b
    ids, pv, grid = PW.page_inputs(cfg, 300, 1024, 1024, MAXPIX_A, 20, 30, 31)
    assert grid == (1, 70, 70) and len(ids) == 20 + 1225 + 2 + 30
    steps = 12
    # ---- ViT: one 4900-token attention segment, 256x256 GEMM tiles (M = 4900)
    got_img = eng.vit_forward(pv, [grid])
    eng.stream.synchronize()
    got_img = got_img.float().cpu().numpy()
    ref_img = O.vit_forward(pv, [grid], w, cfg.vision, policy="bf16")
    scale = float(np.abs(ref_img).max())
    vit_err = float(np.abs(got_img - ref_img).max())
    assert vit_err < 0.02 * scale, f"ViT merged output off by {vit_err} (range {scale})"
    # ---- prefill + decode, eager with logits
    page = PageRequest(ids, pv, [grid])
    res = eng.generate([page], steps, ignore_eos=True, return_logits=True)
    o_tok, o_log = O.generate_greedy(cfg, w, ids[None], None, [grid], steps, policy="bf16", ignore_eos=True, return_logits=True,
                                     image_embeds=ref_img)
    tol = TOL_REL * float(np.abs(o_log[0, 0]).max())
    errs = [float(np.abs(res.logits[0, i] - o_log[0, i]).max()) for i in range(steps)]
    assert errs[0] < tol, f"prefill logits off by {errs[0]} (tol {tol})"
    decisive, walked = PW.compare_generation(res.tokens[0], res.logits[0], o_tok[0], o_log[0], tol, "2B widths")
    # teacher-forced: every step compares, whatever the near-ties did to the free run
    forced = eng.generate([page], steps, ignore_eos=True, return_logits=True, force_tokens=o_tok[:, :steps - 1])
    f_errs = [float(np.abs(forced.logits[0, i] - o_log[0, i]).max()) for i in range(steps)]
    f_decisive = PW.compare_teacher_forced(forced.tokens[0], forced.logits[0], o_tok[0], o_log[0], tol, "2B widths (forced)")
    _record("2b_w_v4_l4", vit_err=vit_err, vit_range=scale, tol=tol, logit_err_per_step=errs[:walked + 1], forced_logit_err=f_errs,
            margins=PW.margins(o_log[0]).tolist(), decisive=decisive, walked=walked, forced_decisive=f_decisive,
            tokens=[int(t) for t in res.tokens[0]], oracle_tokens=[int(t) for t in o_tok[0]])
    assert f_decisive >= 8, f"only {f_decisive} of {steps} teacher-forced steps were decisive"
    assert walked >= 4 and decisive >= 3, f"free run: only {decisive} decisive / {walked} walked steps of {steps}"
    # ---- the FAST-RESIDUAL decode step (per-head o_proj + float atomics, no merge launch): same tolerance, same decisive
    # tokens against the oracle; not bit-identical to the deterministic step by construction (sum order)
    assert not hasattr(eng, "set_fast_residual")          # the product engine holds no experiment mode (csrc/tools/experiment_engine.py)
    if eng.L.experiments:      # an experiment build (-DKR_EXPERIMENTS, loaded through KARANTA_HIP_LIB): the same weights on the ExperimentEngine
        from karanta_ocr_amd.csrc.tools.experiment_engine import ExperimentEngine
        prod, eng = eng, ExperimentEngine(cfg, max_batch=prod.B, s_max=prod.s_max, max_patches=prod.max_patches,
                                          max_prompt_tokens=prod.max_tokens, decode_splits=prod.n_split)
        eng.w = prod.w
        assert eng.set_fast_residual(True)
        try:
            fast = eng.generate([page], steps, ignore_eos=True, return_logits=True, force_tokens=o_tok[:, :steps - 1])
            fast_decisive = PW.compare_teacher_forced(fast.tokens[0], fast.logits[0], o_tok[0], o_log[0], tol, "2B widths (fast residual)")
            dev = float(np.abs(fast.logits[0] - forced.logits[0]).max())
            fast_graph = eng.generate([page], steps, ignore_eos=True)
            m = PW.margins(o_log[0])
            for i in range(steps):           # the replayed graph of the fast step follows the oracle wherever it is decisive
                if m[i] <= 2 * tol:
                    break
                assert int(fast_graph.tokens[0][i]) == int(o_tok[0, i]), f"fast graph step {i}"
            _record("2b_w_v4_l4_fast_residual", decisive=fast_decisive, max_logit_dev_vs_deterministic=dev, tol=tol)
            # (dev is sum-order noise of the float atomics: 0.3-0.5 tol from run to run; the oracle check above is the bound)
            assert fast_decisive == f_decisive and dev < tol
        finally:
            eng.set_fast_residual(False)
            eng = prod
    # ---- the replayed graph gives the eager tokens; a ragged batch of 3 gives page 0 its solo tokens
    graph = eng.generate([page], steps, ignore_eos=True)
    np.testing.assert_array_equal(graph.tokens[0], res.tokens[0])
    others = [PageRequest(*_pg(cfg, 301, 448, 616, 7)), PageRequest(*_pg(cfg, 302, 1024, 700, 3))]
    batch = eng.generate([others[0], page, others[1]], steps, ignore_eos=True)
    np.testing.assert_array_equal(batch.tokens[1], res.tokens[0])


def _pg(cfg, index, h, w, n_pre, max_pixels=MAXPIX_A):
    ids, pv, grid = PW.page_inputs(cfg, index, h, w, max_pixels, n_pre, 5, 1000 + index)
    return ids, pv, [grid]


# ------------------------------------------------------------------------------------------------------------ (a')
@pytest.fixture(scope="module")
def m2b_full():
    """BASELINE.json's bench model as shipped: Qwen2-VL-2B at FULL depth (32 ViT blocks, 28 decoder layers, tied head),
    seeded random weights, an engine with the bench's 8 decode slots."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from karanta_ocr_amd.config import CONFIGS
    cfg = CONFIGS["Qwen2-VL-2B"]
    w = random_weights(cfg, 0, as_bits=True)
    eng = Engine(cfg, max_batch=8, s_max=1536, max_patches=8 * 4960, max_prompt_tokens=8 * 1400, decode_splits=16)
    eng.load_weights(w)
    assert eng.wide_mode and eng.narrow_mode and eng.defer_down
    yield cfg, w, eng
    eng.close()


def test_full_depth_2b_matches_oracle_teacher_forced(m2b_full):
    """VERDICT r2 Missing #2: the bench model at FULL depth against the oracle — bf16 rounding accumulated through 32 ViT
    blocks and 28 decoder layers is what the 4 + 4 truncation cannot show.  One 1024x1024 scan (the bench's page
    geometry: 70x70 patches, 1225 image tokens), call sequence of
    /root/reference/karanta/training/test_trained_model.py:76-99; the oracle at the engine's dtype policy (bf16 storage,
    fp32 accumulate) costs ~40 s of ViT + ~6 s of prefill + ~50 ms per token on the GPU box's host cores.  Two heads on
    the same 28-layer stack:
      * the model AS SHIPPED (tied head).  A random tied head echoes its last input token with a margin of half the logit
        range (logit_i = h . E_i with h ~ E_token), so its 17 steps are all decisive and all equal — what this run pins
        is the logit ERROR at full depth: measured 1.2 - 1.5 % of the range (ViT merged output 1.6 %), against 0.5 - 0.6 %
        at 4 layers;
      * the same stack with an UNTIED random head (the variant every other prodwidth test runs: logits that depend on the
        whole computation, top-2 margins of a few per cent of the range): 33 teacher-forced steps, >= 8 of them decisive
        (measured: logits off by 1.2 - 1.7 % of the range, 9 decisive steps, argmax equal on 31 of 33).
    Stated tolerance at full depth (measured errors with headroom): ViT merged output within 2.5 % of its range, logits
    within TOL = 2 % of the logit range at every step, argmax equal wherever the oracle's top-2 margin exceeds 2 x TOL."""
    cfg, w, eng = m2b_full
    ids, pv, grid = PW.page_inputs(cfg, 300, 1024, 1024, MAXPIX_A, 20, 30, 31)
    assert grid == (1, 70, 70) and cfg.vision.depth == 32 and cfg.text.num_layers == 28 and cfg.text.tie_word_embeddings
    got_img = eng.vit_forward(pv, [grid])
    eng.stream.synchronize()
    got_img = got_img.float().cpu().numpy()
    ref_img = O.vit_forward(pv, [grid], w, cfg.vision, policy="bf16")
    scale = float(np.abs(ref_img).max())
    vit_err = float(np.abs(got_img - ref_img).max())
    assert vit_err < 0.025 * scale, f"ViT merged output off by {vit_err} (range {scale})"
    # the decoder's parameters as resident fp32 arrays (left as bf16 bit patterns, every oracle call re-expands them)
    wl = {k: (O._w(w, k) if not k.startswith("model.visual.") else v) for k, v in w.items()}
    page = PageRequest(ids, pv, [grid])
    TOL_FULL = 0.02

    def run(engine, mcfg, weights, steps, key, min_decisive):
        o_tok, o_log = O.generate_greedy(mcfg, weights, ids[None], None, [grid], steps, policy="bf16", ignore_eos=True,
                                         return_logits=True, image_embeds=ref_img)
        forced = engine.generate([page], steps, ignore_eos=True, return_logits=True, force_tokens=o_tok[:, :steps - 1])
        rng_ = float(np.abs(o_log[0, 0]).max())
        tol = TOL_FULL * rng_
        errs = [float(np.abs(forced.logits[0, i] - o_log[0, i]).max()) for i in range(steps)]
        m = PW.margins(o_log[0])
        decisive = [i for i in range(steps) if m[i] > 2 * tol]
        agree = sum(1 for i in range(steps) if int(forced.tokens[0][i]) == int(o_tok[0, i]))
        _record(key, vit_err=vit_err, vit_range=scale, logit_range=rng_, tol=tol, forced_logit_err=errs, margins=m.tolist(),
                decisive=len(decisive), argmax_equal=agree, steps=steps, tokens=[int(t) for t in forced.tokens[0]],
                oracle_tokens=[int(t) for t in o_tok[0]])
        assert max(errs) < tol, f"{key}: logits off by {max(errs)} (tol {tol}, range {rng_}); per step {errs}"
        for i in decisive:
            assert int(forced.tokens[0][i]) == int(o_tok[0, i]), \
                f"{key} step {i}: engine {int(forced.tokens[0][i])} vs oracle {int(o_tok[0, i])} at margin {m[i]:.3f}"
        assert len(decisive) >= min_decisive, f"{key}: only {len(decisive)} of {steps} steps were decisive (margins {m.tolist()})"
        free = engine.generate([page], steps, ignore_eos=True)     # the replayed graph follows the oracle up to the first near-tie
        for i in range(steps):
            if m[i] <= 2 * tol:
                break
            assert int(free.tokens[0][i]) == int(o_tok[0, i]), f"{key}: graph step {i}"

    run(eng, cfg, wl, 17, "2b_full_depth_v32_l28_tied", 8)
    # the same stack under an untied random head
    ucfg = PW.truncated_config("Qwen2-VL-2B", 32, 28, untie=True)
    head = random_weights(ucfg, 0, as_bits=True, only=["lm_head.weight"])
    assert ucfg.text.num_layers == 28 and ucfg.vision.depth == 32 and not ucfg.text.tie_word_embeddings and len(head) == 1
    ueng = Engine(ucfg, max_batch=1, s_max=1536, max_patches=4960, max_prompt_tokens=1400, decode_splits=16)
    try:
        ueng.load_weights({**w, **head})
        run(ueng, ucfg, {**wl, "lm_head.weight": O._w(head, "lm_head.weight")}, 33, "2b_full_depth_v32_l28_untied", 8)
    finally:
        ueng.close()


def test_full_size_qwen2_vl_2b_batch_of_8_properties(m2b_full):
    """BASELINE.json config 2 exactly — Qwen2-VL-2B, full depth, a batch of EIGHT pages (six 1024x1024 scans -> 70x70
    patches, two of other sizes so the batch is ragged): every page's tokens in the batch equal its solo run (ViT
    segments, the GEMM tile choice that changes with M, prefill, split-KV decode: all batch-independent by construction),
    the eager decode loop equals the replayed graph, and the slot scheduler (other slots, other history) reproduces them
    once more.  The oracle comparison at this depth is the test above."""
    from karanta_ocr_amd.scheduler import SlotRequest, SlotScheduler
    cfg, w, eng = m2b_full
    rng = np.random.default_rng(31)
    pages = []
    for i, (h, wd) in enumerate([(1024, 1024)] * 3 + [(700, 1000)] + [(1024, 1024)] * 3 + [(448, 616)]):
        pv, grid = IP.image_to_patches(IP.synthetic_page(200 + i, h, wd), max_pixels=MAXPIX_A)
        T = grid[1] * grid[2] // 4
        ids = np.concatenate([rng.integers(0, 150000, 20 + 7 * (i % 3)), [cfg.vision_start_token_id], [cfg.image_token_id] * T,
                              [cfg.vision_end_token_id], rng.integers(0, 150000, 30)]).astype(np.int64)
        pages.append(PageRequest(ids, pv, [grid]))
    assert pages[0].grids[0] == (1, 70, 70) and len(pages) == 8
    steps = 24
    together = eng.generate(pages, steps, ignore_eos=True)
    for t in together.tokens:
        assert len(t) == steps and t.min() >= 0 and t.max() < cfg.text.vocab_size
    assert not np.array_equal(together.tokens[0], together.tokens[1]), "different scans, different text"
    for i, pg in enumerate(pages):
        alone = eng.generate([pg], steps, ignore_eos=True)
        np.testing.assert_array_equal(alone.tokens[0], together.tokens[i], err_msg=f"page {i}: batch vs solo")
    eager = eng.generate(pages, steps, ignore_eos=True, use_graph=False)
    for i in range(8):
        np.testing.assert_array_equal(eager.tokens[i], together.tokens[i], err_msg=f"page {i}: eager vs graph")
    sch = SlotScheduler(eng, max_tokens_cap=steps, chunk=5, eos_token_ids=())
    res = sch.run([SlotRequest(pg, steps, tag=i) for i, pg in enumerate(pages[::-1])])
    for r, want in zip(res, together.tokens[::-1]):
        assert r.error is None
        np.testing.assert_array_equal(r.tokens[:steps], want[:len(r.tokens[:steps])])


# ------------------------------------------------------------------------------------------------------------ (b)
@pytest.mark.parametrize("weight_dtype", ["bf16", "fp8"])
def test_7b_width_batch_of_4_matches_oracle(m7b, weight_dtype):
    """BASELINE.json config 3's per-GPU share (4 pages) at the 7B widths; fp8 = config 5's weight format, checked against
    the oracle on the dequantised state dict (the same model, so the same tolerance and token rule).  (The W8A8 prefill
    has its own test below: its tolerance is another one.)"""
    cfg, w, cache = m7b
    ids, pv, grid = PW.page_inputs(cfg, 310, 1024, 1024, MAXPIX_A, 12, 25, 77)
    small = [PageRequest(*_pg(cfg, 311 + k, h, wd, 3 + k)) for k, (h, wd) in enumerate([(336, 448), (560, 420), (224, 224)])]
    pages = [small[0], small[1], PageRequest(ids, pv, [grid]), small[2]]
    steps = 10
    eng = Engine(cfg, max_batch=4, s_max=2048, max_patches=sum(len(p.pixel_values) for p in pages),
                 max_prompt_tokens=sum(len(p.input_ids) for p in pages), decode_splits=16,
                 weight_dtype=weight_dtype)
    eng.load_weights(w)
    try:
        assert eng.wide_mode and eng.narrow_mode and eng.defer_down and not eng.fp8_act
        res = eng.generate(pages, steps, ignore_eos=True, return_logits=True)
        wref = fp8_dequantized_weights(w, cfg) if weight_dtype != "bf16" else w
        if "img" not in cache:
            cache["img"] = O.vit_forward(pv, [grid], w, cfg.vision, policy="bf16")
        o_tok, o_log = O.generate_greedy(cfg, wref, ids[None], None, [grid], steps, policy="bf16", ignore_eos=True, return_logits=True,
                                         image_embeds=cache["img"])
        tol = TOL_REL * float(np.abs(o_log[0, 0]).max())
        errs = [float(np.abs(res.logits[2, i] - o_log[0, i]).max()) for i in range(steps)]
        assert errs[0] < tol, f"prefill logits off by {errs[0]} (tol {tol})"
        decisive, walked = PW.compare_generation(res.tokens[2], res.logits[2], o_tok[0], o_log[0], tol, f"7B widths {weight_dtype}")
        # teacher-forced: page 2 follows the oracle's tokens, the other pages their own (i.e. unchanged)
        ft = np.stack([np.asarray(t[:steps - 1], np.int64) for t in res.tokens])
        ft[2] = o_tok[0, :steps - 1]
        forced = eng.generate(pages, steps, ignore_eos=True, return_logits=True, force_tokens=ft)
        f_errs = [float(np.abs(forced.logits[2, i] - o_log[0, i]).max()) for i in range(steps)]
        f_decisive = PW.compare_teacher_forced(forced.tokens[2], forced.logits[2], o_tok[0], o_log[0], tol, f"7B widths {weight_dtype} (forced)")
        for k in (0, 1, 3):
            np.testing.assert_array_equal(forced.tokens[k], res.tokens[k])       # self-forced pages: nothing changes
        _record(f"7b_w_v2_l2_{weight_dtype}", tol=tol, logit_err_per_step=errs[:walked + 1], forced_logit_err=f_errs,
                margins=PW.margins(o_log[0]).tolist(), decisive=decisive, walked=walked, forced_decisive=f_decisive)
        assert f_decisive >= 4, f"only {f_decisive} of {steps} teacher-forced steps were decisive"
        graph = eng.generate(pages, steps, ignore_eos=True)
        for a, b in zip(graph.tokens, res.tokens):
            np.testing.assert_array_equal(a, b)
    finally:
        eng.close()


def test_7b_width_w8a8_prefill_against_the_w8a8_oracle(m7b):
    """The W8A8 prefill (Engine(weight_dtype="fp8", fp8_activations=True): per-token e4m3 activations through
    v_mfma_f32_16x16x32_fp8_fp8, kr_quantize_rows_fp8 + kr_gemm_fp8a; what vLLM runs for the reference's OLMO_7B_0725_FP8,
    /root/reference/karanta/constants.py:23) at the 7B widths, one 1024x1024 page, against the oracle's "w8a8" policy.
    The kernels themselves are pinned exactly (codes bit-identical to the host quantiser, integer GEMMs exact:
    tests/test_gpu_kernels.py).  End to end the comparison cannot be as tight as the bf16-activation one: an e4m3 code has 3
    mantissa bits, so a 1-ulp bf16 difference between the engine's and the oracle's hidden state flips ~7 % of the codes by
    a whole 6 % step — noise of the size of the quantisation error itself, uncorrelated between the two runs.  Stated
    tolerance: the engine is within TWICE the activation-quantisation effect (oracle w8a8 vs oracle with bf16 activations
    on the same fp8 weights, measured in the same test) and within 10 % of the logit range, at the prefill and at every
    teacher-forced decode step; the measured figures go to the report and DESIGN.md section 5f.  This is why bf16 activations
    stay the engine's default for fp8 checkpoints."""
    cfg, w, cache = m7b
    ids, pv, grid = PW.page_inputs(cfg, 310, 1024, 1024, MAXPIX_A, 12, 25, 77)
    page = PageRequest(ids, pv, [grid])
    steps = 8
    eng = Engine(cfg, max_batch=1, s_max=2048, max_patches=len(pv), max_prompt_tokens=len(ids), decode_splits=16, weight_dtype="fp8",
                 fp8_activations=True)
    eng.load_weights(w)
    try:
        assert eng.fp8_act and eng.wide_mode and eng.narrow_mode
        wref = fp8_dequantized_weights(w, cfg)
        if "img" not in cache:
            cache["img"] = O.vit_forward(pv, [grid], w, cfg.vision, policy="bf16")
        oq_tok, oq_log = O.generate_greedy(cfg, wref, ids[None], None, [grid], steps, policy="w8a8", ignore_eos=True, return_logits=True,
                                           image_embeds=cache["img"])
        ob_tok, ob_log = O.generate_greedy(cfg, wref, ids[None], None, [grid], 1, policy="bf16", ignore_eos=True, return_logits=True,
                                           image_embeds=cache["img"])
        forced = eng.generate([page], steps, ignore_eos=True, return_logits=True, force_tokens=oq_tok[:, :steps - 1])
        rng_ = float(np.abs(oq_log[0, 0]).max())
        q_eff = float(np.abs(oq_log[0, 0] - ob_log[0, 0]).max())             # what quantising the activations does to the logits
        errs = [float(np.abs(forced.logits[0, i] - oq_log[0, i]).max()) for i in range(steps)]
        e_bf = float(np.abs(forced.logits[0, 0] - ob_log[0, 0]).max())       # engine (W8A8) vs the bf16-activation oracle
        m = PW.margins(oq_log[0])
        agree = sum(int(forced.tokens[0][i]) == int(oq_tok[0, i]) for i in range(steps))
        _record("7b_w_v2_l2_w8a8", logit_range=rng_, act_quant_effect=q_eff, engine_vs_w8a8_oracle=errs, engine_vs_bf16act_oracle=e_bf,
                margins=m.tolist(), argmax_equal=agree, steps=steps)
        assert max(errs) < 2.0 * q_eff and max(errs) < 0.10 * rng_, f"W8A8 logits off by {max(errs)} (quantisation effect {q_eff}, range {rng_})"
        for i in range(steps):
            if m[i] > 2 * max(errs):
                assert int(forced.tokens[0][i]) == int(oq_tok[0, i]), f"step {i}"
        graph = eng.generate([page], steps, ignore_eos=True)     # the W8A8 prefill feeds the same decode graph
        free = eng.generate([page], steps, ignore_eos=True, use_graph=False)
        np.testing.assert_array_equal(graph.tokens[0], free.tokens[0])
    finally:
        eng.close()


def test_qwen2_5_vl_7b_width_matches_oracle():
    """The architecture of the reference's DEFAULT checkpoint (olmOCR-7B-0725 = Qwen2.5-VL-7B,
    /root/reference/karanta/constants.py:22-24) at production widths: the windowed vision tower (112-pixel windows of 64
    patches, RMSNorm, biased SwiGLU MLP of width 3420 padded to 3456) with EIGHT blocks, so that blocks 0-6 run the
    window work list and block 7 the full-attention one over the 4900-token page, the gather into window order and the
    scatter back at real size, and the 7B-width decoder (2 layers), one 1024x1024 scan: ViT merged output, prefill logits
    and 10 teacher-forced decode steps against the oracle (vit_forward_qwen2_5, pinned by the HF Qwen2.5-VL goldens).
    Tolerances as in (b)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cfg = PW.truncated_config("Qwen2.5-VL-7B", 8, 2)
    assert cfg.vision.variant == "qwen2_5" and 7 in cfg.vision.fullatt_block_indexes
    w = random_weights(cfg, 13, as_bits=True)
    ids, pv, grid = PW.page_inputs(cfg, 340, 1024, 1024, MAXPIX_A, 14, 27, 91)
    assert grid == (1, 70, 70)
    steps = 11
    eng = Engine(cfg, max_batch=1, s_max=2048, max_patches=len(pv), max_prompt_tokens=len(ids), decode_splits=16)
    eng.load_weights(w)
    try:
        assert eng.wide_mode and eng.narrow_mode and eng.defer_down
        got_img = eng.vit_forward(pv, [grid])
        eng.stream.synchronize()
        got_img = got_img.float().cpu().numpy()
        ref_img = O.vit_forward(pv, [grid], w, cfg.vision, policy="bf16")
        scale, vit_err = float(np.abs(ref_img).max()), float(np.abs(got_img - ref_img).max())
        assert vit_err < 0.02 * scale, f"Qwen2.5-VL ViT merged output off by {vit_err} (range {scale})"
        o_tok, o_log = O.generate_greedy(cfg, w, ids[None], None, [grid], steps, policy="bf16", ignore_eos=True, return_logits=True,
                                         image_embeds=ref_img)
        page = PageRequest(ids, pv, [grid])
        forced = eng.generate([page], steps, ignore_eos=True, return_logits=True, force_tokens=o_tok[:, :steps - 1])
        tol = TOL_REL * float(np.abs(o_log[0, 0]).max())
        f_errs = [float(np.abs(forced.logits[0, i] - o_log[0, i]).max()) for i in range(steps)]
        assert f_errs[0] < tol, f"prefill logits off by {f_errs[0]} (tol {tol})"
        f_decisive = PW.compare_teacher_forced(forced.tokens[0], forced.logits[0], o_tok[0], o_log[0], tol, "Qwen2.5-VL-7B widths (forced)")
        _record("qwen2_5_vl_7b_w_v8_l2", vit_err=vit_err, vit_range=scale, tol=tol, forced_logit_err=f_errs,
                margins=PW.margins(o_log[0]).tolist(), forced_decisive=f_decisive)
        assert f_decisive >= 3, f"only {f_decisive} of {steps} teacher-forced steps were decisive"
        graph = eng.generate([page], steps, ignore_eos=True)
        m = PW.margins(o_log[0])
        for i in range(steps):
            if m[i] <= 2 * tol:
                break
            assert int(graph.tokens[0][i]) == int(o_tok[0, i]), f"graph step {i}"
    finally:
        eng.close()


def test_qwen2_5_vl_3b_width_matches_oracle():
    """The architecture the reference's OWN fine-tunes start from (Qwen2.5-VL-3B,
    /root/reference/configs/training/ocr/karanta_set_qwen_2_5_3B_vl.yaml:2) at production widths: the windowed vision tower
    (eight blocks: window blocks 0-6, full attention in block 7; merger output 2048) and the 3B-width decoder — hidden 2048
    (the K = 2048 instantiations of the decode kernels), 16 query / 2 KV heads, intermediate 11008 (down_proj in two K halves of
    86 chunks), untied lm_head — two layers, one 1024x1024 scan: ViT merged output, prefill logits and 10 teacher-forced decode
    steps against the oracle; the hipGraph path reproduces the oracle's greedy tokens up to the first undecisive step."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cfg = PW.truncated_config("Qwen2.5-VL-3B", 8, 2)
    t = cfg.text
    assert (t.hidden_size, t.num_heads, t.num_kv_heads, t.intermediate_size) == (2048, 16, 2, 11008) and cfg.vision.variant == "qwen2_5"
    w = random_weights(cfg, 17, as_bits=True)
    ids, pv, grid = PW.page_inputs(cfg, 341, 1024, 1024, MAXPIX_A, 14, 27, 91)
    assert grid == (1, 70, 70)
    steps = 11
    eng = Engine(cfg, max_batch=1, s_max=2048, max_patches=len(pv), max_prompt_tokens=len(ids), decode_splits=16)
    eng.load_weights(w)
    try:
        assert eng.wide_mode and eng.narrow_mode and eng.defer_down
        got_img = eng.vit_forward(pv, [grid])
        eng.stream.synchronize()
        got_img = got_img.float().cpu().numpy()
        ref_img = O.vit_forward(pv, [grid], w, cfg.vision, policy="bf16")
        scale, vit_err = float(np.abs(ref_img).max()), float(np.abs(got_img - ref_img).max())
        assert vit_err < 0.02 * scale, f"Qwen2.5-VL-3B ViT merged output off by {vit_err} (range {scale})"
        o_tok, o_log = O.generate_greedy(cfg, w, ids[None], None, [grid], steps, policy="bf16", ignore_eos=True, return_logits=True,
                                         image_embeds=ref_img)
        page = PageRequest(ids, pv, [grid])
        forced = eng.generate([page], steps, ignore_eos=True, return_logits=True, force_tokens=o_tok[:, :steps - 1])
        tol = TOL_REL * float(np.abs(o_log[0, 0]).max())
        f_errs = [float(np.abs(forced.logits[0, i] - o_log[0, i]).max()) for i in range(steps)]
        assert f_errs[0] < tol, f"prefill logits off by {f_errs[0]} (tol {tol})"
        f_decisive = PW.compare_teacher_forced(forced.tokens[0], forced.logits[0], o_tok[0], o_log[0], tol, "Qwen2.5-VL-3B widths (forced)")
        _record("qwen2_5_vl_3b_w_v8_l2", vit_err=vit_err, vit_range=scale, tol=tol, forced_logit_err=f_errs,
                margins=PW.margins(o_log[0]).tolist(), forced_decisive=f_decisive)
        assert f_decisive >= 3, f"only {f_decisive} of {steps} teacher-forced steps were decisive"
        graph = eng.generate([page], steps, ignore_eos=True)
        m = PW.margins(o_log[0])
        for i in range(steps):
            if m[i] <= 2 * tol:
                break
            assert int(graph.tokens[0][i]) == int(o_tok[0, i]), f"graph step {i}"
    finally:
        eng.close()


# ------------------------------------------------------------------------------------------------------------ (c)
def test_config5_1700x2200_fp8_end_to_end_matches_oracle(m7b):
    """BASELINE.json config 5 end to end at the 7B widths (2 + 2 depth): one 1700x2200 newspaper scan at the hub
    preprocessor's max_pixels (12 845 056): 2212x1708 -> 158x122 = 19 276 patches, ONE attention segment of 19 276 tokens
    (302 KV tiles), 4819 image tokens; the chat template around it makes a 4988-token prompt — prefill with 78 causal KV
    tiles per query block through the FP8-weight engine (kr_gemm_fp8 on M = 4988), then 23 teacher-forced decode steps at
    contexts 4988 .. 5010 (16-split decode attention over 5k cached tokens, the fp8 wide / narrow decode kernels), against
    the oracle on the dequantised state dict.  Tolerances as in (b) (measured: logits off by 0.5 % of their range at every
    step; the random model's top-2 margins at this prompt are small — 3 of the first 10 steps decisive — hence 24 steps)."""
    cfg, w, _ = m7b
    pv, grid = IP.image_to_patches(IP.synthetic_page(320, 2200, 1700), max_pixels=MAXPIX_B)
    assert grid == (1, 158, 122) and len(pv) == 19276
    rng = np.random.default_rng(55)
    ids = np.concatenate([rng.integers(0, 150000, 110), [cfg.vision_start_token_id], [cfg.image_token_id] * 4819,
                          [cfg.vision_end_token_id], rng.integers(0, 150000, 57)]).astype(np.int64)
    assert len(ids) == 4988
    steps = 24
    eng = Engine(cfg, max_batch=1, s_max=5184, max_patches=19276, max_prompt_tokens=5056, decode_splits=16, weight_dtype="fp8")
    eng.load_weights(w)
    try:
        assert eng.wide_mode and eng.narrow_mode and eng.defer_down and eng.fp8_prefill_gemm
        got = eng.vit_forward(pv, [grid])
        eng.stream.synchronize()
        got = got.float().cpu().numpy()
        ref = O.vit_forward(pv, [grid], w, cfg.vision, policy="bf16")
        assert got.shape == ref.shape == (4819, 3584)
        scale, err = float(np.abs(ref).max()), float(np.abs(got - ref).max())
        _record("config5_vit_19276", err=err, range=scale, rel=err / scale)
        assert err < 0.02 * scale, f"ViT merged output off by {err} (range {scale})"
        wref = fp8_dequantized_weights(w, cfg)
        o_tok, o_log = O.generate_greedy(cfg, wref, ids[None], None, [grid], steps, policy="bf16", ignore_eos=True,
                                         return_logits=True, image_embeds=ref)
        page = PageRequest(ids, pv, [grid])
        forced = eng.generate([page], steps, ignore_eos=True, return_logits=True, force_tokens=o_tok[:, :steps - 1])
        tol = TOL_REL * float(np.abs(o_log[0, 0]).max())
        f_errs = [float(np.abs(forced.logits[0, i] - o_log[0, i]).max()) for i in range(steps)]
        assert f_errs[0] < tol, f"prefill logits (P = 4988) off by {f_errs[0]} (tol {tol})"
        f_decisive = PW.compare_teacher_forced(forced.tokens[0], forced.logits[0], o_tok[0], o_log[0], tol, "config 5 fp8 (forced)")
        _record("config5_fp8_p4988", tol=tol, forced_logit_err=f_errs, margins=PW.margins(o_log[0]).tolist(), forced_decisive=f_decisive)
        assert f_decisive >= 4, f"only {f_decisive} of {steps} teacher-forced steps were decisive"
        graph = eng.generate([page], steps, ignore_eos=True)
        m = PW.margins(o_log[0])
        for i in range(steps):
            if m[i] <= 2 * tol:
                break
            assert int(graph.tokens[0][i]) == int(o_tok[0, i]), f"graph step {i}"
    finally:
        eng.close()


# ------------------------------------------------------------------------------------------------------------ (d), (e)
class IdTokenizer:
    """ByteTokenizer whose decode spells every id out (`<123>`): a completion's text IS its token ids, so the
    VLLMClient-shaped API can be compared with a direct Engine.generate token for token."""

    def __init__(self, cfg):
        from karanta_ocr_amd.serving import ByteTokenizer
        self._b = ByteTokenizer(cfg)
        self.im_start, self.im_end, self.newline = self._b.im_start, self._b.im_end, self._b.newline

    def encode(self, text):
        return self._b.encode(text)

    def decode(self, ids):
        return "".join(f"<{int(i)}>" for i in ids)


def _jpeg_data_url(img_u8, quality=90):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(img_u8).save(buf, format="JPEG", quality=quality)
    return "data:image/jpeg;base64," + base64.b64encode(buf.getvalue()).decode()


def _vision_request(url, text, max_tokens, **kw):
    # the message shape of create_vision_message (/root/reference/karanta/data/utils.py:283-297): text first, image second
    return dict({"model": "karantaocr", "max_tokens": max_tokens, "temperature": 0.0,
                 "messages": [{"role": "user", "content": [{"type": "text", "text": text},
                                                           {"type": "image_url", "image_url": {"url": url}}]}]}, **kw)


def test_config1_sample_jpg_geometry_through_the_vllm_client(m2b):
    """BASELINE.json config 1's page — tests/sample.jpg is a 1056x1422 JPEG: smart_resize -> 1148x840, grid 82x60, 1230
    image tokens — through VLLMClient.generate -> LocalServer -> ChatFrontend (JPEG decode, GPU front end)
    -> engine; the completion's ids equal a direct Engine.generate on the same prompt and pixels."""
    from karanta_ocr_amd import serving as S
    from karanta_ocr_amd.clients import VLLMClient
    cfg, w, eng = m2b
    url = _jpeg_data_url(IP.synthetic_page(330, 1422, 1056))
    front = S.ChatFrontend(cfg, IdTokenizer(cfg), max_pixels=MAXPIX_A, device_images=True)
    srv = S.LocalServer(eng, front, log=lambda *_: None)
    port = 18461
    S.register_local_server(port, srv)
    try:
        req = _vision_request(url, "Below is the image of one page of a document.", 12)
        parsed = front.parse(req)
        assert parsed.grids == [(1, 82, 60)] and int((parsed.input_ids == cfg.image_token_id).sum()) == 1230
        r = VLLMClient(port=port).generate(req["messages"], model="karantaocr", max_tokens=12, temperature=0.0)
        assert set(r) == {"text", "finish_reason", "model", "usage", "metadata"}
        assert r["usage"]["prompt_tokens"] == len(parsed.input_ids) and r["usage"]["completion_tokens"] >= 1
    finally:
        S.unregister_local_server(port)
        srv.close()
    # the same page decoded and preprocessed on the host (PIL), straight into the engine
    pv, grid = IP.image_to_patches(IP.decode_data_url(url), max_pixels=MAXPIX_A)
    assert grid == (1, 82, 60)
    direct = eng.generate([PageRequest(parsed.input_ids, pv, [grid])], 12)
    eos = set(cfg.eos_token_ids)
    toks = [int(t) for t in direct.tokens[0] if int(t) not in eos]
    assert r["text"] == "".join(f"<{t}>" for t in toks)
    assert r["finish_reason"] == direct.finish_reasons[0]


def test_config4_corpus_through_eight_bulk_clients_equals_solo_runs(m2b):
    """BASELINE.json config 4's shape on one GPU: 64 page requests pulled from a queue by 8 worker threads, each with its
    own VLLMClient (one client per Celery worker process in the reference, inference_worker.py:324-339), through the
    continuous-batching server (8 decode slots); pages of three sizes, ragged max_tokens.  Every completion equals the
    solo static run of its page (greedy decoding is a pure function of the page: a shorter limit is a prefix)."""
    import queue
    from karanta_ocr_amd import serving as S
    from karanta_ocr_amd.clients import VLLMClient
    cfg, w, eng = m2b
    front = S.ChatFrontend(cfg, IdTokenizer(cfg), max_pixels=MAXPIX_A, device_images=True)
    geoms = [(1024, 1024), (448, 616), (700, 504), (336, 336)]
    n_pages, n_req, longest = 16, 64, 14
    urls = [IP.encode_png_data_url(IP.synthetic_page(400 + i, *geoms[i % len(geoms)])) for i in range(n_pages)]
    limits = [3 + (5 * k) % (longest - 2) for k in range(n_req)]
    # solo reference: each distinct page once, at the longest limit
    solo = {}
    for i, url in enumerate(urls):
        p = front.parse(_vision_request(url, f"page {i}", longest))
        out = eng.generate([PageRequest(p.input_ids, None, p.grids, images=p.images)], longest)
        solo[i] = (out.tokens[0], out.finish_reasons[0])
    srv = S.LocalServer(eng, front, log=lambda *_: None, continuous=True, max_tokens_cap=longest, chunk=4)
    port = 18462
    S.register_local_server(port, srv)
    work: "queue.Queue" = queue.Queue()
    for k in range(n_req):
        work.put(k)
    results, errors = {}, []

    def worker():
        client = VLLMClient(port=port, max_retries=0)
        while True:
            try:
                k = work.get_nowait()
            except queue.Empty:
                return
            try:
                i = k % n_pages
                req = _vision_request(urls[i], f"page {i}", limits[k])
                results[k] = client.generate(req["messages"], model="karantaocr", max_tokens=limits[k], temperature=0.0)
            except Exception as e:  # noqa: BLE001
                errors.append((k, repr(e)))

    try:
        ts = [threading.Thread(target=worker) for _ in range(8)]
        [t.start() for t in ts]
        [t.join(timeout=600) for t in ts]
    finally:
        S.unregister_local_server(port)
        srv.close()
    assert not errors, errors[:3]
    assert len(results) == n_req and srv.pages_done == n_req
    eos = set(cfg.eos_token_ids)
    for k in range(n_req):
        toks, reason = solo[k % n_pages]
        lim = limits[k]
        want = list(toks[:lim])
        stopped = reason == "stop" and len(toks) <= lim
        if stopped:
            want = [t for t in want if int(t) not in eos]
        assert results[k]["text"] == "".join(f"<{int(t)}>" for t in want), f"request {k} (page {k % n_pages}, limit {lim})"
        assert results[k]["finish_reason"] == ("stop" if stopped else "length")
        assert results[k]["usage"]["completion_tokens"] == len(want)
    st, m = srv.metrics()
    assert st == 200 and m["pages_done"] == n_req and m["latency_s"]["n"] == n_req
