"""Pin oracle/qwen2vl_oracle.py against the Hugging Face golden fixtures
(tests/golden/make_golden.py; transformers 5.15.0, fp32, eager attention)."""
import numpy as np
import pytest

from oracle import qwen2vl_oracle as O

MODELS = ["tiny", "tiny-gqa"]


def test_smart_resize_table(golden):
    for h, w, mp, eh, ew in golden["smart_resize_table"]:
        assert O.smart_resize(int(h), int(w), 28, 3136, int(mp)) == (int(eh), int(ew))
    # values quoted in SURVEY.md §8 / BASELINE.md §3
    assert O.smart_resize(1024, 1024, 28, 3136, 1003520) == (980, 980)
    assert O.smart_resize(1024, 1024, 28, 3136, 12845056) == (1036, 1036)
    assert O.smart_resize(2200, 1700, 28, 3136, 12845056) == (2212, 1708)


def test_smart_resize_aspect_error():
    with pytest.raises(ValueError):
        O.smart_resize(10, 5000)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_preprocess_matches_hf(golden, tag):
    pv, grid = O.preprocess_image(golden[f"pre_img_{tag}"])
    assert tuple(golden[f"pre_grid_{tag}"][0]) == grid
    np.testing.assert_allclose(pv, golden[f"pre_pv_{tag}"], rtol=0, atol=2e-6)


def test_preprocess_max_pixels_clamp(golden):
    pv, grid = O.preprocess_image(golden["pre_img_c"], min_pixels=56 * 56, max_pixels=28 * 28 * 64)
    assert tuple(golden["pre_grid_c"][0]) == grid
    np.testing.assert_allclose(pv[:8], golden["pre_pv_c_head"], rtol=0, atol=2e-6)
    np.testing.assert_allclose([pv.sum(), np.abs(pv).sum()], golden["pre_pv_c_sum"], rtol=1e-5)


@pytest.mark.parametrize("name", MODELS)
def test_vision_positions_and_rotary(golden, tiny_models, name):
    cfg, _, P = tiny_models[name]
    pos = O.vision_position_ids(golden[P + "vit_grid"], cfg.vision.spatial_merge_size)
    np.testing.assert_array_equal(pos, golden[P + "vit_pos_ids"])
    cos, sin = O.vision_rotary_cos_sin(pos, cfg.vision.head_dim)
    np.testing.assert_allclose(cos, golden[P + "vit_cos"], atol=1e-6)
    np.testing.assert_allclose(sin, golden[P + "vit_sin"], atol=1e-6)


MODELS_ALL = MODELS + ["tiny-2.5"]   # the Qwen2.5-VL fixtures hold the vision tower and the end-to-end arrays


def test_qwen2_5_window_index(golden, tiny_models):
    """Window order of the merged units and cumulative window lengths (HF get_vision_window_index): one grid with
    ragged right / bottom windows, one that divides exactly (the pad-by-a-full-window path); the product's host
    restatement (positions.vision_window_order) gives the same."""
    from karanta_ocr_amd import positions as POS
    cfg, w, P = tiny_models["tiny-2.5"]
    grid = [tuple(int(x) for x in g) for g in golden[P + "vit_grid"]]
    widx, cu = O.vision_window_index(grid, 2, cfg.vision.window_size, cfg.vision.patch_size)
    np.testing.assert_array_equal(widx, golden[P + "vit_window_index"])
    np.testing.assert_array_equal(cu, golden[P + "vit_cu_window_seqlens"])
    order, lens = POS.vision_window_order(grid, 2, cfg.vision.window_size, cfg.vision.patch_size)
    np.testing.assert_array_equal(order, widx)
    np.testing.assert_array_equal(np.cumsum([0] + lens), cu)
    for g in ([(1, 70, 70)], [(1, 2, 2)], [(1, 14, 6), (1, 4, 30)]):          # production-size and degenerate grids
        a, la = POS.vision_window_order(g, 2, 112, 14)
        b, cb = O.vision_window_index(g, 2, 112, 14)
        np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(np.cumsum([0] + la), cb)
        assert sorted(a.tolist()) == list(range(len(a)))


@pytest.mark.parametrize("name", MODELS_ALL)
def test_vit_forward(golden, tiny_models, name):
    cfg, w, P = tiny_models[name]
    merged, inter = O.vit_forward(golden[P + "vit_pixel_values"], golden[P + "vit_grid"], w, cfg.vision,
                                  return_intermediates=True)
    np.testing.assert_allclose(inter["patch_embed"], golden[P + "vit_patch_embed"], atol=2e-5, rtol=1e-5)
    for i in range(cfg.vision.depth):
        np.testing.assert_allclose(inter[f"block{i}"], golden[P + f"vit_block{i}"], atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(merged, golden[P + "vit_merged"], atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("name", MODELS)
@pytest.mark.parametrize("tag", ["1img", "2img"])
def test_get_rope_index(golden, tiny_models, name, tag):
    cfg, _, P = tiny_models[name]
    pos, delta = O.get_rope_index(golden[P + f"rope_{tag}_ids"], golden[P + f"rope_{tag}_grid"],
                                  cfg.image_token_id, cfg.vision.spatial_merge_size)
    np.testing.assert_array_equal(pos, golden[P + f"rope_{tag}_pos"])
    np.testing.assert_array_equal(delta, golden[P + f"rope_{tag}_delta"])


def test_get_rope_index_survey_values():
    """1024² grid A → first image token (4,4,4), last (4,38,38), next text 39, δ=-1190."""
    ids = np.asarray([[1, 2, 3, 4] + [9] * 1225 + [5, 6]])
    pos, delta = O.get_rope_index(ids, [(1, 70, 70)], 9, 2)
    assert tuple(pos[:, 0, 4]) == (4, 4, 4)
    assert tuple(pos[:, 0, 4 + 1224]) == (4, 38, 38)
    assert tuple(pos[:, 0, 4 + 1225]) == (39, 39, 39)
    assert int(delta[0]) == -1190


@pytest.mark.parametrize("name", MODELS)
def test_rmsnorm(golden, tiny_models, name):
    cfg, w, P = tiny_models[name]
    y = O.rms_norm(golden[P + "rms_x"], w["model.language_model.norm.weight"], cfg.text.rms_norm_eps)
    np.testing.assert_allclose(y, golden[P + "rms_y"], atol=1e-6, rtol=1e-6)


@pytest.mark.parametrize("name", MODELS)
def test_mrope(golden, tiny_models, name):
    cfg, _, P = tiny_models[name]
    cos, sin = O.mrope_cos_sin(golden[P + "mrope_pos"], cfg.text.head_dim, cfg.text.rope_theta, cfg.text.mrope_section)
    qe = O.apply_mrope(golden[P + "mrope_q"], cos, sin)
    ke = O.apply_mrope(golden[P + "mrope_k"], cos, sin)
    # positions up to 3000: one fp32 ulp of the angle is 2.4e-4, and |q| reaches ~4
    np.testing.assert_allclose(qe, golden[P + "mrope_qe"], atol=2e-3)
    np.testing.assert_allclose(ke, golden[P + "mrope_ke"], atol=2e-3)


@pytest.mark.parametrize("name", MODELS)
def test_embed_scatter_and_first_layer(golden, tiny_models, name):
    cfg, w, P = tiny_models[name]
    ids, grid = golden[P + "e2e_input_ids"], golden[P + "e2e_grid"]
    img = O.vit_forward(golden[P + "e2e_pixel_values"], grid, w, cfg.vision)
    emb = O.embed_and_scatter(ids, img, w, cfg)
    np.testing.assert_allclose(emb[0], golden[P + "e2e_hidden_layer0_in"], atol=1e-4, rtol=1e-4)
    pos, _ = O.get_rope_index(ids, grid, cfg.image_token_id, 2)
    cos, sin = O.mrope_cos_sin(pos, cfg.text.head_dim, cfg.text.rope_theta, cfg.text.mrope_section)
    cache = O.KVCache.empty(cfg.text.num_layers)
    x1 = O.decoder_layer(emb, 0, w, cfg.text, cos, sin, cache, O._Policy("fp32"))
    np.testing.assert_allclose(x1[0], golden[P + "e2e_hidden_layer1_in"], atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize("name", MODELS_ALL)
def test_prompt_logits_all_positions(golden, tiny_models, name):
    cfg, w, P = tiny_models[name]
    ids, grid = golden[P + "e2e_input_ids"], golden[P + "e2e_grid"]
    img = O.vit_forward(golden[P + "e2e_pixel_values"], grid, w, cfg.vision)
    emb = O.embed_and_scatter(ids, img, w, cfg)
    pos, _ = O.get_rope_index(ids, grid, cfg.image_token_id, 2)
    logits = O.decoder_forward(emb, pos, w, cfg.text, O.KVCache.empty(cfg.text.num_layers), last_only=False)
    np.testing.assert_allclose(logits[0], golden[P + "e2e_prompt_logits"], atol=5e-4, rtol=1e-4)


@pytest.mark.parametrize("name", MODELS_ALL)
def test_greedy_generate_matches_hf(golden, tiny_models, name):
    """Token ids of HF ``generate(do_sample=False)`` and its per-step scores, with KV cache."""
    cfg, w, P = tiny_models[name]
    n = golden[P + "e2e_gen_ids"].shape[1]
    ids, logits = O.generate_greedy(cfg, w, golden[P + "e2e_input_ids"], golden[P + "e2e_pixel_values"],
                                    golden[P + "e2e_grid"], n, return_logits=True)
    np.testing.assert_array_equal(ids, golden[P + "e2e_gen_ids"])
    np.testing.assert_allclose(logits[0], golden[P + "e2e_gen_scores"], atol=1e-3, rtol=1e-4)


@pytest.mark.parametrize("name", MODELS)
def test_bf16_policy_tolerance(golden, tiny_models, name):
    """The bf16-policy oracle (engine rounding points) against HF's own bf16 CPU run and the
    fp32 run.  Stated tolerance for bf16-vs-fp32 last-position logits on these toy models:
    2 % of the logit range (max |logit|); argmax must agree wherever the fp32 top-2 margin
    exceeds twice that."""
    cfg, w, P = tiny_models[name]
    ids, grid = golden[P + "e2e_input_ids"], golden[P + "e2e_grid"]
    img = O.vit_forward(golden[P + "e2e_pixel_values"], grid, w, cfg.vision, policy="bf16")
    emb = O.embed_and_scatter(ids, img, w, cfg)
    pos, _ = O.get_rope_index(ids, grid, cfg.image_token_id, 2)
    lb = O.decoder_forward(emb, pos, w, cfg.text, O.KVCache.empty(cfg.text.num_layers), policy="bf16")[0]
    l32 = golden[P + "e2e_prompt_logits"][-1]
    lhf = golden[P + "e2e_prompt_logits_hf_bf16_last"]
    tol = 0.02 * np.abs(l32).max()
    assert np.abs(lb - l32).max() < tol
    assert np.abs(lhf - l32).max() < tol
    top2 = np.sort(l32)[-2:]
    if top2[1] - top2[0] > 2 * tol:
        assert lb.argmax() == l32.argmax() == lhf.argmax()


def test_gumbel_uniform_is_strictly_inside_the_unit_interval():
    """The sampler's uniform must never be 0 or 1 (noise -ln(-ln u) would be -inf / +inf): every 23-bit code + 0.5 is
    exact in fp32, including the all-ones hash that the former 24-bit construction rounded to u == 1."""
    from oracle import qwen2vl_oracle as O
    h = np.asarray([0, 1, 0x1FF, 0x200, 0x7FFFFFFF, 0xFFFFFE00, 0xFFFFFFFF], np.uint64)
    u = O.gumbel_u(h)
    assert u.dtype == np.float32 and (u > 0).all() and (u < 1).all()
    assert u[0] == np.float32(2.0 ** -24) and u[-1] == np.float32(1.0 - 2.0 ** -24)
    g = -np.log(-np.log(u, dtype=np.float32), dtype=np.float32)
    assert np.isfinite(g).all()
    # exactness: (code + 0.5) * 2^-23 reproduces in float64
    assert np.array_equal(u.astype(np.float64), ((h >> np.uint64(9)).astype(np.float64) + 0.5) * 2.0 ** -23)
