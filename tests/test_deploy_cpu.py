"""The host half of the deployment path on a synthetic HUB-LAYOUT checkpoint directory (tools/synthetic_checkpoint.py):
what `vllm serve <model dir>` (/root/reference/karanta/pipeline.py:707-742) reads before any kernel runs — config.json,
model.safetensors with hub tensor names (bf16 and compressed-tensors fp8), tokenizer.json through the `tokenizers` library,
the checkpoint's own chat template, preprocessor_config.json — and the request the reference sends
(create_vision_message, /root/reference/karanta/data/utils.py:283-297).  The GPU half: tests/test_gpu_deploy.py."""
import dataclasses
import json
import os

import numpy as np
import pytest

from karanta_ocr_amd import cli
from karanta_ocr_amd import image_processing as IP
from karanta_ocr_amd import serving as S
from karanta_ocr_amd.config import CONFIGS
from karanta_ocr_amd.tools import synthetic_checkpoint as SC
from karanta_ocr_amd.weights import as_f32, load_checkpoint, load_config, random_weights


def reference_request(page_u8, text="Below is the image of one page of a document. Return the plain text.", max_tokens=12, **kw):
    """The wire request of karanta.pipeline.build_page_query (:115-171): text part first, PNG data-URL second."""
    return {"model": "karantaocr", "max_tokens": max_tokens, "temperature": 0.0,
            "messages": [{"role": "user", "content": [{"type": "text", "text": text},
                                                      {"type": "image_url", "image_url": {"url": IP.encode_png_data_url(page_u8)}}]}], **kw}


@pytest.mark.parametrize("name,layout,fp8", [("tiny", "v4", False), ("tiny-2.5", "v5", False), ("tiny-gqa", "v4", False),
                                              ("tiny-w512", "v4", True)])
def test_synthetic_checkpoint_round_trips_through_the_loader(tmp_path, name, layout, fp8):
    cfg = CONFIGS[name]
    d = str(tmp_path / "ckpt")
    meaning = SC.write_checkpoint(d, cfg, 11, layout, fp8)
    assert {"config.json", "model.safetensors", "tokenizer.json", "tokenizer_config.json", "chat_template.jinja",
            "preprocessor_config.json", "generation_config.json"} <= set(os.listdir(d))
    got = dataclasses.asdict(load_config(d))
    want = dataclasses.asdict(cfg)
    got.pop("name"), want.pop("name")
    assert got == want                                      # every geometry field survives config.json, both layouts
    cfg2, tensors = load_checkpoint(d)
    w = random_weights(cfg, 11)
    assert set(tensors) == set(w)                           # hub names (v4: visual.*, model.layers.*) -> internal names
    for k in w:
        np.testing.assert_array_equal(as_f32(tensors[k]), meaning[k], err_msg=k)
        if not (fp8 and ".layers." in k and k.endswith("proj.weight")):
            np.testing.assert_array_equal(as_f32(tensors[k]), as_f32(w[k]), err_msg=k)   # bf16 tensors: the seeded weights themselves
    assert cli._checkpoint_is_fp8(d) == fp8
    if fp8:      # e4m3 codes + bf16 channel scales: within one fp8 step of the seeded weights
        k = "model.language_model.layers.0.mlp.down_proj.weight"
        err = np.abs(meaning[k] - as_f32(w[k])).max(axis=1) / np.abs(as_f32(w[k])).max(axis=1)
        assert err.max() < 2 ** -3.5
    assert cli.preprocessor_pixels(d) == (3136, 1003520)


def test_hf_tokenizer_on_the_synthetic_tokenizer_json(tmp_path):
    cfg = CONFIGS["tiny"]
    d = str(tmp_path / "ckpt")
    SC.write_checkpoint(d, cfg, 0)
    tok = S.HFTokenizer(os.path.join(d, "tokenizer.json"), cfg)
    sp = SC.special_tokens(cfg)
    assert (tok.im_start, tok.im_end) == (sp["<|im_start|>"], sp["<|im_end|>"]) and tok.im_end == cfg.eos_token_ids[0]
    text = "Return the plain text of this page.\nTitle: The é-test — done"
    ids = tok.encode(text)
    assert tok.decode(ids) == text and max(ids) < min(sp.values())
    assert len(ids) < len(text.encode())                     # merges exist: multi-byte tokens occur
    assert tok.encode("\n") == [tok.newline]
    assert tok.decode(ids + [tok.im_end]) == text            # specials are skipped in completions
    tb = tok.token_bytes()
    assert len(tb) == cfg.text.vocab_size and all(tb[i] == b"" for i in sp.values())
    assert b"".join(tb[i] for i in ids) == text.encode()     # guided decoding walks these bytes
    # the string of a special token inside text maps to its id (the template path relies on its own split, not on this)
    assert tok.tk.encode("a<|im_end|>b", add_special_tokens=False).ids[1] == tok.im_end


@pytest.mark.parametrize("where", ["jinja", "tokenizer_config", "both"])
def test_checkpoint_template_renders_the_reference_request_like_the_hand_coded_turns(tmp_path, where):
    """The checkpoint's own template (tokenizer_config.json / chat_template.jinja) on create_vision_message's request gives the
    ids of the hand-coded Qwen2-VL turns: default system turn, text part, <|vision_start|> T x <|image_pad|> <|vision_end|>,
    generation prompt."""
    cfg = CONFIGS["tiny"]
    d = str(tmp_path / "ckpt")
    SC.write_checkpoint(d, cfg, 0, template_file=where)
    tpl = S.load_chat_template(d)
    assert tpl and "<|vision_start|>" in tpl
    tok = S.HFTokenizer(os.path.join(d, "tokenizer.json"), cfg)
    req = reference_request(IP.synthetic_page(3, 84, 112))
    with_t = S.ChatFrontend(cfg, tok, chat_template=tpl).parse(req)
    hand = S.ChatFrontend(cfg, tok).parse(req)
    np.testing.assert_array_equal(with_t.input_ids, hand.input_ids)
    ids = with_t.input_ids.tolist()
    T = (84 // 14) * (112 // 14) // 4
    assert ids.count(cfg.image_token_id) == T and ids[0] == tok.im_start
    gen = [tok.im_start] + tok.encode("assistant") + [tok.newline]
    assert ids[-len(gen):] == gen
    np.testing.assert_array_equal(with_t.pixel_values, hand.pixel_values)
    # a system message of the caller's replaces the default one; two images render two placeholders
    req2 = reference_request(IP.synthetic_page(4, 56, 56))
    req2["messages"].insert(0, {"role": "system", "content": "You are an OCR engine."})
    req2["messages"][1]["content"].append({"type": "image_url", "image_url": {"url": IP.encode_png_data_url(IP.synthetic_page(5, 56, 84))}})
    a, b = S.ChatFrontend(cfg, tok, chat_template=tpl).parse(req2), S.ChatFrontend(cfg, tok).parse(req2)
    np.testing.assert_array_equal(a.input_ids, b.input_ids)
    assert len(a.grids) == 2


def test_admission_budget_is_sized_by_the_admission_not_by_the_slot_count():
    cfg = CONFIGS["Qwen2-VL-7B"]
    a = cli.parse_args(["serve", "/m", "--max-num-seqs", "32", "--max-model-len", "16384"])
    tokens, patches = cli.admission_budget(a, cfg, 12845056)        # hub preprocessor_config.json
    assert tokens == 16384 and patches == 4 * 16384 + 64 * 32
    # ... what round 3 allocated for the same flags
    assert 32 * (12845056 // 196 + 64) > 30 * patches and 32 * 16384 // 2 > 15 * tokens
    # eight class-default pages (4900 patches, ~1300 prompt tokens each) fit one admission
    assert 8 * 4900 <= patches and 8 * 1400 <= tokens
    b = cli.parse_args(["serve", "/m", "--max-num-seqs", "8", "--max-model-len", "4096", "--max-num-batched-tokens", "32768"])
    assert cli.admission_budget(b, cfg, 1003520) == (32768, 4 * 32768 + 512)
    c = cli.parse_args(["serve", "/m", "--max-model-len", "1024"])
    t, p = cli.admission_budget(c, CONFIGS["tiny"], 1003520)
    assert t == 16384 and p >= min(1003520 // 196, 4 * 1024) + 64      # the largest admissible single page always fits
    with pytest.raises(SystemExit):
        cli.parse_args(["serve", "/m", "--max-model-len", "8192", "--max-num-batched-tokens", "4096"])


def test_config_json_keeps_the_hub_key_names(tmp_path):
    d = SC.hf_config_dict(CONFIGS["Qwen2-VL-2B"], "v4")
    assert d["model_type"] == "qwen2_vl" and d["hidden_size"] == 1536 and d["vision_config"]["embed_dim"] == 1280
    assert d["rope_scaling"]["mrope_section"] == [16, 24, 24] and d["image_token_id"] == 151655
    d5 = SC.hf_config_dict(CONFIGS["Qwen2.5-VL-7B"], "v5", fp8=True)
    assert d5["text_config"]["num_key_value_heads"] == 4 and d5["vision_config"]["fullatt_block_indexes"] == [7, 15, 23, 31]
    assert "quantization_config" in d5 and json.dumps(d5)
    assert SC.special_tokens(CONFIGS["Qwen2-VL-2B"])["<|im_start|>"] == 151644       # Qwen's own id
    tk = SC.build_tokenizer(CONFIGS["Qwen2-VL-2B"])                                    # production ids: gaps filled with Qwen's names
    assert tk.token_to_id("<|image_pad|>") == 151655 and tk.token_to_id("<|vision_pad|>") == 151654
    assert tk.token_to_id("<|im_start|>") == 151644 and tk.get_vocab_size() == 151657
