"""Write a random-init Qwen2-VL / Qwen2.5-VL checkpoint in the Hugging Face HUB LAYOUT.

    python -m karanta_ocr_amd.tools.synthetic_checkpoint OUT_DIR [--config tiny] [--seed 0] [--layout v4|v5] [--fp8]

There is no network in the build container and no real checkpoint on disk, yet what a deployment hands to the server is
a directory (`vllm serve <model>`: /root/reference/karanta/pipeline.py:707-742; `--model <dir>`:
/root/reference/scripts/start_multiple_vllm_servers.sh:283-294).  This module produces such a directory from a
:class:`ModelConfig` and a seed, so that the whole deployment path — `cli.make_server` -> `weights.load_checkpoint` ->
`Engine` -> `serving.HFTokenizer` -> the checkpoint's own chat template -> HTTP — runs end to end on files with the names, tensor
names and formats a hub checkpoint has:

  config.json                  the transformers-4 flat layout (what Qwen2-VL-*-Instruct ships) or the transformers-5 nested one
  model.safetensors            HF tensor names (`visual.*`, `model.layers.*`, `lm_head.weight` for layout v4; `model.visual.*`,
                               `model.language_model.*` for v5), bf16; --fp8: the decoder Linears as float8_e4m3fn `weight` +
                               bf16 `weight_scale` [N, 1] (the compressed-tensors layout of allenai/olmOCR-7B-0725-FP8,
                               /root/reference/karanta/constants.py:23) with a `quantization_config` in config.json
  tokenizer.json               a byte-level BPE built with the `tokenizers` library: the 256 byte tokens, merges of frequent
                               English letter pairs, and Qwen's special tokens at the ids config.json names
  tokenizer_config.json        with `chat_template` (the Qwen2-VL template's structure), also written as chat_template.jinja
  preprocessor_config.json     min_pixels / max_pixels
  generation_config.json       eos / pad ids

The weights are `weights.random_weights(cfg, seed)` — the same tensors every test regenerates from the seed, so an engine loaded
from the directory can be compared with the oracle run on `random_weights(cfg, seed)` directly."""
from __future__ import annotations

import argparse
import json
import os
from typing import Dict, Optional

import numpy as np

from ..config import CONFIGS, ModelConfig
from ..weights import as_f32, f32_to_fp8_e4m3, fp8_e4m3_to_f32, random_weights

# The structure of Qwen2-VL's chat template: default system turn, per-message role header, image parts as
# <|vision_start|><|image_pad|><|vision_end|> in place, `add_generation_prompt`.
CHAT_TEMPLATE = (
    "{% set image_count = namespace(value=0) %}"
    "{% for message in messages %}"
    "{% if loop.first and message['role'] != 'system' %}<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n{% endif %}"
    "<|im_start|>{{ message['role'] }}\n"
    "{% if message['content'] is string %}{{ message['content'] }}<|im_end|>\n"
    "{% else %}{% for content in message['content'] %}"
    "{% if content['type'] == 'image' or 'image' in content or 'image_url' in content %}"
    "{% set image_count.value = image_count.value + 1 %}"
    "{% if add_vision_id %}Picture {{ image_count.value }}: {% endif %}<|vision_start|><|image_pad|><|vision_end|>"
    "{% elif 'text' in content %}{{ content['text'] }}{% endif %}"
    "{% endfor %}<|im_end|>\n{% endif %}"
    "{% endfor %}"
    "{% if add_generation_prompt %}<|im_start|>assistant\n{% endif %}"
)

_V5_TO_V4 = (("model.visual.", "visual."), ("model.language_model.", "model."))


def hf_config_dict(cfg: ModelConfig, layout: str = "v4", fp8: bool = False) -> dict:
    """config.json of `cfg` (inverse of config.from_hf_config_dict)."""
    t, v = cfg.text, cfg.vision
    text = {"hidden_size": t.hidden_size, "intermediate_size": t.intermediate_size, "num_hidden_layers": t.num_layers,
            "num_attention_heads": t.num_heads, "num_key_value_heads": t.num_kv_heads, "vocab_size": t.vocab_size,
            "rms_norm_eps": t.rms_norm_eps, "rope_theta": t.rope_theta, "hidden_act": "silu", "max_position_embeddings": 32768,
            "rope_scaling": {"type": "mrope", "mrope_section": list(t.mrope_section)}}
    if t.head_dim * t.num_heads != t.hidden_size:
        text["head_dim"] = t.head_dim
    common = {"depth": v.depth, "num_heads": v.num_heads, "patch_size": v.patch_size, "spatial_merge_size": v.spatial_merge_size,
              "temporal_patch_size": v.temporal_patch_size, "in_chans": v.in_channels}
    if v.variant == "qwen2_5":
        vision = {**common, "hidden_size": v.embed_dim, "out_hidden_size": v.hidden_size, "intermediate_size": v.intermediate_size,
                  "window_size": v.window_size, "fullatt_block_indexes": list(v.fullatt_block_indexes), "hidden_act": "silu"}
        arch, mtype = "Qwen2_5_VLForConditionalGeneration", "qwen2_5_vl"
    else:
        vision = {**common, "embed_dim": v.embed_dim, "hidden_size": v.hidden_size, "mlp_ratio": v.mlp_ratio, "hidden_act": "quick_gelu"}
        arch, mtype = "Qwen2VLForConditionalGeneration", "qwen2_vl"
    top = {"architectures": [arch], "model_type": mtype, "torch_dtype": "bfloat16", "tie_word_embeddings": t.tie_word_embeddings,
           "image_token_id": cfg.image_token_id, "video_token_id": cfg.video_token_id,
           "vision_start_token_id": cfg.vision_start_token_id, "vision_end_token_id": cfg.vision_end_token_id,
           "eos_token_id": list(cfg.eos_token_ids), "pad_token_id": cfg.pad_token_id, "bos_token_id": cfg.eos_token_ids[-1],
           "vision_config": vision}
    d = {**top, "text_config": text} if layout == "v5" else {**top, **text}
    if fp8:
        d["quantization_config"] = {"quant_method": "compressed-tensors", "format": "float-quantized",
                                    "config_groups": {"group_0": {"targets": ["Linear"], "weights": {
                                        "num_bits": 8, "type": "float", "strategy": "channel", "symmetric": True, "dynamic": False}}},
                                    "ignore": ["re:visual.*", "lm_head"]}
    return d


def special_tokens(cfg: ModelConfig) -> Dict[str, int]:
    """Qwen's special-token strings -> the ids this config uses for them (production ids for the bench models; the test
    configs pack them at the top of their 512-token vocabulary)."""
    eos = cfg.eos_token_ids
    im_end, eot = eos[0], eos[-1]
    sp = {"<|endoftext|>": eot, "<|im_end|>": im_end, "<|vision_start|>": cfg.vision_start_token_id,
          "<|vision_end|>": cfg.vision_end_token_id, "<|image_pad|>": cfg.image_token_id, "<|video_pad|>": cfg.video_token_id}
    used = set(sp.values())
    im_start = im_end - 1 if (im_end - 1 not in used and im_end - 1 > 255) else max(used) + 1   # Qwen: 151644 = im_end - 1
    sp["<|im_start|>"] = im_start
    if len(set(sp.values())) != len(sp) or max(sp.values()) >= cfg.text.vocab_size:
        raise ValueError(f"special token ids collide or exceed the vocabulary: {sp}")
    return sp


def build_tokenizer(cfg: ModelConfig):
    """A byte-level BPE `tokenizers.Tokenizer` whose special tokens sit at the ids of `special_tokens(cfg)`.  Ids 0..255 are
    the byte tokens; the ids between them and the first special token are merges of lowercase letter pairs (then of pairs
    of those) — enough structure for multi-byte tokens to occur in every text; ids above the last special stay unused, as
    in Qwen's own vocabulary (151 665 used of 151 936)."""
    from tokenizers import AddedToken, Tokenizer, decoders, models, pre_tokenizers

    sp = dict(special_tokens(cfg))
    first_special = min(sp.values())
    # the `tokenizers` library numbers added tokens consecutively after the model's vocabulary: ids between the named specials
    # get Qwen's other special tokens where the ids are Qwen's (151646..151651, 151654), reserved names elsewhere
    qwen_other = {151646: "<|object_ref_start|>", 151647: "<|object_ref_end|>", 151648: "<|box_start|>", 151649: "<|box_end|>",
                  151650: "<|quad_start|>", 151651: "<|quad_end|>", 151654: "<|vision_pad|>"}
    taken = set(sp.values())
    for i in range(first_special, max(taken) + 1):
        if i not in taken:
            sp[qwen_other.get(i, f"<|reserved_{i}|>")] = i
    alphabet = sorted(pre_tokenizers.ByteLevel.alphabet())
    vocab = {ch: i for i, ch in enumerate(alphabet)}
    merges = []
    letters = "etaoinshrdlucmfwypvbgkqjxz"
    space = "Ġ"                       # the byte-level image of ' '
    cands = [(space, a) for a in letters] + [(a, b) for a in letters for b in letters]
    cands += [(space + a, b) for a in letters[:12] for b in letters[:12]]
    for a, b in cands:
        if len(vocab) >= first_special:
            break
        if a in vocab and b in vocab and a + b not in vocab:
            vocab[a + b] = len(vocab)
            merges.append((a, b))
    k = 0
    while len(vocab) < first_special:     # very large vocabularies: filler tokens that no text produces
        vocab[f"<|filler_{k}|>"] = len(vocab)
        k += 1
    tk = Tokenizer(models.BPE(vocab=vocab, merges=merges))
    tk.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False, use_regex=True)
    tk.decoder = decoders.ByteLevel()
    by_id = sorted(sp.items(), key=lambda kv: kv[1])
    tk.add_special_tokens([AddedToken(s, special=True, normalized=False) for s, _ in by_id])
    for s, i in by_id:
        if tk.token_to_id(s) != i:
            raise RuntimeError(f"{s} landed at id {tk.token_to_id(s)}, wanted {i}")
    return tk


def write_checkpoint(out_dir: str, cfg: ModelConfig, seed: int = 0, layout: str = "v4", fp8: bool = False,
                     min_pixels: Optional[int] = 3136, max_pixels: Optional[int] = 1003520, template_file: str = "both",
                     weights: Optional[Dict[str, np.ndarray]] = None) -> Dict[str, np.ndarray]:
    """Write the directory; returns the fp32 weights the checkpoint MEANS (for --fp8: the dequantised values), keyed by the
    internal (transformers-5) names — what an oracle run on this checkpoint uses."""
    import torch
    from safetensors.torch import save_file

    if layout not in ("v4", "v5"):
        raise ValueError("layout: v4 (hub names of Qwen2-VL-*-Instruct) or v5 (transformers 5 names)")
    os.makedirs(out_dir, exist_ok=True)
    w = weights if weights is not None else random_weights(cfg, seed)
    meaning: Dict[str, np.ndarray] = {}
    tensors = {}
    for name, arr in w.items():
        a = as_f32(arr)
        hub = name
        if layout == "v4":
            for new, old in _V5_TO_V4:
                if hub.startswith(new):
                    hub = old + hub[len(new):]
                    break
        is_dec_linear = (".layers." in name and name.endswith("proj.weight"))
        if fp8 and is_dec_linear:
            amax = np.abs(a).max(axis=1)
            scale = np.where(amax > 0, amax / np.float32(448.0), np.float32(1.0)).astype(np.float32)
            scale = as_f32(_bf16(scale))                       # the checkpoint stores the scales as bf16
            codes = f32_to_fp8_e4m3(a / scale[:, None])
            tensors[hub] = torch.from_numpy(codes.copy()).view(torch.float8_e4m3fn)
            tensors[hub[: -len("weight")] + "weight_scale"] = torch.from_numpy(scale.reshape(-1, 1).copy()).to(torch.bfloat16)
            meaning[name] = fp8_e4m3_to_f32(codes) * scale[:, None]
        else:
            t = torch.from_numpy(np.ascontiguousarray(a)).to(torch.bfloat16)
            tensors[hub] = t
            meaning[name] = t.float().numpy()
    save_file(tensors, os.path.join(out_dir, "model.safetensors"), metadata={"format": "pt"})
    with open(os.path.join(out_dir, "config.json"), "w") as f:
        json.dump(hf_config_dict(cfg, layout, fp8), f, indent=1)
    tk = build_tokenizer(cfg)
    tk.save(os.path.join(out_dir, "tokenizer.json"))
    sp = special_tokens(cfg)
    tcfg = {"tokenizer_class": "Qwen2Tokenizer", "eos_token": "<|im_end|>", "pad_token": "<|endoftext|>", "bos_token": None,
            "additional_special_tokens": sorted(sp, key=sp.get), "model_max_length": 32768}
    if template_file in ("both", "tokenizer_config"):
        tcfg["chat_template"] = CHAT_TEMPLATE
    with open(os.path.join(out_dir, "tokenizer_config.json"), "w") as f:
        json.dump(tcfg, f, indent=1)
    if template_file in ("both", "jinja"):
        with open(os.path.join(out_dir, "chat_template.jinja"), "w") as f:
            f.write(CHAT_TEMPLATE)
    pre = {"image_processor_type": "Qwen2VLImageProcessor", "patch_size": cfg.vision.patch_size,
           "merge_size": cfg.vision.spatial_merge_size, "temporal_patch_size": cfg.vision.temporal_patch_size,
           "image_mean": [0.48145466, 0.4578275, 0.40821073], "image_std": [0.26862954, 0.26130258, 0.27577711]}
    if min_pixels:
        pre["min_pixels"] = int(min_pixels)
    if max_pixels:
        pre["max_pixels"] = int(max_pixels)
    with open(os.path.join(out_dir, "preprocessor_config.json"), "w") as f:
        json.dump(pre, f, indent=1)
    with open(os.path.join(out_dir, "generation_config.json"), "w") as f:
        json.dump({"eos_token_id": list(cfg.eos_token_ids), "pad_token_id": cfg.pad_token_id, "do_sample": False}, f, indent=1)
    return meaning


def _bf16(a: np.ndarray) -> np.ndarray:
    from ..weights import to_bf16_bits
    return to_bf16_bits(np.asarray(a, np.float32))


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("out_dir")
    ap.add_argument("--config", default="tiny", choices=sorted(CONFIGS))
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--layout", default="v4", choices=("v4", "v5"))
    ap.add_argument("--fp8", action="store_true")
    ap.add_argument("--max-pixels", type=int, default=1003520)
    a = ap.parse_args(argv)
    write_checkpoint(a.out_dir, CONFIGS[a.config], a.seed, a.layout, a.fp8, max_pixels=a.max_pixels)
    print(a.out_dir)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
