#!/usr/bin/env python3
"""Golden fixtures for the Qwen2.5-VL variant of the oracle (the family of the reference's own
fine-tunes, /root/reference/configs/training/ocr/karanta_set_qwen_2_5_3B_vl.yaml:2).

Run in the BUILD container only (needs `transformers`; written against 5.15.0):

    python tests/golden/make_golden_qwen2_5.py

Outputs of the Hugging Face ``Qwen2_5_VLForConditionalGeneration`` (fp32, eager attention) on the seeded
``tiny-2.5`` config with ``karanta_ocr_amd.weights.random_weights``: window index, every vision block, merged image
tokens for a two-image batch whose grids do and do not divide into whole windows; end-to-end prompt logits, greedy
ids and per-step scores.  Only numbers are stored.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from karanta_ocr_amd.config import CONFIGS  # noqa: E402
from karanta_ocr_amd.weights import random_weights  # noqa: E402
from make_golden import build_prompt, hf_pixels, synth_image  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def hf_model(cfg, seed):
    from transformers import Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration

    t, v = cfg.text, cfg.vision
    hcfg = Qwen2_5_VLConfig(
        text_config=dict(
            hidden_size=t.hidden_size, intermediate_size=t.intermediate_size, num_hidden_layers=t.num_layers,
            num_attention_heads=t.num_heads, num_key_value_heads=t.num_kv_heads, vocab_size=t.vocab_size,
            max_position_embeddings=8192, rms_norm_eps=t.rms_norm_eps,
            rope_parameters={"rope_type": "default", "rope_theta": t.rope_theta, "mrope_section": list(t.mrope_section)},
            tie_word_embeddings=t.tie_word_embeddings, use_sliding_window=False,
        ),
        vision_config=dict(
            depth=v.depth, hidden_size=v.embed_dim, out_hidden_size=v.hidden_size, num_heads=v.num_heads,
            intermediate_size=v.intermediate_size, patch_size=v.patch_size, spatial_merge_size=v.spatial_merge_size,
            temporal_patch_size=v.temporal_patch_size, in_channels=v.in_channels, window_size=v.window_size,
            fullatt_block_indexes=list(v.fullatt_block_indexes), hidden_act="silu",
        ),
        image_token_id=cfg.image_token_id, video_token_id=cfg.video_token_id,
        vision_start_token_id=cfg.vision_start_token_id, vision_end_token_id=cfg.vision_end_token_id,
        tie_word_embeddings=t.tie_word_embeddings,
    )
    hcfg._attn_implementation = "eager"
    model = Qwen2_5_VLForConditionalGeneration(hcfg).eval()
    w = random_weights(cfg, seed)
    sd = {k: torch.from_numpy(np.asarray(a)) for k, a in w.items()}
    if t.tie_word_embeddings:
        sd["lm_head.weight"] = sd["model.language_model.embed_tokens.weight"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("inv_freq" in m for m in missing), missing
    model.config._attn_implementation = "eager"
    for sub in (model.model.visual, model.model.language_model):
        sub.config._attn_implementation = "eager"
    model.generation_config.eos_token_id = list(cfg.eos_token_ids)
    model.generation_config.pad_token_id = cfg.pad_token_id
    model.generation_config.do_sample = False
    return model, w


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    cfg, seed = CONFIGS["tiny-2.5"], 2525
    model, _ = hf_model(cfg, seed)
    rng = np.random.default_rng(seed)
    fx = {}
    P = "tiny_2_5__"
    # vision tower: 84x140 -> 6x10 patches = 3x5 merged units (windows of 2x2 units: ragged right / bottom edges),
    # 112x112 -> 8x8 patches = 4x4 units (divides exactly: the padding-by-a-full-window path)
    imgs = [synth_image(seed + 1, 84, 140), synth_image(seed + 2, 112, 112)]
    pv, grid = hf_pixels(imgs)
    fx[P + "vit_pixel_values"], fx[P + "vit_grid"] = pv, grid
    vis = model.model.visual
    caps = {}

    def cap(name):
        def hook(mod, args, out):
            caps[name] = (out[0] if isinstance(out, tuple) else out).detach().float().numpy()
        return hook

    hs = [vis.patch_embed.register_forward_hook(cap("patch_embed"))]
    for i, blk in enumerate(vis.blocks):
        hs.append(blk.register_forward_hook(cap(f"block{i}")))
    with torch.no_grad():
        vo = vis(torch.from_numpy(pv), grid_thw=torch.from_numpy(grid))
    for h in hs:
        h.remove()
    fx[P + "vit_patch_embed"] = caps["patch_embed"]
    for i in range(cfg.vision.depth):
        fx[P + f"vit_block{i}"] = caps[f"block{i}"]            # in window order
    fx[P + "vit_merged"] = vo.pooler_output.float().numpy()    # back in image order
    from transformers.vision_utils import get_vision_window_index
    widx, cu = get_vision_window_index(torch.from_numpy(grid), spatial_merge_size=cfg.vision.spatial_merge_size,
                                       window_size=cfg.vision.window_size, patch_size=cfg.vision.patch_size)
    fx[P + "vit_window_index"], fx[P + "vit_cu_window_seqlens"] = widx.numpy(), cu.numpy()

    # end to end
    img = synth_image(seed + 3, 112, 168)
    pv1, g1 = hf_pixels([img])
    ids = build_prompt(cfg, rng, [tuple(int(v) for v in g1[0])])
    mm = (ids == cfg.image_token_id).astype(np.int32)
    with torch.no_grad():
        out = model(input_ids=torch.from_numpy(ids), pixel_values=torch.from_numpy(pv1),
                    image_grid_thw=torch.from_numpy(g1), mm_token_type_ids=torch.from_numpy(mm))
        model.model.rope_deltas = None
        gen = model.generate(input_ids=torch.from_numpy(ids), pixel_values=torch.from_numpy(pv1),
                             image_grid_thw=torch.from_numpy(g1), mm_token_type_ids=torch.from_numpy(mm),
                             attention_mask=torch.ones_like(torch.from_numpy(ids)), max_new_tokens=16, do_sample=False,
                             output_scores=True, return_dict_in_generate=True)
    fx[P + "e2e_img"] = img
    fx[P + "e2e_pixel_values"], fx[P + "e2e_grid"] = pv1, g1
    fx[P + "e2e_input_ids"] = ids
    fx[P + "e2e_prompt_logits"] = out.logits[0].float().numpy()
    fx[P + "e2e_gen_ids"] = gen.sequences[:, ids.shape[1]:].numpy()
    fx[P + "e2e_gen_scores"] = torch.stack(gen.scores, dim=1)[0].float().numpy()
    np.savez_compressed(os.path.join(OUT, "qwen2_5vl_tiny_golden.npz"), **fx)
    print(f"wrote {len(fx)} arrays, {sum(v.nbytes for v in fx.values())/1e6:.2f} MB raw")


if __name__ == "__main__":
    main()
