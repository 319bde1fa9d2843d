#!/usr/bin/env python3
"""Generate the golden fixtures that pin ``oracle/qwen2vl_oracle.py``.

Run in the BUILD container only (needs `transformers`; written against 5.15.0):

    python tests/golden/make_golden.py

The reference (karanta-ocr) has no tests or vectors for the VLM hot path,
and its arithmetic lives in third-party code (vLLM / Hugging Face transformers,
/root/reference/karanta/training/test_trained_model.py:6-7,76-99).  The fixtures are therefore
outputs of the Hugging Face Qwen2-VL implementation on seeded tiny configs:

* fp32, ``attn_implementation="eager"``, ``mm_token_type_ids`` always passed;
* weights from ``karanta_ocr_amd.weights.random_weights(cfg, seed)`` (bf16-representable,
  regenerated from the seed everywhere — never shipped);
* inputs seeded here and stored next to the expected outputs.

Nothing from /root/reference or from transformers' sources is copied: only numbers.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from karanta_ocr_amd.config import CONFIGS, ModelConfig  # noqa: E402
from karanta_ocr_amd.weights import random_weights  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def hf_model(cfg: ModelConfig, seed: int, dtype=torch.float32):
    from transformers import Qwen2VLConfig, Qwen2VLForConditionalGeneration

    t, v = cfg.text, cfg.vision
    hcfg = Qwen2VLConfig(
        text_config=dict(
            hidden_size=t.hidden_size, intermediate_size=t.intermediate_size,
            num_hidden_layers=t.num_layers, num_attention_heads=t.num_heads,
            num_key_value_heads=t.num_kv_heads, vocab_size=t.vocab_size,
            max_position_embeddings=8192, rms_norm_eps=t.rms_norm_eps,
            rope_parameters={"rope_type": "default", "rope_theta": t.rope_theta,
                             "mrope_section": list(t.mrope_section)},
            tie_word_embeddings=t.tie_word_embeddings,
        ),
        vision_config=dict(
            depth=v.depth, embed_dim=v.embed_dim, hidden_size=v.hidden_size, num_heads=v.num_heads,
            mlp_ratio=v.mlp_ratio, patch_size=v.patch_size, spatial_merge_size=v.spatial_merge_size,
            temporal_patch_size=v.temporal_patch_size, in_channels=v.in_channels,
        ),
        image_token_id=cfg.image_token_id, video_token_id=cfg.video_token_id,
        vision_start_token_id=cfg.vision_start_token_id, vision_end_token_id=cfg.vision_end_token_id,
        tie_word_embeddings=t.tie_word_embeddings,
    )
    hcfg._attn_implementation = "eager"
    model = Qwen2VLForConditionalGeneration(hcfg).eval()
    w = random_weights(cfg, seed)
    sd = {k: torch.from_numpy(np.asarray(a)) for k, a in w.items()}
    if t.tie_word_embeddings:
        sd["lm_head.weight"] = sd["model.language_model.embed_tokens.weight"]
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("inv_freq" in m for m in missing), missing
    model = model.to(dtype)
    model.config._attn_implementation = "eager"
    for sub in (model.model.visual, model.model.language_model):
        sub.config._attn_implementation = "eager"
    model.generation_config.eos_token_id = list(cfg.eos_token_ids)
    model.generation_config.pad_token_id = cfg.pad_token_id
    model.generation_config.do_sample = False
    return model, w


def synth_image(seed: int, h: int, w: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    img = np.clip(rng.normal(238, 6, size=(h, w, 3)), 0, 255)
    # a few dark "text" runs
    for _ in range(max(2, h // 12)):
        y = int(rng.integers(0, max(1, h - 6)))
        x0 = int(rng.integers(0, max(1, w - 20)))
        img[y:y + 4, x0:x0 + int(rng.integers(10, max(11, w // 2)))] = rng.integers(15, 70)
    return img.astype(np.uint8)


def hf_pixels(imgs, max_pixels=None):
    from transformers.models.qwen2_vl.image_processing_pil_qwen2_vl import Qwen2VLImageProcessorPil
    from PIL import Image

    kw = {}
    if max_pixels is not None:
        kw = dict(min_pixels=56 * 56, max_pixels=max_pixels)
    proc = Qwen2VLImageProcessorPil(**kw)
    out = proc(images=[Image.fromarray(i) for i in imgs], return_tensors="pt")
    return out["pixel_values"].float().numpy(), out["image_grid_thw"].numpy()


def build_prompt(cfg: ModelConfig, rng, grids, n_pre=4, n_mid=3, n_post=6):
    """[text*n_pre, (<vs> img*T <ve> text*n_mid)*, text*n_post] with random text ids."""
    merge = cfg.vision.spatial_merge_size
    ids = list(rng.integers(0, 400, size=n_pre))
    for gi, (t, gh, gw) in enumerate(grids):
        ids.append(cfg.vision_start_token_id)
        ids += [cfg.image_token_id] * int(t * (gh // merge) * (gw // merge))
        ids.append(cfg.vision_end_token_id)
        if gi != len(grids) - 1:
            ids += list(rng.integers(0, 400, size=n_mid))
    ids += list(rng.integers(0, 400, size=n_post))
    return np.asarray(ids, dtype=np.int64)[None, :]


def main():
    from transformers.models.qwen2_vl.image_processing_pil_qwen2_vl import smart_resize

    torch.manual_seed(0)
    torch.set_num_threads(8)
    fx = {}

    # ---- 1. smart_resize table
    sizes = [(1024, 1024), (2200, 1700), (1422, 1056), (956, 1288), (56, 56), (28, 5000), (1288, 956), (37, 4000),
             (2048, 1448), (300, 77)]
    tab = []
    for (h, w) in sizes:
        for mp in (1003520, 12845056):
            tab.append([h, w, mp, *smart_resize(h, w, factor=28, min_pixels=3136, max_pixels=mp)])
    fx["smart_resize_table"] = np.asarray(tab, dtype=np.int64)

    # ---- 2. image preprocessing (resize + normalize + patchify) through the HF PIL processor
    img_a = synth_image(11, 56, 84)       # already a multiple of 28: no resize
    img_b = synth_image(12, 100, 150)     # resized to 112x140 (bicubic)
    pv_a, g_a = hf_pixels([img_a])
    pv_b, g_b = hf_pixels([img_b])
    fx["pre_img_a"], fx["pre_pv_a"], fx["pre_grid_a"] = img_a, pv_a, g_a
    fx["pre_img_b"], fx["pre_pv_b"], fx["pre_grid_b"] = img_b, pv_b, g_b
    img_c = synth_image(13, 640, 480)     # max_pixels clamp path
    pv_c, g_c = hf_pixels([img_c], max_pixels=28 * 28 * 64)
    fx["pre_img_c"], fx["pre_pv_c_sum"], fx["pre_grid_c"] = img_c, np.asarray([pv_c.sum(), np.abs(pv_c).sum()]), g_c
    fx["pre_pv_c_head"] = pv_c[:8]

    for cname, seed in (("tiny", 1234), ("tiny-gqa", 4321)):
        cfg = CONFIGS[cname]
        model, w = hf_model(cfg, seed)
        P = cname.replace("-", "_") + "__"
        rng = np.random.default_rng(seed)

        # ---- 3-6. vision tower, two images → two attention segments
        imgs = [synth_image(seed + 1, 56, 84), synth_image(seed + 2, 84, 56)]
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
            fx[P + f"vit_block{i}"] = caps[f"block{i}"]
        fx[P + "vit_merged"] = vo.pooler_output.float().numpy()
        from transformers.vision_utils import get_vision_position_ids
        pid = get_vision_position_ids(torch.from_numpy(grid), cfg.vision.spatial_merge_size)
        fx[P + "vit_pos_ids"] = pid.numpy()
        rp = vis.rotary_pos_emb(pid)
        emb = torch.cat((rp, rp), dim=-1)
        fx[P + "vit_cos"], fx[P + "vit_sin"] = emb.cos().numpy(), emb.sin().numpy()

        # ---- 7. get_rope_index: one-image and two-image prompts
        for tag, grids in (("1img", [(1, 4, 6)]), ("2img", [(1, 4, 6), (1, 8, 4)])):
            ids = build_prompt(cfg, rng, grids)
            mm = (ids == cfg.image_token_id).astype(np.int32)
            pos, delta = model.model.get_rope_index(
                torch.from_numpy(ids), mm_token_type_ids=torch.from_numpy(mm),
                image_grid_thw=torch.tensor(grids))
            fx[P + f"rope_{tag}_ids"] = ids
            fx[P + f"rope_{tag}_grid"] = np.asarray(grids, dtype=np.int64)
            fx[P + f"rope_{tag}_pos"] = pos.numpy()
            fx[P + f"rope_{tag}_delta"] = delta.numpy().reshape(-1)

        # ---- 8. RMSNorm
        x = torch.from_numpy(rng.standard_normal((5, cfg.text.hidden_size)).astype(np.float32) * 3)
        with torch.no_grad():
            y = model.model.language_model.norm(x)
        fx[P + "rms_x"], fx[P + "rms_y"] = x.numpy(), y.numpy()

        # ---- 9. M-RoPE cos/sin + apply
        from transformers.models.qwen2_vl.modeling_qwen2_vl import apply_multimodal_rotary_pos_emb
        pos3 = torch.from_numpy(rng.integers(0, 3000, size=(3, 1, 7)))
        q = torch.from_numpy(rng.standard_normal((1, cfg.text.num_heads, 7, cfg.text.head_dim)).astype(np.float32))
        k = torch.from_numpy(rng.standard_normal((1, cfg.text.num_kv_heads, 7, cfg.text.head_dim)).astype(np.float32))
        with torch.no_grad():
            cos, sin = model.model.language_model.rotary_emb(q, pos3)
            qe, ke = apply_multimodal_rotary_pos_emb(q, k, cos, sin, list(cfg.text.mrope_section))
        fx[P + "mrope_pos"], fx[P + "mrope_q"], fx[P + "mrope_k"] = pos3.numpy(), q.numpy(), k.numpy()
        fx[P + "mrope_qe"], fx[P + "mrope_ke"] = qe.numpy(), ke.numpy()

        # ---- 10/11. end to end: prompt logits + greedy ids (generate) + per-step logits
        img = synth_image(seed + 3, 112, 168)
        pv1, g1 = hf_pixels([img])
        ids = build_prompt(cfg, rng, [tuple(int(v) for v in g1[0])])
        mm = (ids == cfg.image_token_id).astype(np.int32)
        n_new = 16
        with torch.no_grad():
            out = model(input_ids=torch.from_numpy(ids), pixel_values=torch.from_numpy(pv1),
                        image_grid_thw=torch.from_numpy(g1), mm_token_type_ids=torch.from_numpy(mm),
                        output_hidden_states=True)
            model.model.rope_deltas = None
            gen = model.generate(input_ids=torch.from_numpy(ids), pixel_values=torch.from_numpy(pv1),
                                 image_grid_thw=torch.from_numpy(g1), mm_token_type_ids=torch.from_numpy(mm),
                                 attention_mask=torch.ones_like(torch.from_numpy(ids)),
                                 max_new_tokens=n_new, do_sample=False,
                                 output_scores=True, return_dict_in_generate=True)
        fx[P + "e2e_img"] = img
        fx[P + "e2e_pixel_values"], fx[P + "e2e_grid"] = pv1, g1
        fx[P + "e2e_input_ids"] = ids
        fx[P + "e2e_prompt_logits"] = out.logits[0].float().numpy()          # [P, V]
        fx[P + "e2e_hidden_layer0_in"] = out.hidden_states[0][0].float().numpy()   # embeds after scatter
        fx[P + "e2e_hidden_layer1_in"] = out.hidden_states[1][0].float().numpy()   # after decoder layer 0
        fx[P + "e2e_gen_ids"] = gen.sequences[:, ids.shape[1]:].numpy()
        fx[P + "e2e_gen_scores"] = torch.stack(gen.scores, dim=1)[0].float().numpy()  # [n_new, V]

        # ---- 12. the same model in bf16 on CPU (HF's own bf16 rounding points): tolerance study
        model_bf, _ = hf_model(cfg, seed, dtype=torch.bfloat16)
        with torch.no_grad():
            ob = model_bf(input_ids=torch.from_numpy(ids), pixel_values=torch.from_numpy(pv1).to(torch.bfloat16),
                          image_grid_thw=torch.from_numpy(g1), mm_token_type_ids=torch.from_numpy(mm))
        fx[P + "e2e_prompt_logits_hf_bf16_last"] = ob.logits[0, -1].float().numpy()

    np.savez_compressed(os.path.join(OUT, "qwen2vl_tiny_golden.npz"), **fx)
    total = sum(v.nbytes for v in fx.values())
    print(f"wrote {len(fx)} arrays, {total/1e6:.2f} MB raw")
    for k, v in fx.items():
        print(f"  {k:45s} {str(v.dtype):8s} {v.shape}")


if __name__ == "__main__":
    main()
