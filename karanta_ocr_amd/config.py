"""Model geometry for the Qwen2-VL family served on the karanta OCR hot path.

The reference never states these numbers: it hands a model *name* to ``vllm serve``
(/root/reference/karanta/pipeline.py:707-742) or to ``from_pretrained``
(/root/reference/karanta/training/test_trained_model.py:15-22).  The dimensions
below are the public Qwen2-VL model-card values collected in SURVEY.md §8; a real
checkpoint's ``config.json`` overrides them through :func:`from_hf_config_dict`.
"""
from __future__ import annotations

from dataclasses import dataclass, field, replace
from typing import Dict, List, Tuple


@dataclass(frozen=True)
class VisionConfig:
    depth: int = 32
    embed_dim: int = 1280
    num_heads: int = 16
    mlp_ratio: int = 4
    patch_size: int = 14
    spatial_merge_size: int = 2
    temporal_patch_size: int = 2
    in_channels: int = 3
    hidden_size: int = 1536  # merger output = decoder width
    # "qwen2": LayerNorm + fc1/QuickGELU/fc2, full attention per image (Qwen2-VL).
    # "qwen2_5": RMSNorm + biased SwiGLU MLP of width `intermediate_size`, attention inside `window_size`-pixel
    # windows except in the `fullatt_block_indexes` blocks, tokens reordered window by window (Qwen2.5-VL — the
    # family of the reference's own fine-tunes, configs/training/ocr/karanta_set_qwen_2_5_3B_vl.yaml:2, and of
    # olmOCR-7B-0725, karanta/constants.py:22-24).
    variant: str = "qwen2"
    intermediate_size: int = 0
    window_size: int = 112
    fullatt_block_indexes: Tuple[int, ...] = ()

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.num_heads

    @property
    def mlp_dim(self) -> int:
        return self.intermediate_size if self.variant == "qwen2_5" else self.embed_dim * self.mlp_ratio

    @property
    def mlp_dim_padded(self) -> int:
        """MLP width rounded up to the GEMM's BK = 64 (Qwen2.5-VL: 3420 -> 3456; the padding rows / columns are zero)."""
        return (self.mlp_dim + 63) // 64 * 64

    @property
    def window_merge_units(self) -> int:
        """Side of an attention window in merged (2x2 patch) units."""
        return self.window_size // self.spatial_merge_size // self.patch_size

    @property
    def patch_dim(self) -> int:
        return self.in_channels * self.temporal_patch_size * self.patch_size * self.patch_size

    @property
    def patch_dim_padded(self) -> int:
        """K of the patch-embed GEMM rounded up to the GEMM's BK=64."""
        return (self.patch_dim + 63) // 64 * 64

    @property
    def merge_dim(self) -> int:
        return self.embed_dim * self.spatial_merge_size**2


@dataclass(frozen=True)
class TextConfig:
    hidden_size: int = 1536
    intermediate_size: int = 8960
    num_layers: int = 28
    num_heads: int = 12
    num_kv_heads: int = 2
    head_dim: int = 128
    vocab_size: int = 151936
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1e6
    mrope_section: Tuple[int, int, int] = (16, 24, 24)
    tie_word_embeddings: bool = True

    @property
    def q_dim(self) -> int:
        return self.num_heads * self.head_dim

    @property
    def kv_dim(self) -> int:
        return self.num_kv_heads * self.head_dim

    @property
    def qkv_dim(self) -> int:
        return self.q_dim + 2 * self.kv_dim

    @property
    def kv_bytes_per_token(self) -> int:
        """bf16 K+V bytes cached per token over all layers."""
        return 2 * self.num_layers * self.num_kv_heads * self.head_dim * 2


@dataclass(frozen=True)
class ModelConfig:
    name: str
    vision: VisionConfig
    text: TextConfig
    image_token_id: int = 151655
    video_token_id: int = 151656
    vision_start_token_id: int = 151652
    vision_end_token_id: int = 151653
    eos_token_ids: Tuple[int, ...] = (151645, 151643)
    pad_token_id: int = 151643

    def decoder_weight_bytes(self, weight_dtype: str = "bf16") -> int:
        """Bytes of decoder weights read once per decode step
        (all layer Linears + biases + norms + final norm + lm_head): the ``W_dec``
        of SURVEY.md §8(d) / BASELINE.md §3.  ``fp8``: the layer Linears at one byte per weight plus one f32 scale per
        output row; biases, norms and the lm_head stay bf16 (BASELINE.json config 5)."""
        t = self.text
        linear = t.hidden_size * t.qkv_dim + t.q_dim * t.hidden_size + 3 * t.hidden_size * t.intermediate_size
        rows = t.qkv_dim + t.hidden_size + 2 * t.intermediate_size + t.hidden_size       # output rows of the four matrices
        small = t.qkv_dim + 2 * t.hidden_size                                            # qkv bias, two RMSNorm weights
        head = t.hidden_size + t.vocab_size * t.hidden_size
        if weight_dtype == "fp8":
            return t.num_layers * (linear + 4 * rows + 2 * small) + 2 * head
        return 2 * (t.num_layers * (linear + small) + head)


QWEN2_VL_2B = ModelConfig(
    name="Qwen2-VL-2B",
    vision=VisionConfig(hidden_size=1536),
    text=TextConfig(),
)

QWEN2_VL_7B = ModelConfig(
    name="Qwen2-VL-7B",
    vision=VisionConfig(hidden_size=3584),
    text=TextConfig(
        hidden_size=3584,
        intermediate_size=18944,
        num_layers=28,
        num_heads=28,
        num_kv_heads=4,
        vocab_size=152064,
        tie_word_embeddings=False,
    ),
)

# Test-sized model with the *production* head dims (ViT 80, decoder 128) so the same
# HIP kernels run; every GEMM K is a multiple of 64.  Used by tests/golden.
TINY = ModelConfig(
    name="tiny",
    vision=VisionConfig(depth=2, embed_dim=320, num_heads=4, hidden_size=256),
    text=TextConfig(
        hidden_size=256,
        intermediate_size=512,
        num_layers=2,
        num_heads=2,
        num_kv_heads=1,
        vocab_size=512,
        tie_word_embeddings=False,
    ),
    image_token_id=500,
    video_token_id=501,
    vision_start_token_id=498,
    vision_end_token_id=499,
    eos_token_ids=(497, 496),
    pad_token_id=496,
)

# TINY with a GQA group of 3 and tied embeddings (second golden model).
TINY_GQA = replace(
    TINY,
    name="tiny-gqa",
    vision=replace(TINY.vision, hidden_size=384),
    text=replace(
        TINY.text,
        hidden_size=384,
        intermediate_size=768,
        num_heads=3,
        num_kv_heads=1,
        num_layers=3,
        tie_word_embeddings=True,
    ),
)

_VISION_25 = dict(depth=32, embed_dim=1280, num_heads=16, variant="qwen2_5", intermediate_size=3420, window_size=112,
                  fullatt_block_indexes=(7, 15, 23, 31))
QWEN2_5_VL_3B = ModelConfig(
    name="Qwen2.5-VL-3B",
    vision=VisionConfig(hidden_size=2048, **_VISION_25),
    text=TextConfig(hidden_size=2048, intermediate_size=11008, num_layers=36, num_heads=16, num_kv_heads=2,
                    vocab_size=151936, tie_word_embeddings=True),
)
QWEN2_5_VL_7B = ModelConfig(
    name="Qwen2.5-VL-7B",
    vision=VisionConfig(hidden_size=3584, **_VISION_25),
    text=TextConfig(hidden_size=3584, intermediate_size=18944, num_layers=28, num_heads=28, num_kv_heads=4,
                    vocab_size=152064, tie_word_embeddings=False),
)

# Qwen2.5-VL in test size: 4 vision blocks (1 and 3 full attention, 0 and 2 windowed), 56-pixel windows (2 x 2 merged
# units = 16 patches), an MLP width that needs the zero padding (200 -> 256).
TINY_25 = replace(
    TINY,
    name="tiny-2.5",
    vision=VisionConfig(depth=4, embed_dim=320, num_heads=4, hidden_size=256, variant="qwen2_5", intermediate_size=200,
                        window_size=56, fullatt_block_indexes=(1, 3)),
)

# Test size with a decoder width the wide decode kernel takes (K % 512 == 0): the fp8-weight engine tests.
TINY_W512 = replace(
    TINY,
    name="tiny-w512",
    vision=replace(TINY.vision, hidden_size=512),
    text=replace(TINY.text, hidden_size=512, intermediate_size=1024, num_heads=4, num_kv_heads=1),
)

# Test size at the 7B decoder WIDTH (3584: the K = 3584 decode kernels; 32 rows of x do not fit the LDS, so batches above 16
# rows run the x-staging launches once per 16-row range), everything else small.
TINY_W3584 = replace(
    TINY,
    name="tiny-w3584",
    vision=replace(TINY.vision, hidden_size=3584),
    text=replace(TINY.text, hidden_size=3584, intermediate_size=1024, num_heads=4, num_kv_heads=1),
)

CONFIGS: Dict[str, ModelConfig] = {
    c.name: c for c in (QWEN2_VL_2B, QWEN2_VL_7B, QWEN2_5_VL_3B, QWEN2_5_VL_7B, TINY, TINY_GQA, TINY_25, TINY_W512, TINY_W3584)
}


def from_hf_config_dict(d: dict, name: str = "hf") -> ModelConfig:
    """Build a :class:`ModelConfig` from a Qwen2-VL ``config.json`` dict (either the
    transformers-4 flat layout or the transformers-5 ``text_config`` layout)."""
    t = d.get("text_config", d)
    v = d["vision_config"]
    rope = t.get("rope_parameters") or t.get("rope_scaling") or {}
    heads = t["num_attention_heads"]
    text = TextConfig(
        hidden_size=t["hidden_size"],
        intermediate_size=t["intermediate_size"],
        num_layers=t["num_hidden_layers"],
        num_heads=heads,
        num_kv_heads=t.get("num_key_value_heads", heads),
        head_dim=t.get("head_dim") or t["hidden_size"] // heads,
        vocab_size=t["vocab_size"],
        rms_norm_eps=t.get("rms_norm_eps", 1e-6),
        rope_theta=float(rope.get("rope_theta", t.get("rope_theta", 1e6))),
        mrope_section=tuple(rope.get("mrope_section", (16, 24, 24))),
        tie_word_embeddings=bool(d.get("tie_word_embeddings", t.get("tie_word_embeddings", False))),
    )
    common = dict(depth=v["depth"], num_heads=v["num_heads"], patch_size=v.get("patch_size", 14),
                  spatial_merge_size=v.get("spatial_merge_size", 2), temporal_patch_size=v.get("temporal_patch_size", 2),
                  in_channels=v.get("in_channels", v.get("in_chans", 3)))
    if "fullatt_block_indexes" in v or d.get("model_type") == "qwen2_5_vl":
        # Qwen2.5-VL names the tower width hidden_size and the merger output out_hidden_size
        vision = VisionConfig(embed_dim=v["hidden_size"], hidden_size=v.get("out_hidden_size", t["hidden_size"]),
                              variant="qwen2_5", intermediate_size=v["intermediate_size"],
                              window_size=v.get("window_size", 112),
                              fullatt_block_indexes=tuple(v.get("fullatt_block_indexes", (7, 15, 23, 31))), **common)
    else:
        vision = VisionConfig(embed_dim=v["embed_dim"], mlp_ratio=int(v.get("mlp_ratio", 4)), hidden_size=v["hidden_size"],
                              **common)
    eos = d.get("eos_token_id", 151645)
    eos = tuple(eos) if isinstance(eos, (list, tuple)) else (eos, 151643)
    return ModelConfig(
        name=name,
        vision=vision,
        text=text,
        image_token_id=d.get("image_token_id", 151655),
        video_token_id=d.get("video_token_id", 151656),
        vision_start_token_id=d.get("vision_start_token_id", 151652),
        vision_end_token_id=d.get("vision_end_token_id", 151653),
        eos_token_ids=eos,
        pad_token_id=int(d["pad_token_id"]) if d.get("pad_token_id") is not None else eos[-1],
    )
