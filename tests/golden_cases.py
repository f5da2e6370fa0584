"""Inputs shared by tests/golden/make_golden.py (dev time, real reference) and the parity tests
(run time, HIP backend): fixed sentences for the trained fixture and fixed seeds/shapes for the
synthetic models."""

# Arabic -> English transliteration inputs for the reference's trained fixture
# (first one is the reference's own KAT, tests/translator_test.cc:58-59).
AREN_SOURCES = [
    ["آ", "ت", "ز", "م", "و", "ن"],
    ["آ", "ت", "ش", "ي", "س", "و", "ن"],
    ["آ", "ر", "ب", "ا", "م", "ن", "ت", "و", "س"],
    ["آ", "ر", "ث", "ر"],
    ["آ", "ز", "ا"],
    ["آ", "ك", "ي", "ل", "ي", "ن"],
    ["ن", "و", "ر", "م", "ا"],
    ["م", "ا", "ر", "ي", "ن", "ا"],
]

# Synthetic TransformerSpec models (ctranslate2_amd/model_builder.py). `tiny` and `small` exist so
# the oracle comparisons finish in seconds; `tb` is BASELINE.json's Transformer-base shape.
SYNTHETIC = {
    "tiny": {
        "model": dict(num_encoder_layers=2, num_decoder_layers=2, d_model=64, ffn_dim=128,
                      num_heads=4, vocab_size=100, seed=0),
        "data_seed": 1, "src_lengths": [10, 7, 10, 3, 1, 9], "beams": [1, 2, 4],
        "max_len": 16, "min_len": 1, "logits_steps": 8,
    },
    "small": {
        "model": dict(num_encoder_layers=3, num_decoder_layers=3, d_model=256, ffn_dim=512,
                      num_heads=8, vocab_size=2000, seed=3, proj_std=0.2),
        "data_seed": 2, "src_lengths": [20, 20, 13, 5, 17, 20, 11, 8], "beams": [1, 4],
        "max_len": 24, "min_len": 24, "logits_steps": 12,
    },
    "tb": {
        "model": dict(num_encoder_layers=6, num_decoder_layers=6, d_model=512, ffn_dim=2048,
                      num_heads=8, vocab_size=32000, seed=0),
        "data_seed": 1, "src_lengths": [32] * 32, "beams": [1, 4],
        "max_len": 64, "min_len": 64, "logits_steps": 64,
    },
}

# Synthetic decoder-only (Llama-shaped: RMSNorm, RoPE non-interleaved, GQA, SwiGLU) models.
SYNTHETIC_LM = {
    "lm_tiny": {
        "model": dict(num_layers=2, d_model=64, ffn_dim=128, num_heads=4, num_heads_kv=2,
                      vocab_size=100, seed=5),
        "data_seed": 7, "batch": 4, "prompt_len": 6, "max_len": 12,
    },
    "lm_small": {
        "model": dict(num_layers=4, d_model=256, ffn_dim=512, num_heads=8, num_heads_kv=2,
                      vocab_size=2000, seed=6, w_std=0.05),
        "data_seed": 8, "batch": 6, "prompt_len": 16, "max_len": 16,
    },
}

# Decoder-only model at the Llama-3-8B geometry (BASELINE.json configs[2]): d_head 128, GQA 32 / 8,
# RMSNorm depth 4096, SwiGLU 14336 - two layers, small vocabulary (tests/golden/
# make_golden_llama3_geometry.py; the GPU test rebuilds the ~450 MB model directory from the seed).
LLAMA3_GEOMETRY = {
    "model": dict(num_layers=2, d_model=4096, ffn_dim=14336, num_heads=32, num_heads_kv=8,
                  vocab_size=4096, seed=11, w_std=0.02),
    "cases": {
        # prefill: MKL scales the finished sum (m = n = 40 < 192)
        "p40": {"data_seed": 12, "batch": 4, "prompt_len": 40, "max_len": 24, "teacher_forced": True},
        # prefill: MKL packs K, the packed copy carries the scale (m = n = 200 >= 192)
        "p200": {"data_seed": 13, "batch": 2, "prompt_len": 200, "max_len": 12, "teacher_forced": False},
    },
}

# Encoder-decoder model at the NLLB-200 3.3B geometry (BASELINE.json configs[3]): d_head 128, 16 heads,
# FFN 8192, shared embeddings, float16 weights - two layers each side (tests/golden/
# make_golden_nllb_geometry.py; the GPU test rebuilds the model directory from the seed).
NLLB_GEOMETRY = {
    "model": dict(num_encoder_layers=2, num_decoder_layers=2, d_model=2048, ffn_dim=8192, num_heads=16,
                  vocab_size=8192, seed=21, quantization="float16", direct_f16=True, share_embeddings=True),
    "data_seed": 22, "src_lengths": [24, 24, 17, 24, 9, 24], "max_len": 12, "logits_steps": 12,
}

# Decoder layer variants (TransformerDecoderLayer::operator(), src/layers/transformer.cc:196-323) and
# Su rotary scaling (RotaryEmbeddings::initialize, src/layers/attention_layer.cc:254-343): small int8
# decoder-only models, tests/golden/make_golden_layer_variants.py.
_VARIANT_BASE = dict(num_layers=3, d_model=64, ffn_dim=128, num_heads=4, num_heads_kv=2,
                     vocab_size=96, quantization="int8", w_std=0.2, rotary_base=10000.0)
_SU_LONG = [1.0 + 0.37 * i for i in range(8)]
_SU_SHORT = [1.0 + 0.05 * i for i in range(8)]
LAYER_VARIANTS = {
    # GPT-J / Phi-2 style: one norm feeds attention and feed-forward, LayerNorm with bias, GELU
    "parallel_shared": dict(_VARIANT_BASE, seed=31, layer_variant="parallel_shared", rms=False,
                            bias=True, glu=False, activation=3),
    # Falcon-7B style: input_layer_norm only
    "parallel_input": dict(_VARIANT_BASE, seed=32, layer_variant="parallel_input", rms=False,
                           bias=True, glu=False, activation=3),
    # GPT-NeoX / Falcon-40B style: a second norm for the feed-forward input
    "parallel_input_post": dict(_VARIANT_BASE, seed=33, layer_variant="parallel_input_post",
                                rms=False, bias=True, glu=False, activation=3),
    # Gemma-2 style: norms before and after both blocks, (1 + gamma) RMSNorm, GELU-tanh gate,
    # layer_scalar
    "pre_post": dict(_VARIANT_BASE, seed=34, layer_variant="pre_post", layer_scalar=0.75,
                     norm_use_residual=True, activation=1),
    # Phi-3 style Su scaling: the 2048-position table exceeds the original context -> long factors
    # and the sqrt(1 + log(scale) / log(original)) magnitude factor (scale = 128 / 32)
    "su_long": dict(_VARIANT_BASE, seed=35, rotary_su=(_SU_LONG, _SU_SHORT, 32, 128)),
    # ... within the original context -> short factors, no magnitude factor
    "su_short": dict(_VARIANT_BASE, seed=36, rotary_su=(_SU_LONG, _SU_SHORT, 4096, 4096)),
}
LAYER_VARIANT_CASE = {"data_seed": 41, "batch": 4, "prompt_len": 7, "max_len": 10}
