"""Writes CTranslate2 model directories (model.bin + config.json + vocabulary) for SYNTHETIC
models: random weights of a named architecture, used by bench.py and the tests because no
network/checkpoints are available. The wire format is the reference's
(python/ctranslate2/specs/model_spec.py:382-415, binary version 6), the variable names are the
reference spec's (python/ctranslate2/specs/transformer_spec.py), and weight quantisation follows
model_spec.py:223-245 -- so the reference CPU library loads these directories unchanged (that is
how the golden outputs under tests/golden/ were produced, see tests/golden/make_golden.py).

numpy only; never touches the GPU.
"""
import json
import os
import struct

import numpy as np

BINARY_VERSION = 6
_TYPE_IDS = {"float32": 0, "int8": 1, "int16": 2, "int32": 3, "float16": 4, "bfloat16": 5}


class BF16(np.ndarray):
    """uint16 array holding bfloat16 bit patterns (numpy has no bfloat16); written with type id 5."""


def to_bf16(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16).view(BF16)

# ops::ActivationType (include/ctranslate2/ops/activation.h:9-17)
RELU, GELU_TANH, SWISH, GELU, GELU_SIGMOID, TANH, SIGMOID = range(7)


def _write_string(f, s):
    data = s.encode("utf-8")
    f.write(struct.pack("H", len(data) + 1))
    f.write(data)
    f.write(struct.pack("B", 0))


def write_model_bin(path, spec_name, revision, variables, aliases=()):
    """variables: dict name -> np.ndarray (0-d arrays for attributes); written in sorted order
    like ModelSpec.variables(ordered=True)."""
    names = sorted(variables)
    with open(path, "wb") as f:
        f.write(struct.pack("I", BINARY_VERSION))
        _write_string(f, spec_name)
        f.write(struct.pack("I", revision))
        f.write(struct.pack("I", len(names)))
        for name in names:
            value = np.asarray(variables[name])
            if value.ndim > 0:  # ascontiguousarray would promote 0-d attributes to 1-d
                value = np.ascontiguousarray(value)
            _write_string(f, name)
            f.write(struct.pack("B", value.ndim))
            for dim in value.shape:
                f.write(struct.pack("I", dim))
            type_name = "bfloat16" if isinstance(variables[name], BF16) else str(value.dtype)
            f.write(struct.pack("B", _TYPE_IDS[type_name]))
            f.write(struct.pack("I", value.nbytes))
            f.write(value.tobytes())
        aliases = sorted(aliases)
        f.write(struct.pack("I", len(aliases)))
        for alias, target in aliases:
            _write_string(f, alias)
            _write_string(f, target)


def quantize_int8(weight):
    """model_spec.py:233-240: per-row scale = 127/absmax (absmax 0 -> 127), rint, int8."""
    value = weight.astype(np.float32, copy=True)
    amax = np.amax(np.absolute(value), axis=1)
    amax[amax == 0] = 127.0
    scale = (127.0 / amax).astype(np.float32)
    value *= np.expand_dims(scale, 1)
    value = np.rint(value).astype(np.int8)
    return value, scale


# AutoAWQ order_map of the GEMM layout: nibble i of a packed word holds output channel
# _AWQ_ORDER[i] of its group of eight (python/ctranslate2/converters/transformers.py keeps the
# checkpoint's qweight / qzeros / scales as they are).
_AWQ_ORDER = (0, 2, 4, 6, 1, 3, 5, 7)


def awq_quantize(weight, group):
    """weight [out_c, in_c] float -> (q, z, s): 4-bit asymmetric group quantisation along in_c as
    AutoAWQ's pseudo_quantize_tensor does it (zero point + scale per (output channel, group)):
    q, z in 0..15 and s float16, with w ~ (q - z) * s."""
    out_c, in_c = weight.shape
    w = weight.reshape(out_c, in_c // group, group).astype(np.float32)
    hi, lo = w.max(axis=2), w.min(axis=2)
    s = np.maximum((hi - lo) / 15.0, 1e-5).astype(np.float16)
    sf = s.astype(np.float32)
    z = np.clip(np.rint(-lo / sf), 0, 15).astype(np.int32)
    q = np.clip(np.rint(w / sf[:, :, None]) + z[:, :, None], 0, 15).astype(np.int32)
    return q.reshape(out_c, in_c), z, s


def awq_dequantize(q, z, s, group):
    """(q - z) * s rounded to float16 once: the weight the AWQ kernels multiply with, [out_c, in_c]."""
    zz = np.repeat(z, group, axis=1)
    ss = np.repeat(s.astype(np.float32), group, axis=1)
    return ((q - zz).astype(np.float32) * ss).astype(np.float16)


def _bf16_round(a):
    """float32 array -> the nearest bfloat16 values (round to nearest even), as float32"""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
    return u.astype(np.uint32).view(np.float32)


def awq_dequantize_bf16(q, z, s, group):
    """The weight the bfloat16 arm of the AWQ kernels multiplies with (awq.hip, round 6): the float16
    scale rounded to bfloat16 once, (q - z) * s rounded to bfloat16 once; float32 [out_c, in_c]."""
    zz = np.repeat(z, group, axis=1)
    ss = np.repeat(_bf16_round(s.astype(np.float32)), group, axis=1)
    return _bf16_round((q - zz).astype(np.float32) * ss)


def _pack_nibbles(m, order):
    r, c = m.shape
    blocks = m.reshape(r, c // 8, 8).astype(np.uint32)
    out = np.zeros((r, c // 8), np.uint32)
    for i in range(8):
        out |= blocks[:, :, order[i]] << np.uint32(4 * i)
    return out.view(np.int32)


def awq_pack(q, z, s, group, layout):
    """-> (weight, weight_scale, weight_zero) variables of a Dense in the AWQ GEMM layout (weight
    [in_c, out_c / 8], scales [in_c / G, out_c], zeros [in_c / G, out_c / 8]) or GEMV layout (weight
    [out_c, in_c / 8], scales [out_c, sf_w], zeros [out_c, zeros_w], widths padded as
    src/ops/awq/gemv_gpu.cu:296-302 / 365-369 expect)."""
    out_c, in_c = q.shape
    if layout == "gemm":
        return (_pack_nibbles(q.T.copy(), _AWQ_ORDER), np.ascontiguousarray(s.T),
                _pack_nibbles(z.T.copy(), _AWQ_ORDER))
    md = lambda c, d: (c + d - 1) // d
    groups = in_c // group
    zeros_w = md(md(groups, 8), 2) * 2 if group == 64 else md(groups, 8)
    zp = np.zeros((out_c, zeros_w * 8), np.int32)
    zp[:, :groups] = z
    sc = np.zeros((out_c, zeros_w * 8), np.float16)
    sc[:, :groups] = s
    seq = tuple(range(8))
    return _pack_nibbles(q, seq), sc, _pack_nibbles(zp, seq)


class _Builder:
    def __init__(self, seed, quantization, w_std=0.02, b_std=0.01, direct_int8=False, direct_f16=False):
        self.rng = np.random.default_rng(seed)
        self.quantization = quantization
        self.w_std, self.b_std = w_std, b_std
        # direct_int8: draw the int8 weights themselves (uniform in [-127, 127]) and per-row scales
        # 127/absmax with absmax ~ w_std*sqrt(3) (same variance as N(0, w_std^2)) instead of
        # quantising a float matrix: multi-GB benchmark models in seconds, not minutes.
        self.direct_int8 = direct_int8 and quantization == "int8"
        # direct_f16: the float16 weights are uniform multiples of absmax / 127 (same variance as
        # N(0, w_std^2)), drawn as int8 and scaled in float16 - a 3.3 B parameter model in about a
        # minute instead of ten
        self.direct_f16 = direct_f16 and quantization in ("float16", "bfloat16")
        self.vars = {}
        # awq = (layout, group, twin): layer linears are stored as AWQ int4 weights; with twin=True the
        # builder instead stores the float16 weight those int4 values dequantise to (the comparison
        # model of tests/test_awq_gpu.py)
        self.awq = None

    def attr(self, name, value, dtype):
        self.vars[name] = np.array(value, dtype=dtype)

    def random_weight(self, name, n, k):
        if self.direct_int8:
            q = self.rng.integers(-127, 128, (n, k), dtype=np.int8)
            amax = self.w_std * np.sqrt(3.0) * (1.0 + 0.1 * self.rng.random(n))
            self.vars[name] = q
            self.vars[name + "_scale"] = (127.0 / amax).astype(np.float32)
        elif self.direct_f16:
            q = self.rng.integers(-127, 128, (n, k), dtype=np.int8)
            if self.quantization == "bfloat16":
                w = q.astype(np.float32)
                w *= np.float32(self.w_std * np.sqrt(3.0) / 127.0)
                self.vars[name] = to_bf16(w)
            else:
                w = q.astype(np.float16)
                w *= np.float16(self.w_std * np.sqrt(3.0) / 127.0)
                self.vars[name] = w
        else:
            self.weight(name, self.rng.normal(0, self.w_std, (n, k)).astype(np.float32))

    def linear(self, scope, n, k, bias=True, awq_ok=False):
        if self.awq is not None and awq_ok:
            layout, group, twin = self.awq
            w = self.rng.normal(0, self.w_std, (n, k)).astype(np.float32)
            q, z, s = awq_quantize(w, group)
            if twin == "bf16":  # the bfloat16 arm's weights, stored exactly (float32)
                self.vars[scope + "/weight"] = awq_dequantize_bf16(q, z, s, group)
            elif twin:
                self.vars[scope + "/weight"] = awq_dequantize(q, z, s, group)
            else:
                qw, sc, qz = awq_pack(q, z, s, group, layout)
                self.vars[scope + "/weight"] = qw
                self.vars[scope + "/weight_scale"] = sc
                self.vars[scope + "/weight_zero"] = qz
        else:
            self.random_weight(scope + "/weight", n, k)
        if bias:
            self.vars[scope + "/bias"] = self.rng.normal(0, self.b_std, (n,)).astype(np.float32)

    def weight(self, name, w):
        if self.quantization == "int8":
            q, s = quantize_int8(w)
            self.vars[name] = q
            self.vars[name + "_scale"] = s
        elif self.quantization == "float16":
            self.vars[name] = w.astype(np.float16)
        elif self.quantization == "bfloat16":
            self.vars[name] = to_bf16(w)
        else:
            self.vars[name] = w.astype(np.float32)

    def layer_norm(self, scope, d, rms=False):
        # gamma ~ 1, beta ~ 0 with a little noise so every norm is a distinct variable
        self.vars[scope + "/gamma"] = (1.0 + self.rng.normal(0, 0.02, (d,))).astype(np.float32)
        if not rms:
            self.vars[scope + "/beta"] = self.rng.normal(0, 0.02, (d,)).astype(np.float32)


def _vocab_tokens(vocab_size):
    return ["<blank>", "<s>", "</s>", "<unk>"] + ["tok%d" % i for i in range(4, vocab_size)]


def build_transformer(output_dir, num_encoder_layers=6, num_decoder_layers=6, d_model=512,
                      ffn_dim=2048, num_heads=8, vocab_size=32000, seed=0, quantization="int8",
                      activation=RELU, pre_norm=True, w_std=0.02, proj_std=None, direct_f16=False,
                      share_embeddings=False):
    """TransformerSpec.from_config((enc, dec), heads) with sinusoidal positions, shared
    vocabulary -- the Transformer-base En->De shape of BASELINE.json when left at the defaults."""
    os.makedirs(output_dir, exist_ok=True)
    b = _Builder(seed, quantization, w_std=w_std, direct_f16=direct_f16)
    d = d_model
    aliases = []

    def mha(scope, self_attention):
        b.layer_norm(scope + "/layer_norm", d)
        if self_attention:
            b.linear(scope + "/linear_0", 3 * d, d)
            b.linear(scope + "/linear_1", d, d)
        else:
            b.linear(scope + "/linear_0", d, d)
            b.linear(scope + "/linear_1", 2 * d, d)
            b.linear(scope + "/linear_2", d, d)

    def ffn(scope):
        b.layer_norm(scope + "/layer_norm", d)
        b.linear(scope + "/linear_0", ffn_dim, d)
        b.linear(scope + "/linear_1", d, ffn_dim)

    emb_std = w_std
    if share_embeddings:
        # one table for source embeddings, target embeddings and the output projection (NLLB,
        # transformer_spec.py share_embeddings / the converters' aliases): stored once, aliased
        b.random_weight("decoder/embeddings/weight", vocab_size, d)
        aliases.append(("encoder/embeddings_0/weight", "decoder/embeddings/weight"))
        aliases.append(("decoder/projection/weight", "decoder/embeddings/weight"))
        for suffix in ("_scale",):
            if "decoder/embeddings/weight" + suffix in b.vars:
                aliases.append(("encoder/embeddings_0/weight" + suffix, "decoder/embeddings/weight" + suffix))
                aliases.append(("decoder/projection/weight" + suffix, "decoder/embeddings/weight" + suffix))
    elif direct_f16:
        b.random_weight("encoder/embeddings_0/weight", vocab_size, d)
        b.random_weight("decoder/embeddings/weight", vocab_size, d)
    else:
        b.weight("encoder/embeddings_0/weight", b.rng.normal(0, emb_std, (vocab_size, d)).astype(np.float32))
        b.weight("decoder/embeddings/weight", b.rng.normal(0, emb_std, (vocab_size, d)).astype(np.float32))
    for l in range(num_encoder_layers):
        mha("encoder/layer_%d/self_attention" % l, True)
        ffn("encoder/layer_%d/ffn" % l)
    for l in range(num_decoder_layers):
        mha("decoder/layer_%d/self_attention" % l, True)
        mha("decoder/layer_%d/attention" % l, False)
        ffn("decoder/layer_%d/ffn" % l)
    if pre_norm:
        b.layer_norm("encoder/layer_norm", d)
        b.layer_norm("decoder/layer_norm", d)
    if proj_std is not None:
        b.w_std = proj_std
    if share_embeddings:
        b.vars["decoder/projection/bias"] = b.rng.normal(0, b.b_std, (vocab_size,)).astype(np.float32)
    else:
        b.linear("decoder/projection", vocab_size, d)

    for scope in ("encoder", "decoder"):
        b.attr(scope + "/num_heads", num_heads, np.int16)
        b.attr(scope + "/pre_norm", int(pre_norm), np.int8)
        b.attr(scope + "/activation", activation, np.int8)
        b.attr(scope + "/scale_embeddings", 1, np.int8)
    b.attr("encoder/embeddings_merge", 0, np.int8)
    b.attr("encoder/multi_query_attention", 0, np.int8)
    b.attr("decoder/alibi", 0, np.int8)
    b.attr("decoder/alibi_use_positive_positions", 0, np.int8)
    b.attr("decoder/scale_alibi", 0, np.int8)
    b.attr("decoder/alignment_heads", 1, np.int16)
    b.attr("decoder/alignment_layer", -1, np.int16)
    b.attr("decoder/start_from_zero_embedding", 0, np.int8)

    write_model_bin(os.path.join(output_dir, "model.bin"), "TransformerSpec", 7, b.vars, aliases)
    config = {
        "add_source_bos": False,
        "add_source_eos": False,
        "bos_token": "<s>",
        "decoder_start_token": "<s>",
        "eos_token": "</s>",
        "layer_norm_epsilon": None,
        "multi_query_attention": False,
        "unk_token": "<unk>",
    }
    with open(os.path.join(output_dir, "config.json"), "w") as f:
        json.dump(config, f, indent=2, sort_keys=True)
    with open(os.path.join(output_dir, "shared_vocabulary.json"), "w", encoding="utf-8") as f:
        json.dump(_vocab_tokens(vocab_size), f, indent=2)
    return output_dir


def build_decoder_lm(output_dir, num_layers=32, d_model=4096, ffn_dim=14336, num_heads=32,
                     num_heads_kv=8, vocab_size=128256, seed=0, quantization="int8",
                     rotary_base=500000.0, w_std=0.02, direct_int8=False, awq=None,
                     layer_variant=None, layer_scalar=None, rotary_su=None, glu=True, bias=False,
                     rms=True, activation=SWISH, norm_use_residual=False, rotary_interleave=False):
    """TransformerDecoderModelSpec.from_config(layers, heads, activation=SWISH, ffn_glu=True,
    rms_norm=True, rotary_dim=0, rotary_interleave=False, num_heads_kv=...) -- the Llama-3 shape
    (python/ctranslate2/converters/transformers.py:1736-1752) when left at the defaults.

    layer_variant (TransformerDecoderLayerSpec, python/ctranslate2/specs/transformer_spec.py):
      "parallel_shared"      shared_layer_norm (GPT-J / Phi: parallel_residual + shared_layer_norm)
      "parallel_input"       input_layer_norm only (parallel_residual, Falcon-7B style)
      "parallel_input_post"  input_layer_norm + post_attention_layer_norm (GPT-NeoX / Falcon-40B)
      "pre_post"             input / post_attention / pre_feedforward / post_feedforward norms
                             (pre_post_layer_norm, Gemma 2 / 3); layer_scalar multiplies the output
    rotary_su = (long_factors, short_factors, original_max_position_embeddings,
    max_position_embeddings): Su scaling (Phi-3, RotaryScalingType.Su = 1)."""
    os.makedirs(output_dir, exist_ok=True)
    b = _Builder(seed, quantization, w_std=w_std, direct_int8=direct_int8)
    b.awq = awq  # None, or (layout "gemm" | "gemv", group size, twin)
    d = d_model
    dh = d // num_heads

    def norm(scope):
        b.layer_norm(scope, d, rms=rms)
        if norm_use_residual:
            b.attr(scope + "/layer_norm_use_residual", 1, np.int8)

    b.random_weight("decoder/embeddings/weight", vocab_size, d)
    for l in range(num_layers):
        s = "decoder/layer_%d" % l
        if layer_variant is None:
            norm(s + "/self_attention/layer_norm")
        elif layer_variant == "parallel_shared":
            norm(s + "/shared_layer_norm")
        elif layer_variant == "parallel_input":
            norm(s + "/input_layer_norm")
        elif layer_variant == "parallel_input_post":
            norm(s + "/input_layer_norm")
            norm(s + "/post_attention_layer_norm")
        elif layer_variant == "pre_post":
            for name in ("input_layer_norm", "post_attention_layer_norm",
                         "pre_feedforward_layer_norm", "post_feedforward_layer_norm"):
                norm(s + "/" + name)
            if layer_scalar is not None:
                b.attr(s + "/layer_scalar", layer_scalar, np.float32)
        else:
            raise ValueError("unknown layer_variant %r" % (layer_variant,))
        b.linear(s + "/self_attention/linear_0", (num_heads + 2 * num_heads_kv) * dh, d, bias=bias, awq_ok=True)
        b.linear(s + "/self_attention/linear_1", d, d, bias=bias, awq_ok=True)
        b.attr(s + "/self_attention/rotary_dim", 0, np.int32)
        b.attr(s + "/self_attention/rotary_interleave", int(rotary_interleave), np.int8)
        b.attr(s + "/self_attention/rotary_base", rotary_base, np.float32)
        b.attr(s + "/self_attention/num_heads_kv", num_heads_kv, np.int32)
        if rotary_su is not None:
            long_f, short_f, orig_max, max_pos = rotary_su
            b.attr(s + "/self_attention/rotary_scaling_type", 1, np.int8)
            b.vars[s + "/self_attention/rotary_scaling_long_factor"] = np.asarray(long_f, np.float32)
            b.vars[s + "/self_attention/rotary_scaling_short_factor"] = np.asarray(short_f, np.float32)
            b.attr(s + "/self_attention/original_max_position_embeddings", orig_max, np.int32)
            b.attr(s + "/self_attention/max_position_embeddings", max_pos, np.int32)
        if layer_variant is None:  # (drawn here: the seeds of the committed goldens depend on the order)
            norm(s + "/ffn/layer_norm")
        b.linear(s + "/ffn/linear_0", ffn_dim, d, bias=bias, awq_ok=True)
        if glu:
            b.linear(s + "/ffn/linear_0_noact", ffn_dim, d, bias=bias, awq_ok=True)
        b.linear(s + "/ffn/linear_1", d, ffn_dim, bias=bias, awq_ok=True)
    norm("decoder/layer_norm")
    b.linear("decoder/projection", vocab_size, d, bias=False)
    b.attr("decoder/num_heads", num_heads, np.int16)
    b.attr("decoder/pre_norm", 1, np.int8)
    b.attr("decoder/activation", activation, np.int8)
    b.attr("decoder/scale_embeddings", 0, np.int8)
    b.attr("decoder/alibi", 0, np.int8)
    b.attr("decoder/alibi_use_positive_positions", 0, np.int8)
    b.attr("decoder/scale_alibi", 0, np.int8)
    b.attr("decoder/start_from_zero_embedding", 0, np.int8)
    write_model_bin(os.path.join(output_dir, "model.bin"), "TransformerDecoderSpec", 8, b.vars)
    config = {
        "bos_token": "<s>",
        "eos_token": "</s>",
        "layer_norm_epsilon": 1e-05,
        "multi_query_attention": True,
        "quantization_bits": None,
        "quantization_group_size": None,
        "quantization_type": 0,
        "unk_token": "<unk>",
    }
    if awq is not None and not awq[2]:
        # common_spec.Quantization: CT2 = 0, AWQ_GEMM = 1, AWQ_GEMV = 2
        config["quantization_type"] = 1 if awq[0] == "gemm" else 2
        config["quantization_bits"] = 4
        config["quantization_group_size"] = awq[1]
    with open(os.path.join(output_dir, "config.json"), "w") as f:
        json.dump(config, f, indent=2, sort_keys=True)
    with open(os.path.join(output_dir, "vocabulary.json"), "w", encoding="utf-8") as f:
        json.dump(_vocab_tokens(vocab_size), f, indent=2)
    return output_dir


def whisper_vocab_tokens(num_text_tokens, languages, num_timestamps):
    """The Whisper vocabulary layout WhisperReplica relies on (src/models/whisper.cc:62-77):
    text tokens..., <|endoftext|>, <|startoftranscript|>, language tokens..., <|translate|>,
    <|transcribe|>, <|startoflm|>, <|startofprev|>, <|nospeech|>, <|notimestamps|>, time tokens.
    Token 0 is the empty string: this copy of the reference calls a model multilingual when
    Vocabulary::to_id("") is not the unknown token (whisper.cc:72)."""
    tokens = [""] + ["tx%d" % i for i in range(1, num_text_tokens)]
    tokens += ["<|endoftext|>", "<|startoftranscript|>"]
    tokens += ["<|%s|>" % lang for lang in languages]
    tokens += ["<|translate|>", "<|transcribe|>", "<|startoflm|>", "<|startofprev|>",
               "<|nospeech|>", "<|notimestamps|>"]
    tokens += ["<|%.2f|>" % (0.02 * i) for i in range(num_timestamps)]
    return tokens


def build_whisper(output_dir, num_encoder_layers=2, num_decoder_layers=2, d_model=64, ffn_dim=256,
                  num_heads=4, n_mels=8, max_source_positions=12, max_target_positions=32,
                  num_text_tokens=40, languages=("en", "fr", "de", "es"), num_timestamps=21,
                  seed=0, quantization="float32", w_std=0.08, alignment_heads=((1, 0), (1, 2), (0, 3)),
                  direct_f16=False, awq=None):
    """WhisperSpec (python/ctranslate2/specs/whisper_spec.py): encoder = two Conv1D (kernel 3,
    padding 1, stride 1 / 2, GELU) + learned positions + pre-norm GELU encoder layers + layer_norm;
    decoder = TransformerDecoderSpec(activation=GELU) with learned positions, scale_embeddings
    False and the output projection aliased to the embeddings
    (python/ctranslate2/converters/transformers.py WhisperLoader). Whisper-large-v3 shape:
    32 + 32 layers, d_model 1280, 20 heads, ffn 5120, n_mels 128, 1500 / 448 positions,
    51866 tokens.

    awq = (layout "gemm" | "gemv", group size, twin): the DECODER layers' Dense weights are stored as
    AWQ int4 (BASELINE configs[4]: "bf16 + AWQ int4 weights"; the reference forces float16 for
    such models) - or, with twin=True, as the float16 values those int4 weights dequantise to."""
    os.makedirs(output_dir, exist_ok=True)
    b = _Builder(seed, quantization, w_std=w_std, direct_f16=direct_f16)
    d = d_model
    awq_layers = [False]  # set while the decoder layers are drawn

    def conv(scope, out_c, in_c):
        if b.direct_f16:
            b.random_weight(scope + "/weight", out_c, in_c * 3)
            w16 = b.vars[scope + "/weight"]
            b.vars[scope + "/weight"] = w16.reshape(out_c, in_c, 3).view(type(w16))
            b.vars[scope + "/bias"] = b.rng.normal(0, b.b_std, (out_c,)).astype(np.float32)
            return
        w = b.rng.normal(0, w_std, (out_c, in_c, 3)).astype(np.float32)
        if quantization == "int8":
            # model_spec.py:229-243: scale per output channel over the [in, kernel] plane
            q, s = quantize_int8(w.reshape(out_c, -1))
            b.vars[scope + "/weight"] = q.reshape(w.shape)
            b.vars[scope + "/weight_scale"] = s
        elif quantization == "float16":
            b.vars[scope + "/weight"] = w.astype(np.float16)
        elif quantization == "bfloat16":
            b.vars[scope + "/weight"] = to_bf16(w)
        else:
            b.vars[scope + "/weight"] = w
        b.vars[scope + "/bias"] = b.rng.normal(0, b.b_std, (out_c,)).astype(np.float32)

    def mha(scope, self_attention):
        b.layer_norm(scope + "/layer_norm", d)
        if self_attention:
            b.linear(scope + "/linear_0", 3 * d, d, awq_ok=awq_layers[0])
            b.linear(scope + "/linear_1", d, d, awq_ok=awq_layers[0])
        else:
            b.linear(scope + "/linear_0", d, d, awq_ok=awq_layers[0])
            b.linear(scope + "/linear_1", 2 * d, d, awq_ok=awq_layers[0])
            b.linear(scope + "/linear_2", d, d, awq_ok=awq_layers[0])

    def ffn(scope):
        b.layer_norm(scope + "/layer_norm", d)
        b.linear(scope + "/linear_0", ffn_dim, d, awq_ok=awq_layers[0])
        b.linear(scope + "/linear_1", d, ffn_dim, awq_ok=awq_layers[0])

    def positions(scope, n):
        t = b.rng.normal(0, w_std, (n, d)).astype(np.float32)
        if quantization == "float16":
            t = t.astype(np.float16)
        elif quantization == "bfloat16":
            t = to_bf16(t)
        b.vars[scope + "/encodings"] = t

    conv("encoder/conv1", d, n_mels)
    conv("encoder/conv2", d, d)
    positions("encoder/position_encodings", max_source_positions)
    for l in range(num_encoder_layers):
        mha("encoder/layer_%d/self_attention" % l, True)
        ffn("encoder/layer_%d/ffn" % l)
    b.layer_norm("encoder/layer_norm", d)
    b.attr("encoder/num_heads", num_heads, np.int16)

    tokens = whisper_vocab_tokens(num_text_tokens, languages, num_timestamps)
    vocab_size = len(tokens)
    if direct_f16:
        b.random_weight("decoder/embeddings/weight", vocab_size, d)
    else:
        b.weight("decoder/embeddings/weight", b.rng.normal(0, w_std, (vocab_size, d)).astype(np.float32))
    aliases = [("decoder/projection/weight", "decoder/embeddings/weight")]
    if "decoder/embeddings/weight_scale" in b.vars:
        aliases.append(("decoder/projection/weight_scale", "decoder/embeddings/weight_scale"))
    positions("decoder/position_encodings", max_target_positions)
    b.awq = awq
    awq_layers[0] = awq is not None
    for l in range(num_decoder_layers):
        mha("decoder/layer_%d/self_attention" % l, True)
        mha("decoder/layer_%d/attention" % l, False)
        ffn("decoder/layer_%d/ffn" % l)
    awq_layers[0] = False
    b.layer_norm("decoder/layer_norm", d)
    b.attr("decoder/num_heads", num_heads, np.int16)
    b.attr("decoder/pre_norm", 1, np.int8)
    b.attr("decoder/activation", GELU, np.int8)
    b.attr("decoder/scale_embeddings", 0, np.int8)
    b.attr("decoder/alibi", 0, np.int8)
    b.attr("decoder/alibi_use_positive_positions", 0, np.int8)
    b.attr("decoder/scale_alibi", 0, np.int8)
    b.attr("decoder/alignment_heads", 1, np.int16)
    b.attr("decoder/alignment_layer", -1, np.int16)
    b.attr("decoder/start_from_zero_embedding", 0, np.int8)
    write_model_bin(os.path.join(output_dir, "model.bin"), "WhisperSpec", 3, b.vars, aliases)

    eot = num_text_tokens
    sot = eot + 1
    first_lang = sot + 1
    no_timestamps = first_lang + len(languages) + 5
    config = {
        # a few text tokens, the special tokens below <|notimestamps|> except the languages' tasks
        "suppress_ids": [1, 2, 7, sot, first_lang + len(languages), first_lang + len(languages) + 1,
                         first_lang + len(languages) + 2, first_lang + len(languages) + 3,
                         first_lang + len(languages) + 4],
        "suppress_ids_begin": [3, eot],
        "lang_ids": list(range(first_lang, first_lang + len(languages))),
        "alignment_heads": [list(p) for p in alignment_heads],
    }
    assert no_timestamps + 1 + num_timestamps == vocab_size
    if awq is not None and not awq[2]:
        # common_spec.Quantization: CT2 = 0, AWQ_GEMM = 1, AWQ_GEMV = 2
        config["quantization_type"] = 1 if awq[0] == "gemm" else 2
        config["quantization_bits"] = 4
        config["quantization_group_size"] = awq[1]
    with open(os.path.join(output_dir, "config.json"), "w") as f:
        json.dump(config, f, indent=2, sort_keys=True)
    with open(os.path.join(output_dir, "vocabulary.json"), "w", encoding="utf-8") as f:
        json.dump(tokens, f, indent=2)
    return output_dir
