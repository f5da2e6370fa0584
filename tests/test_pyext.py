"""The reference's own pybind11 module (python/cpp/*.cc -> `_ext`) built against the Level-1
library (oracle/Makefile `pyext`): SURVEY 8(f2). Run in a subprocess (tests/integration/
run_pyext.py). The CPU test proves the module and the library load and translate; the GPU test runs
`ctranslate2._ext.Translator(model, device="cuda")` = the reference's Python API on this backend."""
import glob
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
RUNNER = os.path.join(HERE, "integration", "run_pyext.py")
AREN = os.path.join(HERE, "golden", "models", "aren-transliteration-i8")
# tests/python/test_translator.py of the reference translates this fixture: "آتزمون" -> "atzmon"
AREN_SRC = [["آ", "ت", "ز", "م", "و", "ن"], ["آ", "ت", "ش", "ي", "س", "و", "ن"]]


def _built():
    return bool(glob.glob(os.path.join(ROOT, "oracle", "_ref", "pyext", "_ext*.so")))


def _run(spec):
    p = subprocess.run([sys.executable, RUNNER, json.dumps(spec)], capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


@pytest.mark.skipif(not _built(), reason="oracle/_ref/pyext is built where /root/reference exists")
def test_reference_python_extension_loads_and_translates_on_cpu():
    out = _run({"model": AREN, "device": "cpu", "compute_type": "int8", "tokens": AREN_SRC,
                "beams": [1, 4]})
    for run in out["runs"]:
        assert run["hypotheses"][0] == ["a", "t", "z", "m", "o", "n"]


@pytest.mark.gpu
@pytest.mark.skipif(not _built(), reason="oracle/_ref/pyext is built where /root/reference exists")
@pytest.mark.parametrize("compute_type", ["int8", "float32", "float16", "bfloat16"])
def test_reference_python_extension_on_this_backend(gpu, compute_type):
    """float16 / bfloat16: the reference's front-end keeps 16-bit activations end to end and every op
    goes through the 16-bit arms of the stubs (integration/half_mi355_stub.cc)."""
    model = AREN if compute_type == "int8" else AREN.replace("-i8", "")
    if compute_type in ("float16", "bfloat16"):
        spec = {"model": model, "device": "cuda", "compute_type": compute_type, "tokens": AREN_SRC,
                "beams": [1, 4]}
        dev = _run(spec)
        assert compute_type in dev["compute_types"]
        for run in dev["runs"]:
            assert run["hypotheses"][0] == ["a", "t", "z", "m", "o", "n"]
        return
    spec = {"model": model, "device": "cuda", "compute_type": compute_type, "tokens": AREN_SRC,
            "beams": [1, 4]}
    dev = _run(spec)
    cpu = _run(dict(spec, device="cpu"))
    assert dev["device_count"] >= 1
    assert compute_type in dev["compute_types"]
    for d, c in zip(dev["runs"], cpu["runs"]):
        assert d["hypotheses"] == c["hypotheses"]
        assert d["hypotheses"][0] == ["a", "t", "z", "m", "o", "n"]
        # (the reference's Device::CUDA front-end is not its Device::CPU front-end op for op - no
        # u8 compensation term, unfused attention ops - so the int8 scores agree to quantisation
        # noise, the float32 ones to rounding)
        tol = 2e-2 if compute_type == "int8" else 1e-4
        for a, b in zip(d["scores"], c["scores"]):
            assert abs(a - b) < tol


# Level 1.5, round 6: the whole of TranslationOptions reaches the fused engine
# (integration/level15_mi355_stub.cc). Each case runs through the reference's own Python API on
# Device::CUDA with the fused engine, again with CT2X_LEVEL1_FUSED=0 (the reference's op-by-op route
# on the same kernels) and on Device::CPU (the reference itself).
LEVEL15_CASES = [
    {"name": "prefix_greedy", "target_prefix": [["a", "t"], []], "beam_size": 1, "return_scores": True},
    {"name": "prefix_beam", "target_prefix": [["a", "t"], ["a"]], "beam_size": 2, "return_scores": True},
    {"name": "attention", "beam_size": 1, "return_attention": True, "return_scores": True},
    {"name": "attention_beam", "beam_size": 2, "return_attention": True, "num_hypotheses": 2},
    {"name": "replace_unknowns", "beam_size": 2, "replace_unknowns": True},
    {"name": "callback", "beam_size": 1, "callback": True, "return_scores": True},
    {"name": "num_hypotheses", "beam_size": 4, "num_hypotheses": 3, "return_scores": True},
    {"name": "alternatives", "beam_size": 1, "num_hypotheses": 3, "return_alternatives": True,
     "target_prefix": [["a"], ["a"]], "return_scores": True},
    {"name": "end_token", "beam_size": 1, "end_token": "m", "return_end_token": True},
    {"name": "bias", "beam_size": 2, "prefix_bias_beta": 0.5, "target_prefix": [["a", "t"], ["a", "t"]]},
    {"name": "logits", "beam_size": 1, "return_logits_vocab": True, "max_decoding_length": 6},
    {"name": "vmap_flag_without_a_map", "beam_size": 2, "use_vmap": True},
    {"name": "processors", "beam_size": 2, "repetition_penalty": 1.2, "no_repeat_ngram_size": 2,
     "disable_unk": True, "suppress_sequences": [["z"]], "min_decoding_length": 2},
    {"name": "sampling", "beam_size": 1, "sampling_topk": 4, "sampling_temperature": 0.8, "num_hypotheses": 2},
]


@pytest.mark.gpu
@pytest.mark.skipif(not _built(), reason="oracle/_ref/pyext is built where /root/reference exists")
def test_level15_serves_every_translation_option(gpu):
    spec = {"model": AREN, "device": "cuda", "compute_type": "int8", "tokens": AREN_SRC, "cases": LEVEL15_CASES}
    fused = _run(spec)
    assert fused["served_calls"] == len(LEVEL15_CASES), "every case must reach the fused engine"
    env_plain = dict(os.environ, CT2X_LEVEL1_FUSED="0")
    p = subprocess.run([sys.executable, RUNNER, json.dumps(spec)], capture_output=True, text=True, timeout=900,
                       env=env_plain)
    assert p.returncode == 0, p.stderr[-3000:]
    plain = json.loads(p.stdout.strip().splitlines()[-1])
    assert plain["served_calls"] == 0
    cpu = _run(dict(spec, device="cpu"))
    for f, q, c in zip(fused["runs"], plain["runs"], cpu["runs"]):
        name = f["name"]
        if name == "sampling":  # (its own random stream: shape and vocabulary only)
            assert all(len(h) == 2 for h in f["hypotheses"])
            continue
        assert f["hypotheses"] == c["hypotheses"], (name, f["hypotheses"], c["hypotheses"])
        assert f["hypotheses"] == q["hypotheses"], name
        for a, b, d in zip(f["scores"], c["scores"], q["scores"]):
            # (the reference on Device::CPU, or - where its int8 CPU arithmetic drifts from its own
            # device route by more than that - the reference's op-by-op route on Device::CUDA)
            # (alternatives: the 2nd / 3rd best first tokens of this trained int8 fixture sit at log-probabilities
            # of -3 .. -6 where the three int8 routes differ by up to 0.07; the synthetic goldens of
            # tests/test_decoding_features_gpu.py hold the engine's alternatives to 2e-5)
            tol = 1e-1 if name == "alternatives" else 2e-2
            close_cpu = len(a) == len(b) and all(abs(x - y) < tol for x, y in zip(a, b))
            close_dev = len(a) == len(d) and all(abs(x - y) < tol for x, y in zip(a, d))
            assert close_cpu or close_dev, (name, "fused", a, "cpu", b, "op-by-op", d)
        if "attention" in f:
            for b, (fa, qa, ca) in enumerate(zip(f["attention"], q["attention"], c["attention"])):
                assert [len(h) for h in fa] == [len(h) for h in ca], name
                for h, (fh, qh, ch) in enumerate(zip(fa, qa, ca)):
                    for t, (fv, qv, cv) in enumerate(zip(fh, qh, ch)):
                        # Beam search: against the reference's own Device::CUDA route. Its Device::CPU
                        # route expands the batch after the first step (src/decoding.cc:453-454) and
                        # then reads the first step's attention with row numbers of the EXPANDED batch:
                        # with this fixture sentence 0 reports sentence 1's first vector on the CPU
                        # (measured: same on the reference's CPU library alone); the device routes
                        # have no such step.
                        want = cv if name == "attention" else qv
                        assert len(fv) == len(cv) and max(abs(x - y) for x, y in zip(fv, want)) < 5e-2, \
                            (name, b, h, t, "fused", fv, "op-by-op", qv, "cpu", cv)
        if name == "callback":
            assert f["steps"] == c["steps"], (f["steps"], c["steps"])
            toks = {}
            for b, _, tok, _ in f["steps"]:
                toks.setdefault(b, []).append(tok)
            for b, hyp in enumerate(f["hypotheses"]):
                assert toks[b][:len(hyp[0])] == hyp[0]
        if name == "logits":
            assert f["logits_shapes"] == c["logits_shapes"], (f["logits_shapes"], c["logits_shapes"])


LEVEL15_GEN_CASES = [
    {"name": "greedy", "beam_size": 1, "max_length": 12, "min_length": 12, "return_scores": True},
    {"name": "beam", "beam_size": 4, "num_hypotheses": 2, "max_length": 10, "return_scores": True},
    {"name": "callback", "beam_size": 1, "max_length": 8, "callback": True},
    {"name": "end_token", "beam_size": 1, "max_length": 16, "end_token": "tok8", "return_end_token": True},
    {"name": "alternatives", "beam_size": 1, "num_hypotheses": 3, "return_alternatives": True, "max_length": 14,
     "return_scores": True},
    {"name": "sampling", "beam_size": 1, "sampling_topk": 5, "sampling_temperature": 0.9, "max_length": 8},
    # the reference's DEFAULT: the prompt decoded step by step as a returned hard prefix
    {"name": "prompt_greedy", "beam_size": 1, "max_length": 12, "include_prompt_in_result": True,
     "return_scores": True},
    {"name": "prompt_beam_ragged", "beam_size": 4, "max_length": 14, "min_length": 9, "include_prompt_in_result": True,
     "return_scores": True,
     "tokens": [["tok17", "tok5", "tok230"], ["<s>", "tok9", "tok77", "tok300", "tok12", "tok40"], ["tok9"]]},
    {"name": "prompt_callback", "beam_size": 1, "max_length": 9, "include_prompt_in_result": True, "callback": True},
    # GenerationOptions::static_prompt, with the prompt in the result and without
    {"name": "static_prompt", "beam_size": 1, "max_length": 10, "include_prompt_in_result": True,
     "return_scores": True, "static_prompt": ["tok33", "tok71", "tok250"]},
    # return_logits_vocab of a greedy search: the prompt hidden (ragged: forced steps first) and returned
    {"name": "logits_hidden", "beam_size": 1, "max_length": 6, "return_logits_vocab": True,
     "tokens": [["tok17", "tok5", "tok230"], ["tok9", "tok77", "tok300", "tok12", "tok40"]]},
    {"name": "logits_prompt", "beam_size": 1, "max_length": 8, "return_logits_vocab": True,
     "include_prompt_in_result": True,
     "tokens": [["tok17", "tok5", "tok230"], ["<s>", "tok9", "tok77", "tok300"]]},
    {"name": "static_hidden_beam", "beam_size": 4, "max_length": 8, "return_scores": True,
     "static_prompt": ["tok33", "tok71", "tok250"],
     "tokens": [["tok17", "tok5", "tok230"], ["tok9", "tok77", "tok300", "tok12", "tok40"]]},
]


@pytest.mark.gpu
@pytest.mark.skipif(not _built(), reason="oracle/_ref/pyext is built where /root/reference exists")
def test_level15_generator_options(gpu):
    """The same for ctranslate2.Generator: greedy / beam search, the step callback, a custom end
    token, return_alternatives and sampling with include_prompt_in_result=False, and the reference's
    default include_prompt_in_result=True (greedy, ragged beam search with min_length, the callback
    over the forced prompt steps) reach the fused engine; the
    sequences equal the reference's on Device::CPU and its own op-by-op route on the device."""
    import golden_cases as GC
    from ctranslate2_amd import model_builder as mb
    path = "/tmp/ct2x_test_lm_small"
    if not os.path.exists(os.path.join(path, "model.bin")):
        mb.build_decoder_lm(path, **GC.SYNTHETIC_LM["lm_small"]["model"])
    prompts = [["<s>", "tok17", "tok5", "tok230", "tok41"], ["<s>", "tok9", "tok77", "tok300", "tok12"]]
    spec = {"model": path, "device": "cuda", "compute_type": "float32", "tokens": prompts,
            "gen_cases": LEVEL15_GEN_CASES}
    fused = _run(spec)
    assert fused["served_calls"] == len(LEVEL15_GEN_CASES)
    cpu = _run(dict(spec, device="cpu"))
    for f, c in zip(fused["runs"], cpu["runs"]):
        if f["name"] == "sampling":
            assert all(len(s) == 1 and len(s[0]) == 8 for s in f["sequences"])
            continue
        assert f["sequences"] == c["sequences"], (f["name"], f["sequences"], c["sequences"])
        for a, b in zip(f["scores"], c["scores"]):
            assert len(a) == len(b) and all(abs(x - y) < 2e-3 for x, y in zip(a, b)), f["name"]
        if "callback" in f["name"]:
            assert f["steps"] == c["steps"] and len(f["steps"]) > 0
        if "logits" in f["name"]:
            import numpy as np
            assert len(f["logits"]) == len(c["logits"])
            for fa, ca in zip(f["logits"], c["logits"]):
                assert [len(h) for h in fa] == [len(h) for h in ca], f["name"]   # one vector per step, per hypothesis
                for fh, ch in zip(fa, ca):
                    np.testing.assert_allclose(np.array(fh), np.array(ch), atol=2e-4)
