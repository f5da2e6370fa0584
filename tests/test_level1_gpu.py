"""-m gpu: the Level-1 boundary EXECUTED. integration/libctranslate2_mi355.so is the reference's
OWN front-end - ctranslate2::Translator, ReplicaPool, models::Model, layers::Dense /
MultiHeadAttention / TransformerDecoder, decoding.cc, every ops::X front-end - compiled from the
sources where they lie with -DCT2_WITH_CUDA by the plain host compiler (oracle/Makefile target
mi355, integration/shim_cuda/ for the CUDA runtime names its headers leak) and linked with
integration/{ops,host}_mi355_stub.cc + libct2_mi355.so in place of src/cuda/* and
src/ops/*_gpu.cu. Here it runs Translator(model, Device::CUDA, ComputeType::INT8) op by op on the
MI355X kernels and must reproduce what the same front-end computes on Device::CPU
(tests/golden/model_golden.json; include/ctranslate2/translator.h:26-60,
tests/translator_test.cc:53-97)."""
import json
import os
import subprocess
import sys

import pytest

import golden_cases as GC

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(ROOT, "oracle", "_ref", "libct2ref_capi_mi355.so")


def run_level1(model_dir, compute_type, sources, runs, fused=None):
    if not os.path.exists(LIB):
        pytest.skip("Level-1 integration library not built (make -C oracle mi355)")
    spec = json.dumps({"prompts": sources["prompts"], "runs": runs} if isinstance(sources, dict)
                      else {"sources": sources, "runs": runs})
    env = dict(os.environ)
    if fused is not None:
        env["CT2X_LEVEL1_FUSED"] = "1" if fused else "0"
    p = subprocess.run([sys.executable, os.path.join(HERE, "integration", "run_level1.py"), model_dir,
                        compute_type, spec], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "model_golden.json")) as f:
        return json.load(f)


def test_reference_translator_on_device_cuda_transliteration(gpu, golden):
    """The reference's own KAT (tests/translator_test.cc:53-97) through the reference's own
    Translator on Device::CUDA = this backend: "a t z m o n", and every golden sentence."""
    g = golden["aren-transliteration-i8"]
    out = run_level1(os.path.join(HERE, "golden", "models", "aren-transliteration-i8"), "int8",
                     g["sources"], [{"beam": 1, "max_len": g["max_len"], "min_len": g["min_len"]},
                                    {"beam": 4, "max_len": g["max_len"], "min_len": g["min_len"]}])
    assert out["compute_type"] == "int8_float32"
    vocab = [l.rstrip("\n") for l in open(os.path.join(HERE, "golden", "models",
                                                        "aren-transliteration-i8",
                                                        "target_vocabulary.txt"), encoding="utf-8")]
    first = [vocab[i] for i in out["runs"][0]["hypotheses"][0]]
    assert first == ["a", "t", "z", "m", "o", "n"], first
    for run, key in zip(out["runs"], ("beam1", "beam4")):
        want = [h[0] for h in g["runs"][key]["hypotheses"]]
        same = sum(int(a == b) for a, b in zip(run["hypotheses"], want))
        print("Level-1 aren-i8 %s: %d/%d sentences identical to Device::CPU" % (key, same, len(want)))
        assert run["hypotheses"] == want


@pytest.mark.parametrize("key", ["tiny", "small", "tb"])
def test_reference_translator_on_device_cuda_synthetic(gpu, golden, key):
    from ctranslate2_amd import model_builder as mb
    g = golden[key]
    path = "/tmp/ct2x_test_%s" % key
    if not os.path.exists(os.path.join(path, "model.bin")):
        mb.build_transformer(path, **GC.SYNTHETIC[key]["model"])
    out = run_level1(path, "int8", g["sources"],
                     [{"beam": 1, "max_len": g["max_len"], "min_len": g["min_len"]}])
    want = [h[0] for h in g["runs"]["beam1"]["hypotheses"]]
    got = out["runs"][0]["hypotheses"]
    same = sum(int(a == b) for a, b in zip(got, want))
    print("Level-1 %s greedy: %d/%d sentences token-identical to the reference on Device::CPU"
          % (key, same, len(want)))
    # the position table is made by the host's MKL VML here too; everything else is bit-exact
    assert same >= len(want) - max(1, len(want) // 16)


def test_level15_reference_translator_on_the_fused_engine(gpu, golden):
    """Level 1.5 (integration/level15_mi355_stub.cc): the reference's own
    ctranslate2::Translator(model, Device::CUDA) with EncoderDecoderReplica::run_translation served
    by the fused engine - API identical, the model handed over from the reference's memory
    (ct2x_translator_create_from_memory). Same tokens as the op-by-op route (CT2X_LEVEL1_FUSED=0)
    and as the reference on Device::CPU, greedy and beam search; the served-call counter shows which
    route ran."""
    from ctranslate2_amd import model_builder as mb
    for key, path in (("aren-transliteration-i8", os.path.join(HERE, "golden", "models", "aren-transliteration-i8")),
                      ("tb", "/tmp/ct2x_test_tb")):
        g = golden[key]
        if key == "tb" and not os.path.exists(os.path.join(path, "model.bin")):
            mb.build_transformer(path, **GC.SYNTHETIC[key]["model"])
        runs = [{"beam": 1, "max_len": g["max_len"], "min_len": g["min_len"]},
                {"beam": 4, "max_len": g["max_len"], "min_len": g["min_len"]}]
        fused = run_level1(path, "int8", g["sources"], runs, fused=True)
        plain = run_level1(path, "int8", g["sources"], runs, fused=False)
        assert fused["fused_calls"] == len(runs) and plain["fused_calls"] == 0
        for a, b, name in zip(fused["runs"], plain["runs"], ("beam1", "beam4")):
            want = [h[0] for h in g["runs"][name]["hypotheses"]]
            same = sum(int(x == y) for x, y in zip(a["hypotheses"], want))
            print("Level 1.5 %s %s: %d/%d sentences identical to Device::CPU" % (key, name, same, len(want)))
            assert a["hypotheses"] == want
            # (the op-by-op route builds the position table and its softmax through generic ops: the
            # bound of test_reference_translator_on_device_cuda_synthetic)
            same_b = sum(int(x == y) for x, y in zip(b["hypotheses"], want))
            assert same_b >= len(want) - max(1, len(want) // 8), same_b
            ref_scores = [sc[0] for sc in g["runs"][name]["scores"]]
            # (the fused engine restates the reference's LogSoftMax order; the op-by-op route's generic
            # ops are within the int8 noise bound of the other Level-1 tests)
            assert max(abs(x - y) for x, y in zip(a["scores"], ref_scores)) < (1e-4 if key == "tb" else 1e-2)
            ok = [x == y for x, y in zip(b["hypotheses"], want)]
            assert max(abs(x - y) for x, y, k in zip(b["scores"], ref_scores, ok) if k) < 1e-2


@pytest.mark.parametrize("ct", ["int8", "float32"])
def test_level15_reference_generator_on_the_fused_engine(gpu, golden, ct):
    """The same for ctranslate2::Generator: DecoderReplica::run_generation (include_prompt_in_result =
    false) of the reference's own Generator(model, Device::CUDA) served by ct2x_generate_batch - the
    decoder-only golden continuations of the reference on Device::CPU, token for token."""
    from ctranslate2_amd import model_builder as mb
    key = "lm_small"
    g = golden[key + ("_f32" if ct == "float32" else "")]
    path = "/tmp/ct2x_test_%s" % key
    if not os.path.exists(os.path.join(path, "model.bin")):
        mb.build_decoder_lm(path, **GC.SYNTHETIC_LM[key]["model"])
    runs = [{"beam": 1, "max_len": g["max_len"], "min_len": g["max_len"]}]
    fused = run_level1(path, ct, {"prompts": g["prompts"]}, runs, fused=True)
    plain = run_level1(path, ct, {"prompts": g["prompts"]}, runs, fused=False)
    assert fused["fused_calls"] == 1 and plain["fused_calls"] == 0
    same = sum(int(a == b) for a, b in zip(fused["runs"][0]["hypotheses"], g["generated"]))
    print("Level 1.5 generator %s %s: %d/%d continuations identical to Device::CPU" % (key, ct, same, len(g["generated"])))
    assert fused["runs"][0]["hypotheses"] == g["generated"]
    assert max(abs(x - y) for x, y in zip(fused["runs"][0]["scores"], g["scores"])) < 1e-3
    same_plain = sum(int(a == b) for a, b in zip(plain["runs"][0]["hypotheses"], g["generated"]))
    assert same_plain >= len(g["generated"]) // 2
