"""-m gpu: decoder layer variants and Su rotary scaling in the fused engine
against the reference's CPU path (tests/golden/layer_variants_golden.*, oracle/_ref):
parallel residual with a shared norm / an input norm / input + post-attention norms
(src/layers/transformer.cc:285-317), norms before and after both blocks with layer_scalar
(:218-283), Su scaling with the long and the short factors (src/layers/attention_layer.cc:254-343).
int8 models: teacher-forced logits identical, greedy and beam-2 continuations token for token."""
import json
import os

import numpy as np
import pytest

import golden_cases as GC

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "layer_variants_golden.json")) as f:
        return json.load(f), np.load(os.path.join(HERE, "golden", "layer_variants_golden.npz"))


@pytest.mark.parametrize("name", sorted(GC.LAYER_VARIANTS))
def test_layer_variant_matches_the_reference(gpu, golden, name):
    import ctranslate2_amd as ct2
    from ctranslate2_amd import model_builder as mb
    j, z = golden
    g = j[name]
    path = "/tmp/ct2x_variant_%s" % name
    if not os.path.exists(os.path.join(path, "model.bin")):
        mb.build_decoder_lm(path, **GC.LAYER_VARIANTS[name])
    t = ct2.Translator(path, device="cuda", compute_type="int8")
    logits = t.step_logits([], np.array(g["tgt_in"], np.int32))
    ref = z[name + ".logits"]
    err = float(np.abs(logits - ref).max())
    agree = float(np.mean(logits.argmax(-1) == ref.argmax(-1)))
    print("%s: teacher-forced logits max|d| = %.3g (|x| up to %.2f), arg-max agreement %.4f"
          % (name, err, float(np.abs(ref).max()), agree))
    case = GC.LAYER_VARIANT_CASE
    gen, scores = t.generate_ids(g["prompts"], max_length=case["max_len"],
                                 min_length=case["max_len"], return_scores=True)
    same = sum(int(a == b) for a, b in zip(gen, g["generated"]))
    gen2, scores2 = t.generate_ids(g["prompts"], beam_size=2, max_length=case["max_len"],
                                   min_length=case["max_len"], return_scores=True)
    same2 = sum(int(a == b) for a, b in zip(gen2, g["beam2"]))
    print("%s: greedy %d/%d, beam 2 %d/%d continuations token-identical"
          % (name, same, len(gen), same2, len(gen2)))
    assert err <= 2e-4 * max(1.0, float(np.abs(ref).max()))
    assert agree >= 0.99
    assert gen == g["generated"]
    np.testing.assert_allclose(scores, np.array(g["scores"], np.float32), rtol=1e-4, atol=1e-4)
    assert gen2 == g["beam2"]
