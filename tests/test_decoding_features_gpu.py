"""-m gpu parity for the remaining decoding features against the REAL reference
(tests/golden/make_golden_decoding_features.py -> decoding_features_golden.{json,npz}):
  * Generator.forward_batch / score_batch (one sequence-mode pass),
  * Translator return_alternatives (+ min_alternative_expansion_prob), biased decoding
    (prefix_bias_beta), return_attention, coverage_penalty.
float32: tokens identical, scores / logits / attention to float rounding; int8: the path is
bit-exact with the reference's CPU arithmetic, so tokens and scores are identical too (the biased
log-probabilities go through logf instead of MKL's vsLn: 1e-5)."""
import json
import os

import numpy as np
import pytest

import golden_cases as GC
from test_generator_gpu import get_lm
from test_translator_gpu import get_translator, host_math_is_mkl

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "decoding_features_golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def arrays():
    return np.load(os.path.join(HERE, "golden", "decoding_features_golden.npz"))


@pytest.mark.parametrize("key", ["lm_tiny", "lm_small"])
@pytest.mark.parametrize("ct", ["float32", "int8"])
def test_generator_score_batch(gpu, golden, key, ct):
    """Generator::score_batch: log P(token k+1 | tokens ..k), sequences of < 2 tokens skipped,
    max_input_length truncation (Vocabulary::to_ids)."""
    g = golden[key + ("_f32" if ct == "float32" else "")]
    t = get_lm(key, ct)
    for field, kw in (("scores", {}), ("scores_max_input_length_7", {"max_input_length": 7})):
        got = t.score_tokens_ids(g["sequences"], **kw)
        assert [len(s) for s in got] == [len(s) for s in g[field]]
        for a, b in zip(got, g[field]):
            np.testing.assert_allclose(a, np.array(b, np.float32), atol=2e-5 if ct == "float32" else 1e-6)


@pytest.mark.parametrize("ct", ["float32", "int8"])
@pytest.mark.parametrize("log_probs", [False, True])
def test_generator_forward_batch(gpu, golden, arrays, ct, log_probs):
    """Generator::forward_batch: the logits (log-probabilities) of every position of ragged
    sequences; positions past a sequence's end are the reference's padding rows and not compared."""
    suffix = "_f32" if ct == "float32" else ""
    g = golden["lm_tiny" + suffix]
    ref = arrays["lm_tiny%s_forward_%s" % (suffix, "log_probs" if log_probs else "logits")]
    seqs = [s for s in g["sequences"] if len(s) > 0]
    got = get_lm("lm_tiny", ct).forward_ids(seqs, return_log_probs=log_probs)
    assert got.shape == ref.shape
    for b, s in enumerate(seqs):
        np.testing.assert_allclose(got[b, :len(s)], ref[b, :len(s)], atol=2e-5 if ct == "float32" else 2e-6)


def _run(t, g, run, **extra):
    kw = dict(run["options"])
    kw.update(extra)
    prefixes = kw.pop("prefixes", None)
    one_by_one = kw.pop("one_by_one", False)
    args = dict(beam_size=kw.pop("beam_size"), max_decoding_length=g["max_len"],
                min_decoding_length=kw.pop("min_len", 1), num_hypotheses=kw.pop("num_hypotheses", 1),
                length_penalty=kw.pop("length_penalty", 1.0), return_scores=True, **kw)
    if not one_by_one:
        return t.translate_ids(g["sources"], target_prefix_ids=prefixes, **args)
    # (goldens made one sentence at a time: see make_golden_decoding_features.py on the first-step
    # attention of the reference's batched CPU beam search)
    outs = [t.translate_ids([src], target_prefix_ids=None if prefixes is None else [prefixes[b]], **args)
            for b, src in enumerate(g["sources"])]
    merged = [[o[0][0] for o in outs], np.concatenate([o[1] for o in outs], axis=0)]
    if len(outs[0]) == 3:
        merged.append([o[2][0] for o in outs])
    return tuple(merged)


def _check(got_hyps, got_scores, run, ct, score_tol):
    same = 0
    total = 0
    for b, ref_hyps in enumerate(run["hypotheses"]):
        assert len(got_hyps[b]) == len(ref_hyps), (b, got_hyps[b], ref_hyps)
        for k, ref in enumerate(ref_hyps):
            total += 1
            if got_hyps[b][k] == ref:
                same += 1
                assert abs(got_scores[b, k] - run["scores"][b][k]) < score_tol, (b, k)
    exact = ct == "float32" or host_math_is_mkl()
    assert same >= total - (0 if exact else 2), (same, total)
    return same, total


ALT_RUNS = ["alt3", "alt3_prefix", "alt4_minprob", "alt2_lp"]


@pytest.mark.parametrize("key", ["tiny", "small"])
@pytest.mark.parametrize("ct", ["float32", "int8"])
@pytest.mark.parametrize("beam", [1, 2])
@pytest.mark.parametrize("name", ALT_RUNS)
def test_return_alternatives(gpu, golden, key, ct, beam, name):
    """decode_alternatives (src/decoding.cc:1122-1290): prefix in one sequence-mode pass, the
    num_hypotheses best next tokens, each continued independently; alternatives below
    min_alternative_expansion_prob are dropped (fewer hypotheses for that sentence)."""
    g = golden[key + ("_f32" if ct == "float32" else "")]
    run = g["runs"]["%s_beam%d" % (name, beam)]
    kw = dict(run["options"])
    hyps, scores = _run(get_translator(gpu, key, ct), g, run)
    same, total = _check(hyps, scores, run, ct, 2e-4 if ct == "float32" else 2e-5)
    print("%s %s %s beam %d: %d/%d alternatives token-identical" % (key, ct, name, beam, same, total))
    if "prefixes" in kw:
        for b, p in enumerate(kw["prefixes"]):
            assert all(h[:len(p)] == p for h in hyps[b])


@pytest.mark.parametrize("key", ["tiny", "small"])
@pytest.mark.parametrize("ct", ["float32", "int8"])
@pytest.mark.parametrize("beam", [1, 2])
@pytest.mark.parametrize("name", ["bias03", "bias08", "bias05_own"])
def test_biased_decoding(gpu, golden, key, ct, beam, name):
    """prefix_bias_beta (BiasedDecoder, src/decoding.cc:256-343): while a beam follows the prefix its
    distribution is (1 - beta) softmax + beta onehot(prefix token); always a beam search."""
    g = golden[key + ("_f32" if ct == "float32" else "")]
    run = g["runs"]["%s_beam%d" % (name, beam)]
    hyps, scores = _run(get_translator(gpu, key, ct), g, run)
    same, total = _check(hyps, scores, run, ct, 2e-4)
    print("%s %s %s beam %d: %d/%d token-identical" % (key, ct, name, beam, same, total))


@pytest.mark.parametrize("key", ["tiny", "small"])
@pytest.mark.parametrize("ct", ["float32", "int8"])
@pytest.mark.parametrize("beam", [1, 2])
@pytest.mark.parametrize("name", ["attention", "attention_prefix"])
def test_return_attention(gpu, golden, arrays, key, ct, beam, name):
    """return_attention: one source attention vector (alignment layer / heads, implicit special
    tokens removed) per returned token, for greedy and beam search, with and without a prefix."""
    suffix = "_f32" if ct == "float32" else ""
    g = golden[key + suffix]
    run = g["runs"]["%s_beam%d" % (name, beam)]
    hyps, scores, att = _run(get_translator(gpu, key, ct), g, run, return_attention=True)
    _check(hyps, scores, run, ct, 2e-4 if ct == "float32" else 2e-5)
    checked = 0
    for b in range(len(g["sources"])):
        for k in range(len(run["hypotheses"][b])):
            if hyps[b][k] != run["hypotheses"][b][k]:
                continue
            ref = arrays["%s%s_%s_beam%d_att_%d_%d" % (key, suffix, name, beam, b, k)]
            n = len(g["sources"][b])
            assert att[b][k].shape == (len(hyps[b][k]), n)
            np.testing.assert_allclose(att[b][k], ref[:, :n], atol=2e-5)
            assert np.all(ref[:, n:] == 0)
            np.testing.assert_allclose(att[b][k].sum(axis=1), 1.0, atol=2e-2)
            checked += 1
    assert checked >= 1


@pytest.mark.parametrize("key", ["tiny", "small"])
@pytest.mark.parametrize("ct", ["float32", "int8"])
def test_coverage_penalty(gpu, golden, key, ct):
    """coverage_penalty (src/decoding.cc:176-207): the GNMT coverage term of the final scores needs
    the attention history of every hypothesis; it also disables the early exit of the beam search."""
    g = golden[key + ("_f32" if ct == "float32" else "")]
    run = g["runs"]["coverage_beam2"]
    hyps, scores = _run(get_translator(gpu, key, ct), g, run)
    _check(hyps, scores, run, ct, 5e-4)


def test_decoding_feature_validation(gpu):
    """The reference's option checks (src/decoding.cc:1040-1051)."""
    t = get_translator(gpu, "tiny", "float32")
    src = [[5, 6, 7]]
    with pytest.raises(Exception, match="smaller than 1"):
        t.translate_ids(src, target_prefix_ids=[[5]], prefix_bias_beta=1.0)
    with pytest.raises(Exception, match="not compatible"):
        t.translate_ids(src, target_prefix_ids=[[5]], prefix_bias_beta=0.5, return_alternatives=True,
                        num_hypotheses=2)
    with pytest.raises(Exception, match="between 0 and 1"):
        t.translate_ids(src, return_alternatives=True, num_hypotheses=2, min_alternative_expansion_prob=1.5)
    # num_hypotheses > beam_size is fine with return_alternatives, an error without
    t.translate_ids(src, beam_size=1, num_hypotheses=3, return_alternatives=True, max_decoding_length=5)
    with pytest.raises(Exception, match="hypotheses"):
        t.translate_ids(src, beam_size=1, num_hypotheses=3, max_decoding_length=5)
