# -*- coding: utf-8 -*-
"""The reference's own translator tests (tests/translator_test.cc, the SearchVariantTest and
TranslatorTest cases on the hot path) restated against this backend: same trained fixture
(data/models/v2/aren-transliteration, float32 - default_model_dir()), same inputs, same expected
outputs, beam sizes 1 and 4 as in INSTANTIATE_TEST_SUITE_P(:375-379). Each test names the
reference test it mirrors. Cases that need attention weights, biased decoding or alternatives are
not mirrored (those options are rejected, see include/ct2_mi355.hpp)."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
MODEL = os.path.join(HERE, "golden", "models", "aren-transliteration")

pytestmark = pytest.mark.gpu

INPUT = ["آ", "ت", "ز", "م", "و", "ن"]
_translator = {}


def translator():
    import ctranslate2_amd as ct2
    if "t" not in _translator:
        _translator["t"] = ct2.Translator(MODEL, device="cuda")
    return _translator["t"]


def translate(source, target_prefix=None, **options):
    return translator().translate_batch(source, target_prefix=target_prefix, **options)


BEAMS = [1, 4]


@pytest.mark.parametrize("beam", BEAMS)
def test_set_max_decoding_length(gpu, beam):  # SetMaxDecodingLength :106
    assert len(translate([INPUT], beam_size=beam, max_decoding_length=3)[0].hypotheses[0]) == 3


@pytest.mark.parametrize("beam", BEAMS)
def test_set_min_decoding_length(gpu, beam):  # SetMinDecodingLength :116
    assert len(translate([INPUT], beam_size=beam, min_decoding_length=8)[0].hypotheses[0]) == 8


@pytest.mark.parametrize("beam", BEAMS)
def test_set_max_input_length(gpu, beam):  # SetMaxInputLength :126 (without the attention part)
    assert len(translate([INPUT], beam_size=beam, max_input_length=3)[0].hypotheses[0]) == 3


@pytest.mark.parametrize("beam", BEAMS)
def test_return_all_hypotheses(gpu, beam):  # ReturnAllHypotheses :148
    assert len(translate([INPUT], beam_size=beam, num_hypotheses=beam)[0].hypotheses) == beam


@pytest.mark.parametrize("beam", BEAMS)
def test_return_end_token(gpu, beam):  # ReturnEndToken :206
    res = translate([INPUT], beam_size=beam, return_end_token=True)[0]
    assert res.hypotheses[0] == ["a", "t", "z", "m", "o", "n", "</s>"]


@pytest.mark.parametrize("beam", BEAMS)
def test_translate_with_prefix(gpu, beam):  # TranslateWithPrefix :218
    res = translate([INPUT], [["a", "t", "s"]], beam_size=beam, num_hypotheses=beam)[0]
    assert len(res.hypotheses) == beam
    assert res.hypotheses[0] == ["a", "t", "s", "u", "m", "o", "n"]


@pytest.mark.parametrize("beam", BEAMS)
def test_translate_batch(gpu, beam):  # TranslateBatch :238
    res = translate([["آ", "ز", "ا"], INPUT], beam_size=beam, return_scores=True)
    assert len(res[0].scores) > 0 and len(res[1].scores) > 0
    assert res[0].hypotheses[0] == ["a", "z", "z", "a"]
    assert res[1].hypotheses[0] == ["a", "t", "z", "m", "o", "n"]


@pytest.mark.parametrize("beam", BEAMS)
def test_suppress_sequences(gpu, beam):  # SuppressSequences :256
    res = translate([INPUT], beam_size=beam, suppress_sequences=[["o"], ["t", "z", "m"]])[0]
    assert res.hypotheses[0] == ["a", "t", "z", "u", "m", "u", "n"]


@pytest.mark.parametrize("beam", BEAMS)
def test_suppress_sequence_oov(gpu, beam):  # SuppressSequenceOOV :268 (std::invalid_argument)
    with pytest.raises(ValueError):
        translate([INPUT], beam_size=beam, suppress_sequences=[["o"], ["t", "oovtoken", "m"]])


@pytest.mark.parametrize("beam", BEAMS)
def test_end_token(gpu, beam):  # EndToken :278
    assert translate([INPUT], beam_size=beam, end_token="m")[0].hypotheses[0] == ["a", "t", "z"]


@pytest.mark.parametrize("beam", BEAMS)
def test_end_token_oov(gpu, beam):  # EndTokenOOV :290
    with pytest.raises(ValueError):
        translate([INPUT], beam_size=beam, end_token="oovtoken")


@pytest.mark.parametrize("beam", BEAMS)
def test_repetition_penalty(gpu, beam):  # RepetitionPenalty :314
    tokens = translate([["ن"] * 5], beam_size=beam, repetition_penalty=100)[0].hypotheses[0]
    assert len(set(tokens)) == len(tokens)


@pytest.mark.parametrize("beam", BEAMS)
def test_no_repeat_ngram(gpu, beam):  # NoRepeatNgram :326
    n = 3
    out = "".join(translate([["ن"] * 50], beam_size=beam, no_repeat_ngram_size=n)[0].hypotheses[0])
    ngrams = {out[i:i + n] for i in range(len(out) - n)}
    assert len(ngrams) == len(out) - n


def _check_normalized_score(options, output_has_eos=True):  # check_normalized_score :344
    score = translate([INPUT], return_scores=True, length_penalty=0, **options)[0].scores[0]
    norm = translate([INPUT], return_scores=True, length_penalty=1, **options)[0]
    length = len(norm.hypotheses[0]) + (1 if output_has_eos else 0)
    assert abs(norm.scores[0] - score / length) < 1e-6


@pytest.mark.parametrize("beam", BEAMS)
def test_normalize_scores(gpu, beam):  # NormalizeScores :361
    _check_normalized_score(dict(beam_size=beam))


@pytest.mark.parametrize("beam", BEAMS)
def test_normalize_scores_no_eos(gpu, beam):  # NormalizeScoresNoEos :367
    _check_normalized_score(dict(beam_size=beam, max_decoding_length=6), output_has_eos=False)


def test_translate_empty_batch(gpu):  # TranslateEmptyBatch :381
    assert translate([]) == []


def _check_empty(result, with_score=False):  # check_empty_result :388
    assert result.hypotheses == [[]]
    if with_score:
        assert list(result.scores) == [0.0]
    else:
        assert len(result.scores) == 0


def test_translate_batch_with_empty_source(gpu):  # TranslateBatchWithEmptySource :411
    res = translate([[], ["آ", "ز", "ا"], [], INPUT, []])
    assert len(res) == 5
    _check_empty(res[0])
    assert res[1].hypotheses[0] == ["a", "z", "z", "a"]
    _check_empty(res[2])
    assert res[3].hypotheses[0] == ["a", "t", "z", "m", "o", "n"]
    _check_empty(res[4])


def test_translate_batch_with_only_empty_source(gpu):  # TranslateBatchWithOnlyEmptySource :424
    res = translate([[], []])
    assert len(res) == 2
    _check_empty(res[0])
    _check_empty(res[1])


def test_translate_empty_source_without_score(gpu):  # TranslateEmptySourceWithoutScore :433
    assert len(translate([[]], return_scores=False)[0].scores) == 0


def test_translate_batch_with_hard_prefix_and_empty(gpu):  # TranslateBatchWithHardPrefixAndEmpty :440
    res = translate([INPUT, INPUT, INPUT, [], ["آ", "ز", "ا"]],
                    [["a", "t", "s"], [], ["a", "t", "z", "o"], [], []])
    assert res[0].hypotheses[0] == ["a", "t", "s", "u", "m", "o", "n"]
    assert res[1].hypotheses[0] == ["a", "t", "z", "m", "o", "n"]
    assert res[2].hypotheses[0] == ["a", "t", "z", "o", "m", "o", "n"]
    assert res[3].hypotheses[0] == []
    assert res[4].hypotheses[0] == ["a", "z", "z", "a"]


def test_translate_prefix_with_large_beam(gpu):  # TranslatePrefixWithLargeBeam :725 (issue #277)
    res = translate([["أ", "و", "ل", "ي", "س", "س"]], [["u", "l", "i", "s", "e"]], beam_size=5)[0]
    assert res.hypotheses[0] == ["u", "l", "i", "s", "e", "s"]


def test_invalid_num_hypotheses(gpu):  # InvalidNumHypotheses :875
    with pytest.raises(Exception, match="hypotheses"):
        translate([INPUT], num_hypotheses=0)


def test_patience(gpu):  # Patience :883
    res = translate([INPUT], beam_size=4, patience=2, num_hypotheses=8)[0]
    assert len(res.hypotheses) == 8


def test_ignore_score(gpu):  # IgnoreScore :894
    res = translate([INPUT], beam_size=1, return_scores=False)[0]
    assert len(res.scores) == 0
    assert res.hypotheses[0] == ["a", "t", "z", "m", "o", "n"]


def test_same_beam_and_greedy_score(gpu):  # SameBeamAndGreedyScore :905
    greedy = translate([INPUT], beam_size=1, return_scores=True)[0].scores[0]
    beam = translate([INPUT], beam_size=2, return_scores=True)[0].scores[0]
    assert abs(greedy - beam) < 1e-5


def test_beam_size_larger_than_vocab_size(gpu):  # BeamSizeLargerThanVocabSize :917
    # 22 * 2 = 44 candidates are retrieved from the model output but the vocabulary size is 42
    res = translate([INPUT], beam_size=22, num_hypotheses=22)[0]
    assert len(res.hypotheses) == 22


def test_scoring(gpu):  # TranslatorTest.Scoring :942 (expected scores of the reference, 1e-5)
    source = [INPUT, ["آ", "ت", "ش", "ي", "س", "و", "ن"], ["آ", "ر", "ب", "ا", "ك", "ه"], [],
              ["آ", "ر", "ث", "ر"]]
    target = [["a", "t", "z", "m", "o", "n"], ["a", "c", "h", "i", "s", "o", "n"],
              ["a", "r", "b", "a", "k", "e"], [], ["a", "r", "t", "h", "e", "r"]]
    expected = [
        [-0.106023, -0.065410, -0.056002, -0.447953, -0.230714, -0.092184, -0.063463],
        [-0.072660, -0.300309, -0.181187, -0.395671, -0.025631, -0.123466, -0.002034, -0.012639],
        [-0.103136, -0.089504, -0.063889, -0.007327, -0.452072, -0.060154, -0.016636],
        [0],
        [-0.076704, -0.036037, -0.029253, -0.030273, -0.149276, -0.002440, -0.003742],
    ]
    results = translator().score_batch(source, target)
    assert len(results) == len(expected)
    for res, tgt, exp in zip(results, target, expected):
        assert res.tokens == tgt + ["</s>"]
        np.testing.assert_allclose(res.log_probs, exp, atol=1e-5)


def test_scoring_max_input_length(gpu):  # ScoringMaxInputLength :974
    res = translator().score_batch([INPUT], [["a", "t", "z", "m", "o", "n"]], max_input_length=4)[0]
    assert res.tokens == ["a", "t", "z", "</s>"]
    assert len(res.log_probs) == 4
