"""not gpu: the committed Whisper goldens are what the reference's own models::Whisper computes on
Device::CPU (oracle/_ref) for the deterministic synthetic model of tests/whisper_cases.py, and the
host-side pieces of the Whisper glue that need no device."""
import json
import os

import numpy as np
import pytest

import _refshim as R
import whisper_cases as WC

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(not R.available(), reason="oracle/_ref (the reference CPU library) is not built")
def test_golden_is_reproduced_by_the_reference():
    with open(os.path.join(HERE, "golden", "whisper_golden.json")) as f:
        j = json.load(f)
    z = np.load(os.path.join(HERE, "golden", "whisper_golden.npz"))
    path = WC.build("/tmp/ct2x_whisper_float32", "float32")
    w = R.RefWhisper(path, compute_type="float32")
    try:
        assert w.info() == j["float32"]["info"]
        res = WC.run_cases(w, only={"greedy_timestamps", "beam5_timestamps", "timestamp_in_prefix"})
    finally:
        w.close()
    np.testing.assert_allclose(res["encode@"], z["float32.encode"], rtol=0, atol=1e-5)
    for name in ("greedy_timestamps", "beam5_timestamps", "timestamp_in_prefix"):
        assert res["generate." + name]["hypotheses"] == j["float32"]["generate." + name]["hypotheses"]
    assert res["detect_language"]["ids"] == j["float32"]["detect_language"]["ids"]
    for name in WC.ALIGN:
        assert res["align." + name]["pairs"] == j["float32"]["align." + name]["pairs"]


def test_golden_exercises_the_timestamp_rules():
    """The fixture is only worth something if the rules fire: timestamps come in pairs, never
    decrease, and runs without <|notimestamps|> start with a timestamp."""
    with open(os.path.join(HERE, "golden", "whisper_golden.json")) as f:
        j = json.load(f)["float32"]
    pairs = 0
    for name in ("greedy_timestamps", "beam5_timestamps", "sot_only", "options"):
        for hyps in j["generate." + name]["hypotheses"]:
            h = hyps[0]
            assert h[0] >= WC.TS0, (name, h)
            ts = [t for t in h if t >= WC.TS0]
            assert ts == sorted(ts), (name, h)
            pairs += sum(1 for a, b in zip(h, h[1:]) if a >= WC.TS0 and b >= WC.TS0)
    assert pairs >= 4
    for hyps in j["generate.greedy_notimestamps"]["hypotheses"]:
        assert len(hyps[0]) >= 1
    # hard prefix after the task tokens is part of the hypothesis (return_prefix)
    assert j["generate.text_after_task"]["hypotheses"][0][0][:2] == [5, 9]


def test_whisper_vocabulary_layout():
    from ctranslate2_amd import model_builder as mb
    tokens = mb.whisper_vocab_tokens(WC.MODEL["num_text_tokens"], WC.MODEL["languages"],
                                     WC.MODEL["num_timestamps"])
    assert tokens[WC.EOT] == "<|endoftext|>" and tokens[WC.SOT] == "<|startoftranscript|>"
    assert tokens[WC.NOSPEECH] == "<|nospeech|>" and tokens[WC.NOTIMESTAMPS] == "<|notimestamps|>"
    assert tokens[WC.TS0] == "<|0.00|>" and len(tokens) == WC.TS0 + WC.MODEL["num_timestamps"]
    assert WC.NOSPEECH - WC.SOT - 5 == WC.NUM_LANGUAGES  # whisper.cc:77


def test_prompt_layout_rules_on_the_host():
    """check_prompts (src/models/whisper.cc:149-200) behind ct2x_host_whisper_check_prompts: the
    start-of-transcript position and the number of task tokens must agree across the batch."""
    import ctypes as C
    from ctranslate2_amd import _lib
    L = _lib.lib()

    def run(prompts):
        lengths = np.array([len(p) for p in prompts], np.int32)
        stride = int(lengths.max())
        ids = np.zeros((len(prompts), stride), np.int32)
        for i, p in enumerate(prompts):
            ids[i, :len(p)] = p
        si, pl = C.c_int32(), C.c_int32()
        rc = L.ct2x_host_whisper_check_prompts(_lib.ptr(ids), _lib.ptr(lengths), C.c_int32(len(prompts)),
                                               C.c_int32(stride), C.c_int32(WC.SOT),
                                               C.c_int32(WC.NOTIMESTAMPS), C.byref(si), C.byref(pl))
        return rc, si.value, pl.value, L.ct2x_last_error().decode()

    assert run([[WC.SOT, WC.EN, WC.TRANSCRIBE, WC.NOTIMESTAMPS]] * 2)[:3] == (0, 0, 4)
    assert run([[WC.SOT, WC.EN, WC.TRANSCRIBE, WC.NOTIMESTAMPS, 5, 9]])[:3] == (0, 0, 4)
    assert run([[WC.SOT]])[:3] == (0, 0, 1)
    # timestamps are not task tokens: the prompt ends in front of them
    assert run([[WC.SOT, WC.EN, WC.TRANSCRIBE, WC.TS0 + 4, 5]])[:3] == (0, 0, 3)
    assert run([[WC.STARTOFPREV, 5, 6, 7, WC.SOT, WC.FR, WC.TRANSCRIBE]])[:3] == (0, 4, 7)
    rc, _, _, msg = run([[5, 6]])
    assert rc != 0 and "<|startoftranscript|> token was not found in the prompt" in msg
    rc, _, _, msg = run([[WC.SOT, WC.EN], [WC.STARTOFPREV, 5, WC.SOT, WC.EN]])
    assert rc != 0 and "to be at the same position" in msg
    rc, _, _, msg = run([[WC.SOT, WC.EN, WC.TRANSCRIBE], [WC.SOT, WC.EN]])
    assert rc != 0 and "same number of task tokens" in msg


@pytest.mark.skipif(not R.available(), reason="oracle/_ref (the reference CPU library) is not built")
@pytest.mark.parametrize("stride,length,in_c,out_c", [(1, 24, 8, 64), (2, 24, 64, 64), (2, 37, 16, 40),
                                                      (1, 5, 3, 7)])
def test_int8_convolution_restatement_is_the_reference(stride, length, in_c, out_c):
    """oracle/whisper_oracle.py conv1d_k3_s8 (what conv_dequantize_kernel + the int8 GEMM reproduce
    on the device) against ops::Conv1D of the reference library on CPU: bit for bit."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
    import whisper_oracle as W
    import _oracle as O
    from ctranslate2_amd import model_builder as mb
    rng = np.random.default_rng(stride * 100 + length + in_c + out_c)
    x = rng.standard_normal((3, in_c, length)).astype(np.float32)
    w = (rng.standard_normal((out_c, in_c, 3)) * 0.2).astype(np.float32)
    bias = (rng.standard_normal(out_c) * 0.1).astype(np.float32)
    q, s = mb.quantize_int8(w.reshape(out_c, -1))
    wq = q.reshape(out_c, in_c, 3)
    for act in (3, -1):  # GELU (Whisper), none
        ref = R.conv1d(x, wq, bias, w_scale=s, stride=stride, padding=1, act=act)
        got = W.conv1d_k3_s8(O, x, wq, s, bias, stride, 1, act)
        assert got.shape == ref.shape
        assert np.array_equal(got, ref), float(np.abs(got - ref).max())
    # the float arm is the same im2col through MKL's sgemm: float rounding of the sum order only
    ref = R.conv1d(x, w, bias, stride=stride, padding=1, act=-1)
    cols = W.im2col_k3(x, stride, 1)
    got = np.einsum("btk,ok->bot", cols.astype(np.float64), w.reshape(out_c, -1).astype(np.float64)) + bias[:, None]
    np.testing.assert_allclose(ref, got, rtol=1e-5, atol=1e-5)


@pytest.mark.skipif(not R.available(), reason="oracle/_ref (the reference CPU library) is not built")
def test_axis_layer_norm_restatement_is_the_reference():
    """layer_norm_axis (the device kernel of Whisper::align follows the same order, FMA
    contractions included) against ops::LayerNorm(-2, 0) of the reference library: bit for bit,
    also on nearly constant columns where the E[x^2] - mean^2 form cancels."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
    import whisper_oracle as W
    rng = np.random.default_rng(5)
    for shape in ((3, 9, 12), (6, 31, 750), (2, 1, 5)):
        x = rng.random(shape).astype(np.float32)
        x[0, :, 0] = 1.0 / 12            # constant column: variance 0 after the clamp -> inf / nan
        x[-1, :, -1] += 1000.0           # large offset: catastrophic cancellation
        ref = R.layer_norm_axis(x)
        got = W.layer_norm_axis(x)
        assert np.array_equal(np.isnan(ref), np.isnan(got))
        ok = ~np.isnan(ref)
        assert np.array_equal(ref[ok], got[ok]), float(np.nanmax(np.abs(ref - got)))
