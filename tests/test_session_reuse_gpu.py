"""Session reuse (Translator::session, c_api_engine.cc): the buffers and captured step graphs of
the last plain greedy call serve the next call of the same padded shape and device-side options.
Every call must return what a fresh session returns. The switch is read once per process
(CT2X_SESSION_CACHE), so both settings run in their own interpreter on the same call sequence."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))

SCRIPT = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(here)r)
import ctranslate2_amd as ct2
from ctranslate2_amd import model_builder as mb
import golden_cases as GC
out, hits = [], []
t = ct2.Translator(os.path.join(%(here)r, "golden", "models", "aren-transliteration-i8"), device="cuda",
                   compute_type="int8")
def run(src, **kw):
    res = t.translate_batch(src, **kw)
    out.append([[r.hypotheses, [float(s) for s in r.scores]] for r in res])
    hits.append(t.last_stats()["session_hits"])
A = GC.AREN_SOURCES[:4]
B = [s[::-1] for s in A]                      # same lengths, other letters: same padded shape
C = [s + s[:2] for s in A]                    # longer: another shape
kw = dict(beam_size=1, return_scores=True, max_decoding_length=24)
run(A, **kw)                                  # miss (first)
run(B, **kw)                                  # hit
run(A, **kw)                                  # hit: identical to call 0
run(C, **kw)                                  # miss: shape
run(A, **kw)                                  # miss: shape again
run(A, **dict(kw, max_decoding_length=20))    # miss: option
run(A, **dict(kw, max_decoding_length=20))    # hit
run(A, **dict(kw, return_scores=False))       # miss: option
run(A, beam_size=2, return_scores=True, max_decoding_length=24)   # miss: another search
run(A, **kw)                                  # miss
run(A, target_prefix=[["a"], [], ["b"], []], **kw)                # not plain
run(A, **kw)                                  # miss
run(A, repetition_penalty=1.3, **kw)          # not plain
run(B[:3], **kw)                              # miss: rows
run(A[:3], **kw)                              # hit (same lengths as B[:3])
run(A[:3], use_graph=False, **kw)             # miss: option
run(B[:3], use_graph=False, **kw)             # hit, eager
run(A[:3], min_decoding_length=9, **kw)       # miss: option
run(B[:3], min_decoding_length=9, **kw)       # hit
# early stop by an end token other than </s> changes a kernel argument: part of the key
run(A, end_token="t", **kw)
run(B, end_token="t", **kw)
# beam search sessions are kept too (ancestry tables back to the identity, first cache set)
kwb = dict(beam_size=4, num_hypotheses=2, return_scores=True, max_decoding_length=24)
run(A, **kwb)                                 # miss
run(B, **kwb)                                 # hit
run(A, **kwb)                                 # hit: identical to the first
run(A, length_penalty=0.6, **kwb)             # hit: the length penalty is host arithmetic
run(A, **dict(kwb, beam_size=3))              # miss
run(C, **dict(kwb, beam_size=3))              # miss: shape
run(C[::-1], **dict(kwb, beam_size=3))        # miss: other lengths per row, same padded shape -> hit
# a synthetic Transformer with a vocabulary wide enough for the fused sampler, long forced decoding
d = "/tmp/ct2x_session_tb"
if not os.path.exists(os.path.join(d, "model.bin")):
    mb.build_transformer(d, num_encoder_layers=2, num_decoder_layers=2, d_model=256, ffn_dim=512,
                         num_heads=4, vocab_size=9000, seed=5)
t2 = ct2.Translator(d, device="cuda", compute_type="int8")
rng = np.random.default_rng(3)
ids = []
for call in range(4):
    src = [rng.integers(4, 9000, 11).tolist() for _ in range(8)]
    h, _ = t2.translate_ids(src, beam_size=1, max_decoding_length=30, min_decoding_length=30)
    ids.append(h)
hits.append(t2.last_stats()["session_hits"])
print(json.dumps({"out": out, "hits": hits, "ids": ids}))
"""


def _run(cache):
    env = dict(os.environ)
    env["CT2X_SESSION_CACHE"] = "1" if cache else "0"
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT, "here": HERE}], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
def test_a_reused_session_returns_what_a_fresh_one_returns(gpu):
    fresh = _run(False)
    reused = _run(True)
    assert reused["out"] == fresh["out"]
    assert reused["ids"] == fresh["ids"]
    assert set(fresh["hits"]) == {0}
    # the calls marked "hit" above, cumulative
    assert reused["hits"][:19] == [0, 1, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 4, 4, 5, 5, 6]
    assert reused["hits"][20] == reused["hits"][19] + 1
    h = reused["hits"][20]
    assert reused["hits"][21:28] == [h, h + 1, h + 2, h + 3, h + 3, h + 3, h + 4]
    assert reused["out"][21] == reused["out"][23]
    assert reused["hits"][-1] == 3
    # the same sentences gave the same result with and without a session in between
    assert reused["out"][0] == reused["out"][2]
