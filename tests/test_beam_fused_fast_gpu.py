"""Beam search of the 16-bit compute types: the LogSoftMax's 2 * beam best scores per row leaving as
candidates (beam_logprobs_kernel, parallel-sum form, rows of any length - round 6) against the same
engine writing the rows and running TopK over them (CT2X_BEAM_FUSED_FAST=0): identical hypotheses
and scores, bit for bit - the candidates carry the same values in the same total order
(src/decoding.cc:548-563, src/ops/topk_gpu.cu:182-335: value descending, flat index ascending).
Each setting runs in its own interpreter (the switch is read once per process)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_SCRIPT = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import ctranslate2_amd as ct2
from ctranslate2_amd import model_builder as mb
cfg = json.loads(sys.argv[1])
path = "/tmp/ct2x_test_beamfast_v%%d" %% cfg["vocab"]
if not os.path.exists(os.path.join(path, "model.bin")):
    mb.build_transformer(path, num_encoder_layers=2, num_decoder_layers=2, d_model=128, ffn_dim=256,
                         num_heads=4, vocab_size=cfg["vocab"], seed=7, proj_std=0.3)
t = ct2.Translator(path, compute_type=cfg["compute_type"])
rng = np.random.default_rng(3)
src = [rng.integers(4, cfg["vocab"], n).tolist() for n in (9, 5, 12, 7, 3, 10)]
out = {}
for beam in cfg["beams"]:
    hyps, scores = t.translate_ids(src, beam_size=beam, max_decoding_length=12, min_decoding_length=4,
                                   return_scores=True, num_hypotheses=min(beam, 2))
    out[str(beam)] = [hyps, np.asarray(scores, np.float32).view(np.uint32).tolist()]
print("RESULT" + json.dumps(out))
"""


def _run(cfg, fused, switch="CT2X_BEAM_FUSED_FAST"):
    env = dict(os.environ)
    env[switch] = "1" if fused else "0"
    r = subprocess.run([sys.executable, "-c", _SCRIPT % {"root": ROOT}, json.dumps(cfg)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1]
    return json.loads(line[len("RESULT"):])


@pytest.mark.parametrize("vocab,compute_type", [(40000, "float16"), (40000, "bfloat16"), (300, "float16"),
                                                (70001, "float16")])
def test_fused_candidates_equal_rows_plus_topk(gpu, vocab, compute_type):
    """40000 / 70001 words: sixteen waves per row, the bounded list (70001: a row longer than the
    exact form's registers could hold, odd length); 300 words: fewer waves than candidates - the
    block-wide rounds."""
    cfg = {"vocab": vocab, "compute_type": compute_type, "beams": [4, 2, 5]}
    a, b = _run(cfg, True), _run(cfg, False)
    assert a == b


@pytest.mark.parametrize("vocab,compute_type", [(32000, "int8"), (32000, "float32"), (300, "int8"),
                                                (20001, "float32")])
def test_exact_candidates_equal_rows_plus_topk(gpu, vocab, compute_type):
    """The exact LogSoftMax's candidate form (int8 / float32 compute: the row in registers, the
    candidates listed by x BEFORE the ordered exp sum - within a margin below the topk-th largest
    wave maximum that covers every x the two roundings of y = b + (x + c) can collapse onto the
    bound's y - and turned into y by one wave once c is known) against the written rows + TopK
    (CT2X_BEAM_FUSED_TOPK=0): identical hypotheses and scores. 300 words: fewer waves than candidates
    (block-wide rounds over the re-read row); 20001: a ragged last register."""
    cfg = {"vocab": vocab, "compute_type": compute_type, "beams": [4, 2, 5]}
    a, b = _run(cfg, True, "CT2X_BEAM_FUSED_TOPK"), _run(cfg, False, "CT2X_BEAM_FUSED_TOPK")
    assert a == b
