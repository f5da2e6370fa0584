"""Padder (src/padder.cc:8-74, TransformerEncoder::operator() src/layers/transformer.cc:
440-470, MultiHeadAttention::operator() src/layers/attention.cc:390-403, 590-596): with source
sentences of different lengths the encoder's row-wise layers run on the valid positions only; the
attention re-pads its fused projection and un-pads its context; the encoder output is padded again
(a padded position shows its sentence's last valid row). Every valid position keeps its bits, so the
hypotheses and scores are those of the padded run AND the reference's.

CT2X_PADDER is read once per process: the runs live in interpreters of their own."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

_SCRIPT = r"""
import json, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
from test_translator_gpu import get_translator
cfg = json.loads(sys.argv[1])
g = json.load(open(os.path.join(%(root)r, "tests", "golden", "model_golden.json")))[cfg["key"]]
model_key = cfg["key"][:-4] if cfg["key"].endswith("_f32") else cfg["key"]
t = get_translator(None, model_key, cfg["ct"])
out = {}
for beam in cfg["beams"]:
    t.profile_begin()
    hyps, scores = t.translate_ids(g["sources"], beam_size=beam, max_decoding_length=g["max_len"],
                                   min_decoding_length=g["min_len"], return_scores=True, length_penalty=1.0)
    prof = t.profile_end()
    out["beam%%d" %% beam] = {"hyps": hyps, "scores": np.asarray(scores).tolist(),
                             "padder_launches": sum(v["launches"] for k, v in prof.items() if k.startswith("padder_")),
                             "layers_rows": prof.get("padder_remove", {}).get("launches", 0)}
# a large ragged batch of the same model (hundreds of rows through every GEMM form)
rng = np.random.default_rng(7)
vocab = 40 if cfg["key"].startswith("aren") else 60
srcs = [rng.integers(4, vocab, int(n)).tolist() for n in rng.integers(1, 24, 96)]
hyps, scores = t.translate_ids(srcs, beam_size=1, max_decoding_length=12, return_scores=True)
out["big"] = {"hyps": hyps, "scores": np.asarray(scores).tolist()}
print("RESULT" + json.dumps(out))
"""


def _run(cfg, padder):
    env = dict(os.environ, CT2X_PADDER="1" if padder else "0")
    r = subprocess.run([sys.executable, "-c", _SCRIPT % {"root": ROOT}, json.dumps(cfg)], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1][6:])


@pytest.mark.parametrize("key,ct", [("aren-transliteration-i8", "int8"), ("aren-transliteration", "float32"),
                                    ("small", "int8"), ("small_f32", "float32"), ("tiny", "int8")])
def test_padding_removal_keeps_every_result(gpu, key, ct):
    with open(os.path.join(HERE, "golden", "model_golden.json")) as f:
        g = json.load(f)[key]
    beams = [int(k[4:]) for k in g["runs"]]
    cfg = {"key": key, "ct": ct, "beams": beams}
    on, off = _run(cfg, True), _run(cfg, False)
    n_layers = None
    for beam in beams:
        a, b = on["beam%d" % beam], off["beam%d" % beam]
        assert a["padder_launches"] > 0 and b["padder_launches"] == 0   # (the sources ARE ragged)
        assert a["hyps"] == b["hyps"]
        assert a["scores"] == b["scores"]                                # bit for bit
        ref = g["runs"]["beam%d" % beam]
        same = [h[0] == r[0] for h, r in zip(a["hyps"], ref["hypotheses"])]
        assert all(same) if ct == "float32" or beam == 1 else sum(same) >= len(same) - 1
    assert on["big"] == off["big"]
