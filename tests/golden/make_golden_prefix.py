"""Dev-time script (needs /root/reference and oracle/_ref built): golden outputs of the REAL
reference Translator::translate_batch(source, target_prefix, options) - greedy and beam 4 - on the
trained int8 fixture and a synthetic model -> tests/golden/prefix_golden.json.

  python tests/golden/make_golden_prefix.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import _refshim as R  # noqa: E402
import golden_cases as GC  # noqa: E402
from ctranslate2_amd import model_builder as mb  # noqa: E402


def main():
    assert R.available(), "build oracle/_ref first (make -C oracle ref)"
    out = {}
    cfg = GC.SYNTHETIC["small"]
    path = "/tmp/ct2x_golden_small"
    if not os.path.exists(os.path.join(path, "model.bin")):
        mb.build_transformer(path, **cfg["model"])
    rng = np.random.default_rng(77)
    vocab = cfg["model"]["vocab_size"]
    sources = [rng.integers(4, vocab, n).tolist() for n in (12, 9, 12, 5, 10, 12)]
    prefixes = [rng.integers(4, vocab, n).tolist() for n in (3, 0, 5, 1, 2, 4)]
    for ct, suffix in (("float32", "_f32"), ("int8", "")):
        m = R.RefModel(path, ct, 1, 4)
        runs = {}
        for beam in (1, 4):
            hyp, scores = m.translate_prefix_ids(sources, prefixes, beam_size=beam, max_len=14,
                                                 min_len=8, length_penalty=1.0)
            runs["beam%d" % beam] = {"hypotheses": hyp, "scores": scores.tolist()}
        out["small" + suffix] = {"sources": sources, "prefixes": prefixes, "max_len": 14, "min_len": 8,
                                 "runs": runs}
        m.close()
    with open(os.path.join(HERE, "prefix_golden.json"), "w") as f:
        json.dump(out, f)
    for k, v in out.items():
        print(k, [len(h) for h in v["runs"]["beam1"]["hypotheses"]],
              [h[:len(p)] == p for h, p in zip(v["runs"]["beam4"]["hypotheses"], v["prefixes"])])


if __name__ == "__main__":
    main()
