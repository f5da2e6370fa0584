"""Dev-time script (needs /root/reference and oracle/_ref built): golden outputs of the REAL
reference Translator::translate_batch with the logits-processor options (repetition_penalty,
no_repeat_ngram_size, suppress_sequences incl. single-token = disable_unk style bans), greedy and
beam 4, on a synthetic model in float32 and int8 -> tests/golden/processors_golden.json.

  python tests/golden/make_golden_processors.py
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
    rng = np.random.default_rng(91)
    vocab = cfg["model"]["vocab_size"]
    sources = [rng.integers(4, vocab, n).tolist() for n in (12, 9, 12, 5, 10, 12, 7, 11)]
    max_len, min_len = 24, 10
    for ct, suffix in (("float32", "_f32"), ("int8", "")):
        m = R.RefModel(path, ct, 1, 4)
        base, _ = m.translate_processors_ids(sources, beam_size=1, max_len=max_len, min_len=min_len)
        # bans derived from the unconstrained output so that every processor really bites:
        # the first generated token of sentence 0 (always), a bigram and a trigram that occur.
        suppress = [[base[0][0]], base[1][1:3], base[2][3:6], [base[3][2]]]
        cases = {
            "plain": {},
            "penalty": {"repetition_penalty": 1.7},
            "penalty_lt1": {"repetition_penalty": 0.6},
            "ngram1": {"no_repeat_ngram_size": 1},
            "ngram2": {"no_repeat_ngram_size": 2},
            "ngram3": {"no_repeat_ngram_size": 3},
            "suppress": {"suppress_sequences_ids": suppress},
            "all": {"repetition_penalty": 1.3, "no_repeat_ngram_size": 2,
                    "suppress_sequences_ids": suppress},
        }
        runs = {}
        for name, kw in cases.items():
            for beam in (1, 4):
                hyp, scores = m.translate_processors_ids(sources, beam_size=beam, max_len=max_len,
                                                         min_len=min_len, length_penalty=1.0, **kw)
                runs["%s_beam%d" % (name, beam)] = {"options": kw, "beam": beam, "hypotheses": hyp,
                                                    "scores": scores.tolist()}
        out["small" + suffix] = {"sources": sources, "max_len": max_len, "min_len": min_len,
                                 "runs": runs}
        m.close()
        differ = {k: sum(a != b for a, b in zip(v["hypotheses"], runs["plain_beam%d" % v["beam"]]["hypotheses"]))
                  for k, v in runs.items()}
        print(ct, "sentences changed by each case:", differ)
    with open(os.path.join(HERE, "processors_golden.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
