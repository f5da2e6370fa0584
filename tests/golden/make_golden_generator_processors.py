"""Dev-time script (needs /root/reference and oracle/_ref built): golden outputs of the REAL
reference Generator::generate_batch with logits processors over RAGGED prompts - decoding starts
at step = shortest prompt - 1 and longer prompts continue as a hidden hard prefix, so this pins
the device-resident token history the processors read (greedy: id history offset by start_step;
beam: the tokens of the active beams incl. the hidden prefix) -> generator_processors_golden.json.
min_length = 0 (see make_golden_ragged.py for the reference's min-length quirk with a hidden
prefix in beam search).

  python tests/golden/make_golden_generator_processors.py
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
    key = "lm_small"
    cfg = GC.SYNTHETIC_LM[key]
    path = "/tmp/ct2x_golden_%s" % key
    if not os.path.exists(os.path.join(path, "model.bin")):
        mb.build_decoder_lm(path, **cfg["model"])
    rng = np.random.default_rng(cfg["data_seed"] + 300)
    vocab = cfg["model"]["vocab_size"]
    prompts = [rng.integers(4, vocab, n).tolist() for n in (12, 7, 12, 5, 9, 10)]
    max_len = 20
    out = {}
    for ct, suffix in (("float32", "_f32"), ("int8", "")):
        m = R.RefModel(path, ct, 1, 4)
        base, _ = m.generate_processors_ids(prompts, beam_size=1, max_len=max_len)
        suppress = [[base[0][0]], base[1][1:3], base[2][2:5]]
        cases = {
            "plain": {},
            "penalty": {"repetition_penalty": 1.6},
            "ngram2": {"no_repeat_ngram_size": 2},
            "suppress": {"suppress_sequences_ids": suppress},
            "all": {"repetition_penalty": 1.3, "no_repeat_ngram_size": 3,
                    "suppress_sequences_ids": suppress},
        }
        runs = {}
        for name, kw in cases.items():
            for beam in (1, 4):
                gen, scores = m.generate_processors_ids(prompts, beam_size=beam, max_len=max_len, **kw)
                runs["%s_beam%d" % (name, beam)] = {"options": kw, "beam": beam, "generated": gen,
                                                    "scores": scores.tolist()}
        out[key + suffix] = {"prompts": prompts, "max_len": max_len, "runs": runs}
        m.close()
        print(ct, {k: sum(a != b for a, b in zip(v["generated"], runs["plain_beam%d" % v["beam"]]["generated"]))
                   for k, v in runs.items()}, [len(g) for g in runs["all_beam4"]["generated"]])
    with open(os.path.join(HERE, "generator_processors_golden.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
