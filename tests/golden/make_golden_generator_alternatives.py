"""Dev-time script (needs /root/reference and oracle/_ref built): golden outputs of the REAL
reference Generator::generate_batch with return_alternatives (decode_alternatives,
src/decoding.cc:1122-1290, reached from DecoderReplica::run_generation,
src/models/language_model.cc:241-245) -> generator_alternatives_golden.json.

Prompts of ONE length (the fused engine's envelope for this mode: with ragged prompts the
reference consumes the remainder of a longer prompt in a second, per-example sequence pass), plus a
ragged set kept as a record of what the reference returns there (its sequences then begin with the
rest of the prompt).

  python tests/golden/make_golden_generator_alternatives.py
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
    rng = np.random.default_rng(cfg["data_seed"] + 500)
    vocab = cfg["model"]["vocab_size"]
    equal = [rng.integers(4, vocab, 9).tolist() for _ in range(5)]
    one = [rng.integers(4, vocab, 1).tolist() for _ in range(3)]          # start token only
    ragged = [rng.integers(4, vocab, n).tolist() for n in (9, 6, 11)]
    out = {}
    for ct, suffix in (("float32", "_f32"), ("int8", "")):
        m = R.RefModel(path, ct, 1, 4)
        runs = {}
        cases = [
            ("equal_greedy_nh3", equal, dict(beam_size=1, num_hypotheses=3, max_len=24)),
            ("equal_beam2_nh3", equal, dict(beam_size=2, num_hypotheses=3, max_len=24)),
            # (no min_alternative_expansion_prob case: when fewer than num_hypotheses alternatives pass
            # the threshold the reference's Generator crashes - SIGSEGV in the state resize of
            # src/decoding.cc:1226-1236 with a decoder-only state - on this build; the Translator
            # form of the option is pinned by make_golden_decoding_features.py)
            ("equal_greedy_nh2_short", equal, dict(beam_size=1, num_hypotheses=2, max_len=10)),
            ("equal_greedy_nh2_at_limit", equal, dict(beam_size=1, num_hypotheses=2, max_len=9)),
            ("equal_beam3_minlen", equal, dict(beam_size=3, num_hypotheses=2, max_len=24, min_len=16)),
            ("one_token_greedy_nh3", one, dict(beam_size=1, num_hypotheses=3, max_len=12)),
            ("ragged_greedy_nh2", ragged, dict(beam_size=1, num_hypotheses=2, max_len=24)),
        ]
        for name, prompts, kw in cases:
            seqs, scores = m.generate_ex_ids(prompts, return_alternatives=True, **kw)
            runs[name] = {"prompts": prompts, "options": kw, "sequences": seqs, "scores": scores}
            print(ct, name, [[len(s) for s in q] for q in seqs])
        out[key + suffix] = runs
        m.close()
    with open(os.path.join(HERE, "generator_alternatives_golden.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
