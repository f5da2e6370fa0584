"""Dev-time script (needs /root/reference and oracle/_ref built): golden outputs of the REAL
reference Generator::generate_batch for prompts of DIFFERENT lengths (the shortest prompt minus
one token is forwarded in sequence mode, the rest of each row is a hard decoding prefix;
src/models/language_model.cc:218-238) -> tests/golden/ragged_golden.json.

  python tests/golden/make_golden_ragged.py
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

LENGTHS = {"lm_tiny": [6, 3, 5, 6], "lm_small": [16, 11, 16, 7, 13, 9]}


def main():
    assert R.available(), "build oracle/_ref first (make -C oracle ref)"
    out = {}
    for key, cfg in GC.SYNTHETIC_LM.items():
        path = "/tmp/ct2x_golden_%s" % key
        if not os.path.exists(os.path.join(path, "model.bin")):
            mb.build_decoder_lm(path, **cfg["model"])
        rng = np.random.default_rng(cfg["data_seed"] + 100)
        vocab = cfg["model"]["vocab_size"]
        prompts = [rng.integers(4, vocab, n).tolist() for n in LENGTHS[key]]
        rng_eq = np.random.default_rng(cfg["data_seed"] + 200)
        eq_prompts = None
        for ct, suffix in (("float32", "_f32"), ("int8", "")):
            m = R.RefModel(path, ct, 1, 4)
            runs = {}
            for name, min_len in (("forced", cfg["max_len"]), ("free", 0)):
                gen, scores = m.generate_ids(prompts, max_len=cfg["max_len"], min_len=min_len,
                                             return_scores=True)
                runs[name] = {"min_len": min_len, "generated": gen, "scores": scores.tolist()}
            # beam search over equal-length prompts (Generator beam_size = 4, length_penalty = 1,
            # GenerationOptions defaults, include/ctranslate2/generation.h)
            eq_prompts = [rng_eq.integers(4, vocab, cfg["prompt_len"]).tolist() for _ in range(cfg["batch"])] \
                if suffix == "_f32" else eq_prompts
            gen, scores = m.generate_ids(eq_prompts, beam_size=4, max_len=cfg["max_len"],
                                         min_len=cfg["max_len"] // 2, return_scores=True)
            # beam search over the RAGGED prompts: hard prefix in BeamSearch::search with
            # return_prefix = false (hidden prefix; max length counts from its end). min_length = 0
            # on purpose: with a hidden prefix the reference's apply_min_length disables EOS in the
            # flattened rows 0..batch-1 of the [batch * beam, vocab] logits (disable_tokens.add(i,
            # end_id) with i a BATCH index, src/decoding.cc:392-402), i.e. in the wrong rows.
            rgen, rscores = m.generate_ids(prompts, beam_size=4, max_len=cfg["max_len"],
                                           min_len=0, return_scores=True)
            # ... and WITH min_length > 0 (round 6): the quirk above is reproduced on the device
            # (beam_logprobs_kernel: row i < live entries masked by entry i's threshold, the other
            # rows not at all, from the second step on)
            qgen, qscores = m.generate_ids(prompts, beam_size=4, max_len=cfg["max_len"],
                                           min_len=cfg["max_len"] // 2, return_scores=True)
            out[key + suffix] = {"prompts": prompts, "max_len": cfg["max_len"], "runs": runs,
                                 "compute_type": m.compute_type(),
                                 "beam4": {"prompts": eq_prompts, "min_len": cfg["max_len"] // 2,
                                           "generated": gen, "scores": scores.tolist()},
                                 "beam4_ragged": {"min_len": 0, "generated": rgen,
                                                  "scores": rscores.tolist()},
                                 "beam4_ragged_minlen": {"min_len": cfg["max_len"] // 2,
                                                         "generated": qgen,
                                                         "scores": qscores.tolist()}}
            m.close()
    with open(os.path.join(HERE, "ragged_golden.json"), "w") as f:
        json.dump(out, f)
    print({k: [len(g) for g in v["runs"]["free"]["generated"]] for k, v in out.items()})


if __name__ == "__main__":
    main()
