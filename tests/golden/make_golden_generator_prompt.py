"""Dev-time script (needs /root/reference and oracle/_ref built): golden outputs of the REAL
reference Generator::generate_batch with include_prompt_in_result = True (its default: the prompt
is decoded step by step as a RETURNED hard prefix - DecodingOptions::return_prefix - so it counts
towards max_length / min_length, forced tokens score 0 and the start token is put in front of the
sequences unless it is the BOS token; src/models/language_model.cc:150-282, src/decoding.cc:21-63,
357-408) -> tests/golden/generator_prompt_golden.json.

  python tests/golden/make_golden_generator_prompt.py
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

LENGTHS = {"lm_tiny": [6, 3, 5, 6, 1], "lm_small": [16, 11, 16, 7, 13, 9]}
BOS = 1  # model_builder's vocabulary: <blank> <s> </s> <unk> tok4 ...
STATIC = 5  # tokens of the static prompt


def main():
    assert R.available(), "build oracle/_ref first (make -C oracle ref)"
    out = {}
    for key, cfg in GC.SYNTHETIC_LM.items():
        path = "/tmp/ct2x_golden_%s" % key
        if not os.path.exists(os.path.join(path, "model.bin")):
            mb.build_decoder_lm(path, **cfg["model"])
        rng = np.random.default_rng(cfg["data_seed"] + 300)
        vocab = cfg["model"]["vocab_size"]
        prompts = [rng.integers(4, vocab, n).tolist() for n in LENGTHS[key]]
        prompts[0][0] = BOS  # (a BOS start token is NOT forwarded to the output)
        longest = max(LENGTHS[key])
        room = longest + cfg["max_len"] // 2
        rng_static = np.random.default_rng(cfg["data_seed"] + 400)
        static = None
        for ct, suffix in (("float32", "_f32"), ("int8", "")):
            m = R.RefModel(path, ct, 1, 4)
            runs = {}
            for name, beam, max_len, min_len in (
                    ("greedy", 1, room, 0),
                    ("greedy_forced", 1, room, room),
                    ("greedy_short", 1, longest - 2, 0),  # max_length ends inside the longest prompts
                    ("beam4", 4, room, 0),
                    ("beam4_min", 4, room, longest + 2)):
                gen, scores = m.generate_ids_with_prompt(prompts, beam_size=beam, max_len=max_len,
                                                         min_len=min_len, return_scores=True)
                runs[name] = {"beam": beam, "max_len": max_len, "min_len": min_len, "generated": gen,
                              "scores": scores.tolist()}
            # GenerationOptions::static_prompt (language_model.cc:184-206): a shared head of STATIC tokens in
            # front of every prompt - forwarded in one pass, never returned or counted - with the prompt in
            # the result (the reference's default) and without
            static = rng_static.integers(4, vocab, STATIC).tolist() if suffix == "_f32" else static
            sprompts = [static + p for p in prompts]
            sruns = {}
            for name, include, beam, max_len, min_len in (
                    ("with_prompt_greedy", True, 1, room, 0),
                    ("with_prompt_beam4", True, 4, room, longest + 2),
                    ("hidden_greedy", False, 1, cfg["max_len"], 0),
                    ("hidden_beam4", False, 4, cfg["max_len"], 0)):
                gen, scores = m.generate_ids_static(sprompts, STATIC, include, beam_size=beam, max_len=max_len,
                                                    min_len=min_len)
                sruns[name] = {"include_prompt": include, "beam": beam, "max_len": max_len, "min_len": min_len,
                               "generated": gen, "scores": scores.tolist()}
            out[key + suffix] = {"prompts": prompts, "runs": runs, "compute_type": m.compute_type(),
                                 "static": {"length": STATIC, "prompts": sprompts, "runs": sruns}}
            m.close()
    with open(os.path.join(HERE, "generator_prompt_golden.json"), "w") as f:
        json.dump(out, f)
    print({k: {n: [len(g) for g in r["generated"]] for n, r in v["runs"].items()} for k, v in out.items()})


if __name__ == "__main__":
    main()
