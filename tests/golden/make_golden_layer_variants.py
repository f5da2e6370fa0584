"""Dev-time script (needs /root/reference and oracle/_ref built): golden outputs of the REAL
reference CPU library for the decoder layer variants and Su rotary scaling
(tests/golden_cases.py LAYER_VARIANTS): greedy and beam-2 generation, and the teacher-forced logits
of every position. The models are rebuilt from their seeds by the GPU test.

    python tests/golden/make_golden_layer_variants.py
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
from ctranslate2_amd import model_builder as mb  # noqa: E402
import golden_cases as GC  # noqa: E402


def main():
    out, arrays = {}, {}
    case = GC.LAYER_VARIANT_CASE
    for name, model in GC.LAYER_VARIANTS.items():
        d = "/tmp/ct2x_golden_variant_%s" % name
        if not os.path.exists(os.path.join(d, "model.bin")):
            mb.build_decoder_lm(d, **model)
        V = model["vocab_size"]
        m = R.RefModel(d, "int8", 1, 0)
        rng = np.random.default_rng(case["data_seed"])
        prompts = [rng.integers(4, V, case["prompt_len"]).tolist() for _ in range(case["batch"])]
        gen, scores = m.generate_ids(prompts, max_len=case["max_len"], min_len=case["max_len"],
                                     return_scores=True)
        gen2, scores2 = m.generate_ids(prompts, beam_size=2, max_len=case["max_len"],
                                       min_len=case["max_len"], return_scores=True)
        full = [p + g for p, g in zip(prompts, gen)]
        tgt_in = np.array([f[:-1] for f in full], np.int32)
        logits = m.step_logits([], tgt_in, V)  # [T, B, V]
        out[name] = {"prompts": prompts, "generated": gen, "scores": scores.tolist(),
                     "beam2": gen2, "beam2_scores": scores2.tolist(), "tgt_in": tgt_in.tolist()}
        arrays[name + ".logits"] = np.asarray(logits, np.float32)
        m.close()
        print(name, gen[0])
    with open(os.path.join(HERE, "layer_variants_golden.json"), "w") as f:
        json.dump(out, f)
    np.savez_compressed(os.path.join(HERE, "layer_variants_golden.npz"), **arrays)


if __name__ == "__main__":
    main()
