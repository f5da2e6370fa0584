"""Dev-time script (needs /root/reference and oracle/_ref built): golden outputs of the REAL
reference CPU library for a decoder-only model at the Llama-3-8B GEOMETRY (d_model 4096, 32 query
heads on 8 KV heads, d_head 128, SwiGLU 14336, RMSNorm, RoPE base 5e5) - two layers, small
vocabulary, int8 weights. BASELINE.json configs[2] asks for token-for-token parity at this geometry;
the weights are synthetic (model_builder.py, fixed seed) and far too large to commit, so the GPU test
rebuilds the same model directory from the seed and compares with the outputs committed here.

Two prompt sets: 40 tokens (MKL scales the finished QK^T sum in the prefill) and 200 tokens (MKL
packs K: the packed copy carries 1 / sqrt(d_head); oracle/ct2_oracle.c orc_matmul).
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
    cfg = GC.LLAMA3_GEOMETRY
    d = "/tmp/ct2x_golden_lm_l3g"
    if not os.path.exists(os.path.join(d, "model.bin")):
        mb.build_decoder_lm(d, **cfg["model"])
    V = cfg["model"]["vocab_size"]
    m = R.RefModel(d, "int8", 1, 0)
    out = {"compute_type": m.compute_type(), "model": cfg["model"], "cases": {}}
    for name, case in cfg["cases"].items():
        rng = np.random.default_rng(case["data_seed"])
        prompts = [rng.integers(4, V, case["prompt_len"]).tolist() for _ in range(case["batch"])]
        gen, scores = m.generate_ids(prompts, max_len=case["max_len"], min_len=case["max_len"],
                                     return_scores=True)
        entry = {"prompts": prompts, "max_len": case["max_len"], "generated": gen,
                 "scores": scores.tolist()}
        if case.get("teacher_forced"):
            full = [p + g for p, g in zip(prompts, gen)]
            tgt_in = np.array([f[:-1] for f in full], np.int32)
            logits = m.step_logits([], tgt_in, V)  # [T, B, V]
            rng2 = np.random.default_rng(1234)
            cols = np.sort(rng2.choice(V, size=32, replace=False))
            top2 = np.sort(logits, axis=-1)[..., -2:][..., ::-1]
            entry["teacher_forced"] = {"tgt_in": tgt_in.tolist(), "argmax": logits.argmax(-1).tolist(),
                                       "top2": top2.tolist(), "cols": cols.tolist(),
                                       "values": logits[..., cols].tolist()}
        out["cases"][name] = entry
        print(name, "generated", gen[0][:8], "...")
    m.close()
    with open(os.path.join(HERE, "llama3_geometry_golden.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
