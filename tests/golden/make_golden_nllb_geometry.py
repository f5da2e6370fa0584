"""Dev-time script (needs /root/reference and oracle/_ref built): golden outputs of the REAL
reference CPU library for an encoder-decoder model at the NLLB-200 3.3B GEOMETRY (BASELINE.json
configs[3]: d_model 2048, 16 heads of 128, FFN 8192, ReLU, pre-norm, sinusoidal positions, shared
embeddings) - two layers each side, vocabulary 8192, weights STORED in float16 (drawn by
model_builder.py from a fixed seed; ~0.6 GB, rebuilt by the GPU test). The reference's CPU has no
float16 arithmetic: it loads the float16 weights exactly into float32 (`compute_type=float32`), so
these logits are the float32 evaluation of the SAME weights - the bar for the float16 / bfloat16
compute types of the backend is |logit - golden| < 1e-3 (north_star)."""
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
    cfg = GC.NLLB_GEOMETRY
    d = "/tmp/ct2x_golden_nllb_geom"
    if not os.path.exists(os.path.join(d, "model.bin")):
        mb.build_transformer(d, **cfg["model"])
    V = cfg["model"]["vocab_size"]
    rng = np.random.default_rng(cfg["data_seed"])
    sources = [rng.integers(4, V, n).tolist() for n in cfg["src_lengths"]]
    m = R.RefModel(d, "float32", 1, 0)
    hyps, scores = m.translate_ids(sources, beam_size=1, max_len=cfg["max_len"], min_len=cfg["max_len"],
                                   return_scores=True)
    greedy = [h[0] for h in hyps]
    T = cfg["logits_steps"]
    tgt_in = np.array([[1] + g[:T - 1] for g in greedy], np.int32)
    # equal-length sources only for the teacher-forced pass (step_logits takes a rectangular batch)
    L = max(len(s) for s in sources)
    full = [i for i, s in enumerate(sources) if len(s) == L]
    logits = m.step_logits([sources[i] for i in full], tgt_in[full], V)  # [T, B, V]
    rng2 = np.random.default_rng(1234)
    cols = np.sort(rng2.choice(V, size=64, replace=False))
    top2 = np.sort(logits, axis=-1)[..., -2:][..., ::-1]
    out = {"compute_type": m.compute_type(), "model": cfg["model"], "sources": sources,
           "max_len": cfg["max_len"], "greedy": greedy, "scores": scores.tolist(),
           "teacher_forced": {"rows": full, "tgt_in": tgt_in[full].tolist(), "argmax": logits.argmax(-1).tolist(),
                              "top2": top2.tolist(), "cols": cols.tolist(), "values": logits[..., cols].tolist(),
                              "logit_std": float(logits.std()), "logit_absmax": float(np.abs(logits).max())}}
    m.close()
    with open(os.path.join(HERE, "nllb_geometry_golden.json"), "w") as f:
        json.dump(out, f)
    print("logit std %.3f absmax %.3f" % (out["teacher_forced"]["logit_std"], out["teacher_forced"]["logit_absmax"]))


if __name__ == "__main__":
    main()
