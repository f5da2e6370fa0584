"""Dev-time script (needs /root/reference and oracle/_ref built): the REAL reference
Translator::translate_batch (Device::CPU, int8) on the Transformer-base fixture of the metric with
MORE than 32 sentences - the batches the decoder-stack launch serves with several rows per
workgroup cluster (decoder_cluster_beam_kernel with beam 1: 33 .. 128 rows) - greedy, ragged source
lengths, free end of sentence -> tests/golden/many_rows_golden.json.

  python tests/golden/make_golden_many_rows.py
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

CASES = {"rows64": dict(rows=64, lo=8, hi=33, max_len=20, min_len=20),
         "rows100_ragged_free_end": dict(rows=100, lo=5, hi=40, max_len=16, min_len=1)}


def main():
    assert R.available(), "build oracle/_ref first (make -C oracle ref)"
    path = "/tmp/ct2x_golden_tb"
    if not os.path.exists(os.path.join(path, "model.bin")):
        mb.build_transformer(path, **GC.SYNTHETIC["tb"]["model"])
    m = R.RefModel(path, "int8", 1, 8)
    out = {"model": "tb", "compute_type": m.compute_type(), "cases": {}}
    for name, c in CASES.items():
        rng = np.random.default_rng(len(name) * 1000 + c["rows"])
        src = [rng.integers(4, 32000, int(rng.integers(c["lo"], c["hi"]))).tolist() for _ in range(c["rows"])]
        hyps, scores = m.translate_ids(src, beam_size=1, max_len=c["max_len"], min_len=c["min_len"],
                                       return_scores=True)
        out["cases"][name] = {"sources": src, "max_len": c["max_len"], "min_len": c["min_len"],
                              "hypotheses": [h[0] for h in hyps], "scores": scores[:, 0].tolist()}
        print(name, "lengths", sorted(set(len(h[0]) for h in hyps)))
    m.close()
    with open(os.path.join(HERE, "many_rows_golden.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
