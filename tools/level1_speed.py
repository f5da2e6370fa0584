#!/usr/bin/env python
"""Data point for DESIGN.md: the reference's OWN Translator running op by op on this backend
(Level 1: integration/libctranslate2_mi355.so, ~400 launches per decode step, a StorageView
allocation per op) on the metric's workload - Transformer-base int8, batch 32, 64 forced steps -
next to the fused engine (Level 2). Own process (tests/integration/run_level1.py conventions)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["CT2REF_VARIANT"] = "mi355"
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import _refshim as R  # noqa: E402
import golden_cases as GC  # noqa: E402
from ctranslate2_amd import model_builder as mb  # noqa: E402


def main():
    path = "/tmp/ct2x_test_tb"
    if not os.path.exists(os.path.join(path, "model.bin")):
        mb.build_transformer(path, **GC.SYNTHETIC["tb"]["model"])
    rng = np.random.default_rng(1)
    sources = [rng.integers(4, 32000, 32).tolist() for _ in range(32)]
    m = R.RefModel(path, "int8", 1, 0, device="cuda")
    kw = dict(beam_size=1, max_len=64, min_len=64, length_penalty=1.0, num_hypotheses=1,
              return_scores=False)
    m.translate_ids(sources, **kw)
    t0 = time.perf_counter()
    n = 0
    for _ in range(3):
        hyps, _ = m.translate_ids(sources, **kw)
        n += sum(len(h[0]) for h in hyps)
    dt = time.perf_counter() - t0
    import ctypes
    try:
        fn = ctypes.CDLL(None).ct2x_level15_served_calls
        fn.restype = ctypes.c_int64
        served = int(fn())
    except AttributeError:
        served = None
    print(json.dumps({"level1_tokens_per_s": n / dt, "ms_per_call": dt / 3 * 1e3,
                      "calls_served_by_the_fused_engine": served,
                      "CT2X_LEVEL1_FUSED": os.environ.get("CT2X_LEVEL1_FUSED", "1"),
                      "note": "reference Translator(Device::CUDA) on libct2_mi355.so through the stubs, "
                              "Transformer-base int8, batch 32, 64 forced steps"}))
    m.close()


if __name__ == "__main__":
    main()
