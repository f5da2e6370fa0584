#!/usr/bin/env python
"""What dropping finished rows buys where a decode step's cost follows the row count: a decoder-only
model (8 layers, d 2048, FFN 8192, 16 heads, vocabulary 32000, int8), 256 prompts, greedy, max_length
96 counted from each prompt's end - half of the prompts are 8 tokens long, half 104, so half of the
rows are finished 96 steps before the others (the ragged hard-prefix path: a row's last step is
prefix length + max_length). CT2X_COMPACT_MIN_ROWS is read once per process:

  CT2X_COMPACT_MIN_ROWS=0 python tools/bench_compaction.py      # static batch
  python tools/bench_compaction.py                              # default policy (>= 64 rows)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench_compaction.py needs an MI355X (no CPU fallback)")
    import ctranslate2_amd as ct2
    from ctranslate2_amd import model_builder as mb
    d = "/tmp/ct2x_bench_compaction"
    if not os.path.exists(os.path.join(d, "model.bin")):
        mb.build_decoder_lm(d, num_layers=8, d_model=2048, ffn_dim=8192, num_heads=16, num_heads_kv=16,
                            vocab_size=32000, direct_int8=True)
    g = ct2.Translator(d, device="cuda", compute_type="int8")
    rng = np.random.default_rng(3)
    rows = int(os.environ.get("BENCH_ROWS", "256"))
    prompts = [rng.integers(4, 32000, 8 if i % 2 else 104).tolist() for i in range(rows)]

    def one():
        ids, _ = g.generate_ids(prompts, max_length=96, min_length=96)
        return ids

    first = one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        out = one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    st = g.last_stats()
    import hashlib
    digest = hashlib.sha256(json.dumps(out).encode()).hexdigest()[:16]
    print(json.dumps({"setting": os.environ.get("CT2X_COMPACT_MIN_ROWS", "default (64)"), "rows": rows,
                      "ms_per_call": dt * 1e3, "decode_ms": st["decode_ms"], "decode_steps": st["decode_steps"],
                      "compactions": st["compactions"], "tokens": sum(len(o) for o in out),
                      "same_as_first_call": out == first, "tokens_sha256_16": digest}))


if __name__ == "__main__":
    main()
