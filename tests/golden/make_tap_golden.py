"""Dev-time script (needs oracle/_ref built here): op-level taps of the REAL reference CPU
library on the teacher-forced runs of tests/golden/model_golden.json, reduced to one CRC per row
of every int8 Dense input (+ its row scale) - tests/_taps.py explains the events and their
names. Written to tests/golden/tap_golden.npz together with the reference's sinusoidal position
tables (MKL vsSin / vsCos) for the depths in use, which tests/test_parity_taps_gpu.py injects to
separate that residual from everything else.

Run:  python tests/golden/make_tap_golden.py
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
import _taps as T  # noqa: E402
from ctranslate2_amd import model_builder as mb  # noqa: E402
import golden_cases as GC  # noqa: E402

SHAPES = ["aren-transliteration-i8", "tiny", "small", "tb"]


def model_dir(key):
    if key.startswith("aren"):
        return os.path.join(HERE, "models", key)
    d = "/tmp/ct2x_golden_%s" % key
    if not os.path.exists(os.path.join(d, "model.bin")):
        mb.build_transformer(d, **GC.SYNTHETIC[key]["model"])
    return d


def vocab_size(key):
    if key.startswith("aren"):
        with open(os.path.join(model_dir(key), "target_vocabulary.txt"), encoding="utf-8") as f:
            tokens = [line.rstrip("\n") for line in f]
        return len(tokens) + (0 if "<unk>" in tokens else 1)
    return GC.SYNTHETIC[key]["model"]["vocab_size"]


def main():
    golden = json.load(open(os.path.join(HERE, "model_golden.json")))
    out = {}
    meta = {}
    for key in SHAPES:
        g = golden[key]
        tgt_in = np.array(g["teacher_forced"]["tgt_in"], np.int32)
        m = R.RefModel(model_dir(key), "int8", 1, 0)
        with R.taps() as t:
            logits = m.step_logits(g["sources"], tgt_in, vocab_size(key))
        # layer counts from the tap stream itself: Quantize calls before the first [B, 1, d] one
        quant = [a for k, a in t.records if k == "quantize.in"]
        first_dec = next(i for i, a in enumerate(quant) if a.ndim == 3 and a.shape[1] == 1)
        n_enc = first_dec // 4
        steps = tgt_in.shape[1]
        n_dec = (len(quant) - first_dec - steps) // (6 * steps + 1)
        ev = T.ref_events(t.records, n_enc, n_dec, steps)
        for name, f, q, s in ev:
            out["%s/%s" % (key, name)] = T.row_hashes(q, s)
        out["%s/logits_argmax" % key] = logits.argmax(-1).astype(np.int32)
        d_model = ev[0][1].shape[1]
        meta[key] = {"n_enc": n_enc, "n_dec": n_dec, "steps": steps, "d_model": d_model,
                     "events": len(ev)}
        pos = max(max(len(s) for s in g["sources"]), steps)
        out["postable/%d" % d_model] = R.sinusoidal_table(d_model, max(pos, 64))
        m.close()
        print(key, meta[key])
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "tap_golden.npz"), **out)
    print("wrote", os.path.getsize(os.path.join(HERE, "tap_golden.npz")), "bytes")


if __name__ == "__main__":
    main()
