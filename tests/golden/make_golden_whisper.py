"""Generates tests/golden/whisper_golden.{json,npz}: the reference's own models::Whisper
(src/models/whisper.cc) on Device::CPU, float32, through oracle/_ref, on the synthetic Whisper
model of tests/whisper_cases.py (rebuilt deterministically by model_builder.build_whisper).

    python tests/golden/make_golden_whisper.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _refshim as R  # noqa: E402
import whisper_cases as WC  # noqa: E402


def main():
    out_json, out_npz = {}, {}
    for quant, compute in (("float32", "float32"), ("int8", "int8")):
        path = WC.build("/tmp/ct2x_whisper_%s" % quant, quant)
        w = R.RefWhisper(path, compute_type=compute)
        res = WC.run_cases(w)
        res["info"] = w.info()
        w.close()
        for k, v in res.items():
            if k.endswith("@"):
                out_npz[quant + "." + k[:-1]] = v
            else:
                out_json.setdefault(quant, {})[k] = v
    with open(os.path.join(HERE, "whisper_golden.json"), "w") as f:
        json.dump(out_json, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "whisper_golden.npz"), **out_npz)
    for quant in out_json:
        for k, v in out_json[quant].items():
            if k.startswith("generate."):
                print(quant, k, [h[0] for h in v["hypotheses"]])


if __name__ == "__main__":
    main()
