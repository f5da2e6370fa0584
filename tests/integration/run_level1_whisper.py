"""Level-1 Whisper runner (own process, like run_level1.py): the reference's OWN models::Whisper
(src/models/whisper.cc, src/layers/whisper.cc, decoding.cc, every ops front-end) compiled with
-DCT2_WITH_CUDA and linked against the stub units + libct2_mi355.so, run on Device::CUDA = the
MI355X kernels over every case of tests/whisper_cases.py.

  python tests/integration/run_level1_whisper.py <model_dir> <compute_type> <out.json> <out.npz>
"""
import json
import os
import sys

import numpy as np

os.environ["CT2REF_VARIANT"] = "mi355"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import _refshim as R  # noqa: E402
import whisper_cases as WC  # noqa: E402


def main():
    model_dir, compute_type, out_json, out_npz = sys.argv[1:5]
    w = R.RefWhisper(model_dir, compute_type=compute_type, device="cuda")
    res = WC.run_cases(w)
    res["info"] = w.info()
    w.close()
    np.savez(out_npz, **{k[:-1]: v for k, v in res.items() if k.endswith("@")})
    with open(out_json, "w") as f:
        json.dump({k: v for k, v in res.items() if not k.endswith("@")}, f)
    print("ok")


if __name__ == "__main__":
    main()
