"""Golden rotary tables from the REAL reference (oracle/_ref, RotaryEmbeddings::initialize,
src/layers/attention_layer.cc:254-330): tests/golden/rotary_golden.npz. Run here (needs
/root/reference built by `make -C oracle ref`); the fixture travels to the GPU box."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _refshim as R  # noqa: E402

CASES = {
    "plain64": dict(dim=64, positions=300),
    "plain128_base500k": dict(dim=128, positions=260, base=500000.0),
    "interleave32": dict(dim=32, positions=130, interleave=True),
    "linear64_x4": dict(dim=64, positions=200, scaling_type=0, scaling_factor=4.0),
    # Llama-3.1 style: factor 8, low 1, high 4, original context 8192, base 500000
    "llama3_128": dict(dim=128, positions=300, scaling_type=2, scaling_factor=8.0, base=500000.0,
                       low_freq_factor=1.0, high_freq_factor=4.0,
                       original_max_position_embeddings=8192),
    "llama3_64_small_ctx": dict(dim=64, positions=150, scaling_type=2, scaling_factor=4.0,
                                low_freq_factor=1.0, high_freq_factor=4.0,
                                original_max_position_embeddings=64),
}


def main():
    out = {"meta": np.frombuffer(json.dumps(CASES).encode(), np.uint8)}
    for name, kw in CASES.items():
        sin, cos = R.rotary_tables(**kw)
        out[name + "/sin"] = sin
        out[name + "/cos"] = cos
    np.savez_compressed(os.path.join(HERE, "rotary_golden.npz"), **out)
    print("wrote", {k: v.shape for k, v in out.items() if k != "meta"})


if __name__ == "__main__":
    main()
