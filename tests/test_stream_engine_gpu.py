"""Engine-level check of the streaming decode GEMM (gemm_s8_stream.hip) and of the fragment-major
weight copies: a decoder-only model large enough to take that path (Llama-style: RMSNorm, RoPE, GQA,
gated feed-forward; weights of >= 8 MB per Dense) must generate the same tokens and scores as the
one-shot kernels on row-major weights. The switches are read once per process, so every setting
runs in its own interpreter. (The kernels themselves are pinned to the oracle in test_ops_gpu.py.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import ctranslate2_amd as ct2
from ctranslate2_amd import model_builder as mb
d = "/tmp/ct2x_stream_test_model"
if not os.path.exists(os.path.join(d, "model.bin")):
    mb.build_decoder_lm(d, num_layers=2, d_model=4096, ffn_dim=4096, num_heads=32, num_heads_kv=8,
                        vocab_size=8192, direct_int8=True)
gen = ct2.Translator(d, device="cuda", compute_type="int8")
rng = np.random.default_rng(5)
prompts = [rng.integers(4, 8192, n).tolist() for n in (9, 17, 5, 12, 17, 3)]
out, scores = gen.generate_ids(prompts, max_length=24, min_length=24, return_scores=True)
print(json.dumps({"tokens": out, "scores": [float(s) for s in scores],
                  "kernels_per_step": gen.last_stats()["kernels_per_step"]}))
"""


def _run(env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
def test_streaming_and_packed_weights_are_result_neutral(gpu):
    base = _run({"CT2X_STREAM_GEMM": "0"})
    stream_rowmajor = _run({"CT2X_STREAM_PACKED": "0"})
    packed = _run({})
    assert stream_rowmajor["tokens"] == base["tokens"] and stream_rowmajor["scores"] == base["scores"]
    assert packed["tokens"] == base["tokens"] and packed["scores"] == base["scores"]
    assert len(base["tokens"]) == 6 and all(len(t) == 24 for t in base["tokens"])


@pytest.mark.gpu
def test_streaming_prologues_are_result_neutral(gpu):
    """RMSNorm + Quantize and the element-wise Quantize as prologues of the streaming GEMM
    (gemm_s8_stream.hip AMODE 1 / 2; off by default - measured not faster): the engine
    with them switched on generates the same tokens and scores, with fewer launches per step."""
    base = _run({})
    both = _run({"CT2X_STREAM_PROLOGUE": "3"})
    norm_only = _run({"CT2X_STREAM_PROLOGUE": "1"})
    assert both["tokens"] == base["tokens"] and both["scores"] == base["scores"]
    assert norm_only["tokens"] == base["tokens"] and norm_only["scores"] == base["scores"]
    assert both["kernels_per_step"] < norm_only["kernels_per_step"] < base["kernels_per_step"]
