"""The greedy sampler fused into the vocabulary projection (GreedyFuse, gemm_s8.hip ARGMAX form): a
greedy step that returns no scores takes the arg-max of the raw logits (src/decoding.cc:874-880),
so the projection's workgroups max-combine packed (value, column) keys per row and the last
workgroup finishes the step - same ids as greedy_sample_kernel, one launch less, no logits in
memory. The switch is read once per process (CT2X_FUSED_ARGMAX), so both settings run in their
own interpreter; the launch count of a decode step shows which path ran."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))

SCRIPT = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(here)r)
import ctranslate2_amd as ct2
from ctranslate2_amd import model_builder as mb
out = {}
V = 9000
d = "/tmp/ct2x_fused_argmax_tb"
if not os.path.exists(os.path.join(d, "model.bin")):
    mb.build_transformer(d, num_encoder_layers=2, num_decoder_layers=2, d_model=256, ffn_dim=512,
                         num_heads=4, vocab_size=V, seed=3)
t = ct2.Translator(d, device="cuda", compute_type="int8")
rng = np.random.default_rng(1)
def sources(n):
    return [rng.integers(4, V, int(rng.integers(5, 14))).tolist() for _ in range(n)]
# one and two 16-row tiles per workgroup, a ragged last tile, a single row
for rows in (1, 7, 16, 17, 32):
    src = sources(rows)
    hyps, _ = t.translate_ids(src, beam_size=1, max_decoding_length=20, min_decoding_length=4)
    out["plain_%%d" %% rows] = hyps
    out["plain_%%d_kernels" %% rows] = t.last_stats()["kernels_per_step"]
src = sources(12)
# the end token stays masked below the minimum length; a hard target prefix forces the first steps
hyps, _ = t.translate_ids(src, beam_size=1, max_decoding_length=24, min_decoding_length=24,
                          return_end_token=True)
out["min_length"] = hyps
prefix = [h[0][: i %% 5] for i, h in enumerate(hyps)]
hyps, _ = t.translate_ids(src, beam_size=1, max_decoding_length=24, min_decoding_length=6,
                          target_prefix_ids=prefix, use_graph=False)
out["prefix_eager"] = hyps
hyps, _ = t.translate_ids(src, beam_size=1, max_decoding_length=24, min_decoding_length=6,
                          target_prefix_ids=prefix)
out["prefix"] = hyps
# what the fused form does not serve keeps the sampler launch: scores, logits processors, > 32 rows
hyps, scores = t.translate_ids(src, beam_size=1, max_decoding_length=12, return_scores=True)
out["scores"] = [hyps, [float(s) for s in scores[:, 0]]]
out["scores_kernels"] = t.last_stats()["kernels_per_step"]
hyps, _ = t.translate_ids(src, beam_size=1, max_decoding_length=12, repetition_penalty=1.2)
out["processors"] = hyps
out["processors_kernels"] = t.last_stats()["kernels_per_step"]
hyps, _ = t.translate_ids(sources(40), beam_size=1, max_decoding_length=12)
out["rows40"] = hyps
out["rows40_kernels"] = t.last_stats()["kernels_per_step"]
# sessions that drop their finished rows (64 -> <= 32 rows) switch to the fused form on the way
big = sources(70)
hyps, _ = t.translate_ids(big, beam_size=1, max_decoding_length=40)
out["compaction"] = hyps
# the callback path examines every step
calls = []
hyps, _ = t.translate_ids(src[:3], beam_size=1, max_decoding_length=10,
                          callback=lambda r: calls.append((r.batch_id, r.step, r.token_id)) and False)
out["callback"] = [hyps, sorted(calls)]
# decoder-only generation: ragged prompts are hidden hard prefixes that also push the minimum
# length out (apply_min_length with return_prefix = false)
d = "/tmp/ct2x_fused_argmax_lm"
if not os.path.exists(os.path.join(d, "model.bin")):
    mb.build_decoder_lm(d, num_layers=2, d_model=256, ffn_dim=512, num_heads=8, num_heads_kv=2,
                        vocab_size=V)
g = ct2.Translator(d, device="cuda", compute_type="int8")
prompts = [rng.integers(4, V, n).tolist() for n in (3, 9, 5, 12, 4, 3, 7)]
ids, _ = g.generate_ids(prompts, max_length=16, min_length=5, return_end_token=True)
out["lm"] = ids
out["lm_kernels"] = g.last_stats()["kernels_per_step"]
ids, _ = g.generate_ids(prompts, max_length=16, min_length=16, use_graph=False)
out["lm_eager"] = ids
print(json.dumps(out))
"""


def _run(fused):
    env = dict(os.environ)
    env["CT2X_FUSED_ARGMAX"] = "1" if fused else "0"
    env["CT2X_COMPACT_MIN_ROWS"] = "2"
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT, "here": HERE}], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
def test_fused_greedy_argmax_equals_the_sampler_launch(gpu):
    base = _run(False)
    fused = _run(True)
    for key in base:
        if key.endswith("_kernels"):
            continue
        assert fused[key] == base[key], key
    # the fused form really ran where it applies (one launch less per step) and only there
    for rows in (1, 7, 16, 17, 32):
        assert fused["plain_%d_kernels" % rows] == base["plain_%d_kernels" % rows] - 1, rows
    assert fused["lm_kernels"] == base["lm_kernels"] - 1
    for key in ("scores_kernels", "processors_kernels", "rows40_kernels"):
        assert fused[key] == base[key], key
    # the premise of the masked / forced cases
    assert all(len(h[0]) == 24 for h in base["min_length"])
