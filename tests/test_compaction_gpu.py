"""Finished-sentence removal of the greedy loop (Engine::compact_rows; the reference gathers the alive
rows out of every state tensor whenever a sentence finishes, src/decoding.cc:916-956): dropping the
finished rows must not change a single token or score of the rows that stay. The threshold is read
once per process (CT2X_COMPACT_MIN_ROWS; default 64 rows), so every setting runs in its own
interpreter: 0 = never, 2 = as early as the policy allows (at most half of the rows alive).
The reference-golden suites run once more under the forced setting by `tools/gpu_r4.sh compaction`."""
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
import golden_cases as GC
out = {}

# (a) the reference's int8 transliteration fixture: natural EOS at different steps, with scores,
#     a repetition penalty + n-gram blocking (device-side processors read the repacked history) and
#     a ragged hard target prefix
t = ct2.Translator(os.path.join(%(here)r, "golden", "models", "aren-transliteration-i8"), device="cuda",
                   compute_type="int8")
src = [s for s in GC.AREN_SOURCES] * 3
src = src + [s[: max(1, len(s) // 2)] for s in GC.AREN_SOURCES] + [s + s for s in GC.AREN_SOURCES]
res = t.translate_batch(src, beam_size=1, return_scores=True, max_decoding_length=30)
out["aren"] = [[r.hypotheses[0], float(r.scores[0])] for r in res]
out["aren_compactions"] = t.last_stats()["compactions"]
res = t.translate_batch(src, beam_size=1, return_scores=True, max_decoding_length=30,
                        repetition_penalty=1.3, no_repeat_ngram_size=2)
out["aren_lp"] = [[r.hypotheses[0], float(r.scores[0])] for r in res]
out["aren_lp_compactions"] = t.last_stats()["compactions"]
prefix = [r.hypotheses[0][: i %% 4] for i, r in enumerate(res)]
res = t.translate_batch(src, target_prefix=prefix, beam_size=1, return_scores=True, max_decoding_length=30)
out["aren_prefix"] = [[r.hypotheses[0], float(r.scores[0])] for r in res]
# beam search: finished batch entries leave with all their beams (ancestry tables repacked, caches
# untouched); two hypotheses per sentence, then with the device-side processors
res = t.translate_batch(src, beam_size=4, num_hypotheses=2, return_scores=True, max_decoding_length=30)
out["aren_beam"] = [[r.hypotheses, [float(x) for x in r.scores]] for r in res]
out["aren_beam_compactions"] = t.last_stats()["compactions"]
res = t.translate_batch(src, beam_size=3, return_scores=True, max_decoding_length=30, length_penalty=0.6,
                        repetition_penalty=1.2, no_repeat_ngram_size=2)
out["aren_beam_lp"] = [[r.hypotheses, [float(x) for x in r.scores]] for r in res]
res = t.translate_batch(src, target_prefix=prefix, beam_size=2, return_scores=True, max_decoding_length=30)
out["aren_beam_prefix"] = [[r.hypotheses, [float(x) for x in r.scores]] for r in res]
# random sampling draws from a hash of (seed, step, ROW): such sessions keep their rows, so a seed
# gives the same samples under either setting
res = t.translate_batch(src, beam_size=1, sampling_topk=4, seed=11, return_scores=True, max_decoding_length=30)
out["aren_sampling"] = [[r.hypotheses[0], float(r.scores[0])] for r in res]
out["aren_sampling_compactions"] = t.last_stats()["compactions"]

# (b) decoder-only generation, ragged prompts as hard prefixes: the rows' last step differs, so they
#     finish at different steps without any EOS
d = "/tmp/ct2x_compaction_lm"
if not os.path.exists(os.path.join(d, "model.bin")):
    mb.build_decoder_lm(d, num_layers=2, d_model=256, ffn_dim=512, num_heads=8, num_heads_kv=2,
                        vocab_size=1000)
g = ct2.Translator(d, device="cuda", compute_type="int8")
rng = np.random.default_rng(7)
prompts = [rng.integers(4, 1000, n).tolist() for n in (3, 30, 5, 28, 4, 31, 6, 2, 29, 3, 27, 5)]
ids, scores = g.generate_ids(prompts, max_length=34, return_scores=True)
out["lm"] = [ids, [float(s) for s in scores]]
out["lm_compactions"] = g.last_stats()["compactions"]
ids, scores = g.generate_ids(prompts, max_length=34, return_scores=True, repetition_penalty=1.3,
                             no_repeat_ngram_size=2, suppress_sequences_ids=[[7], [11, 12]])
out["lm_lp"] = [ids, [float(s) for s in scores]]
out["lm_lp_compactions"] = g.last_stats()["compactions"]
ids, scores = g.generate_ids(prompts, max_length=34, return_scores=True, beam_size=3)
out["lm_beam"] = [ids, [float(s) for s in scores]]
out["lm_beam_compactions"] = g.last_stats()["compactions"]
print(json.dumps(out))
"""


def _run(min_rows):
    env = dict(os.environ)
    env["CT2X_COMPACT_MIN_ROWS"] = str(min_rows)
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT, "here": HERE}], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
def test_dropping_finished_rows_changes_nothing(gpu):
    base = _run(0)
    early = _run(2)
    assert base["aren_compactions"] == 0 and base["lm_compactions"] == 0
    # the forced setting really did drop rows
    assert early["aren_compactions"] >= 1
    assert early["lm_compactions"] >= 1 and early["lm_lp_compactions"] >= 1
    assert early["aren_beam_compactions"] >= 1 and early["lm_beam_compactions"] >= 1
    assert early["aren_sampling_compactions"] == 0
    for key in ("aren", "aren_lp", "aren_prefix", "lm", "lm_lp", "aren_beam", "aren_beam_lp",
                "aren_beam_prefix", "lm_beam", "aren_sampling"):
        assert early[key] == base[key], key
    # sentences of different lengths finish at different steps in this batch (the premise)
    assert len({len(h) for h, _ in base["aren"]}) >= 3
