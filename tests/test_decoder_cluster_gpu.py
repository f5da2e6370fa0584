"""The cluster-resident decoder stack (decoder_cluster.hip): every decoder layer of a greedy step
as ONE launch - a batch row owned by eight workgroups (one per head) that hand the row over through
memory-side atomics - against the launch-per-op step it replaces (CT2X_CLUSTER_STACK=0, its own
interpreter: the switch is read once per process). Same formulas in the same order, exact int32
sums: tokens, scores and teacher-forced logits must be the same BITS. The reference goldens at this
geometry (tests/test_translator_gpu.py "tb", tests/test_bench_path_gpu.py) run through the new
launch by default; this file pins the shapes they do not visit (1 .. 32 rows, ragged sources, a
target prefix, minimum lengths, session reuse, finished-row removal)."""
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
path = "/tmp/ct2x_test_tb"
if not os.path.exists(os.path.join(path, "model.bin")):
    mb.build_transformer(path, **GC.SYNTHETIC["tb"]["model"])
t = ct2.Translator(path, device="cuda", compute_type="int8")
rng = np.random.default_rng(11)
V = 32000
def sources(n, lo=5, hi=40):
    return [rng.integers(4, V, int(rng.integers(lo, hi))).tolist() for _ in range(n)]
for rows in (1, 2, 7, 16, 31, 32):
    src = sources(rows)
    hyps, scores = t.translate_ids(src, beam_size=1, max_decoding_length=24, min_decoding_length=3,
                                   return_scores=True)
    out["scores_%%d" %% rows] = [hyps, np.asarray(scores, np.float32).view(np.uint32).tolist()]
    out["scores_%%d_kernels" %% rows] = t.last_stats()["kernels_per_step"]
    hyps, _ = t.translate_ids(src, beam_size=1, max_decoding_length=24, min_decoding_length=3)
    out["plain_%%d" %% rows] = hyps
    out["plain_%%d_kernels" %% rows] = t.last_stats()["kernels_per_step"]
# 33 .. 128 rows: 2 - 4 independent rows per cluster (a ragged last cluster at 33 and 100 rows);
# beyond that the launch-per-op step stays
for rows in (33, 64, 100, 128, 130):
    src = sources(rows)
    hyps, scores = t.translate_ids(src, beam_size=1, max_decoding_length=12, min_decoding_length=3,
                                   return_scores=True)
    out["many_%%d" %% rows] = [hyps, np.asarray(scores, np.float32).view(np.uint32).tolist()]
    out["many_%%d_kernels" %% rows] = t.last_stats()["kernels_per_step"]
# eager launches as well as the graph replay; more than 64 keys in the self-attention cache
src = sources(9)
hyps, _ = t.translate_ids(src, beam_size=1, max_decoding_length=90, min_decoding_length=90, use_graph=False)
out["long_eager"] = hyps
hyps, _ = t.translate_ids(src, beam_size=1, max_decoding_length=90, min_decoding_length=90)
out["long_graph"] = hyps
# reused sessions: the same shape again and again, different sentences
fixed = [rng.integers(4, V, 20).tolist() for _ in range(12)]
other = [rng.integers(4, V, 20).tolist() for _ in range(12)]
seq = []
for call in range(4):
    hyps, _ = t.translate_ids(other if call == 1 else fixed, beam_size=1, max_decoding_length=16,
                              min_decoding_length=16)
    seq.append(hyps)
out["reuse"] = seq
# sources of more than 64 tokens: the encoder-decoder attention walks several key tiles (the first
# from the staged LDS tile, the rest from memory), greedy, several rows per cluster and beam search
long_src = sources(6, 70, 140)
hyps, scores = t.translate_ids(long_src, beam_size=1, max_decoding_length=10, return_scores=True)
out["long_source"] = [hyps, np.asarray(scores, np.float32).view(np.uint32).tolist()]
hyps, scores = t.translate_ids(sources(40, 60, 100), beam_size=1, max_decoding_length=8, return_scores=True)
out["long_source_many"] = [hyps, np.asarray(scores, np.float32).view(np.uint32).tolist()]
hyps, scores = t.translate_ids(long_src, beam_size=4, max_decoding_length=10, return_scores=True)
out["long_source_beam4"] = [hyps, np.asarray(scores, np.float32).view(np.uint32).tolist()]
# sessions of different row counts one after the other: the hand-over lines of a new session come
# from the allocator's pool with whatever an earlier session left in them
big = sources(32, 32, 33)
full, _ = t.translate_ids(big, beam_size=1, max_decoding_length=40, min_decoding_length=40)
half, _ = t.translate_ids(big[:16], beam_size=1, max_decoding_length=40, min_decoding_length=40)
again, _ = t.translate_ids(big, beam_size=1, max_decoding_length=40, min_decoding_length=40)
out["pool_reuse"] = [full, half, again]
# a hard target prefix and the end token (finished rows keep being stepped)
hyps, _ = t.translate_ids(src, beam_size=1, max_decoding_length=20, return_end_token=True)
out["eos"] = hyps
prefix = [h[0][: i %% 4] for i, h in enumerate(hyps)]
hyps, _ = t.translate_ids(src, beam_size=1, max_decoding_length=20, target_prefix_ids=prefix)
out["prefix"] = hyps
# teacher-forced logits of every step (Engine::step without the fused sampler)
tgt = rng.integers(4, V, (6, 10)).astype(np.int32)
logits = t.step_logits(sources(6), tgt)
out["logits"] = np.ascontiguousarray(logits[:, :, ::97], np.float32).view(np.uint32).tolist()
# beam search: a cluster per batch entry with its rows, through the ancestry table
for beam, rows, max_len, min_len in ((4, 4, 12, 1), (2, 7, 20, 5), (3, 5, 20, 1), (4, 32, 30, 30), (4, 9, 80, 80),
                                     (4, 1, 16, 1)):
    bs = sources(rows)
    hyps, scores = t.translate_ids(bs, beam_size=beam, max_decoding_length=max_len, min_decoding_length=min_len,
                                   num_hypotheses=min(beam, 2), return_scores=True, length_penalty=1.0)
    out["beam%%d_rows%%d" %% (beam, rows)] = [hyps, np.asarray(scores, np.float32).view(np.uint32).tolist()]
    out["beam%%d_rows%%d_kernels" %% (beam, rows)] = t.last_stats()["kernels_per_step"]
# ... reused beam sessions, and the eager launches
fixed = [rng.integers(4, V, 20).tolist() for _ in range(8)]
seq = []
for call in range(3):
    hyps, _ = t.translate_ids(fixed, beam_size=4, max_decoding_length=16, min_decoding_length=16)
    seq.append(hyps)
hyps, _ = t.translate_ids(fixed, beam_size=4, max_decoding_length=16, min_decoding_length=16, use_graph=False)
seq.append(hyps)
out["beam_reuse"] = seq
print(json.dumps(out))
"""


def _run(cluster, handover=None):
    env = dict(os.environ)
    env["CT2X_CLUSTER_STACK"] = "1" if cluster else "0"
    if handover:
        env["CT2X_CLUSTER_HANDOVER"] = handover
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT, "here": HERE}], env=env,
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
def test_cluster_stack_equals_launch_per_op(gpu):
    base = _run(False)
    fused = _run(True)
    for key in base:
        if key.endswith("_kernels"):
            continue
        assert fused[key] == base[key], key
    for rows in (1, 2, 7, 16, 31, 32):
        # embedding + 8 launches per layer x 6 layers + the output norm became one launch
        # ... and from 8 rows on the vocabulary projection + greedy arg-max ride in the launch's tail
        # (round 6, ClusterProjArgs: 64 workgroups x <= 4 passes of 128 columns cover 32000): ONE launch
        assert fused["plain_%d_kernels" % rows] == base["plain_%d_kernels" % rows] - (50 if rows >= 8 else 49), rows
        if rows >= 8:
            assert fused["plain_%d_kernels" % rows] == 1, rows
        assert fused["scores_%d_kernels" % rows] == base["scores_%d_kernels" % rows] - 49, rows
    for rows in (33, 64, 100, 128):
        assert fused["many_%d_kernels" % rows] <= base["many_%d_kernels" % rows] - 40, rows
    assert fused["many_130_kernels"] == base["many_130_kernels"]
    for key in base:
        if key.startswith("beam") and key.endswith("_kernels"):
            assert fused[key] <= base[key] - 49, key
    assert fused["beam_reuse"][0] == fused["beam_reuse"][1] == fused["beam_reuse"][2] == fused["beam_reuse"][3]
    assert fused["long_eager"] == fused["long_graph"]
    assert fused["reuse"][0] == fused["reuse"][2] == fused["reuse"][3]
    assert fused["reuse"][0] != fused["reuse"][1]
    assert fused["pool_reuse"][1] == fused["pool_reuse"][0][:16] and fused["pool_reuse"][2] == fused["pool_reuse"][0]
    # the arrival-counter form of the hand-over (the default is the tagged one)
    counter = _run(True, "counter")
    for key in base:
        if not key.endswith("_kernels"):
            assert counter[key] == base[key], key


# ---- round 6: the launch is safe to ship -----------------------------------------------------
# Several workers per device is a supported deployment of the reference
# (include/ctranslate2/replica_pool.h:339-359): the decoder-stack launch, whose workgroups wait for
# each other, must not turn a second handle / process on the GPU into errors or wrong tokens.

WORKER = r"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(here)r)
import ctranslate2_amd as ct2
from ctranslate2_amd import model_builder as mb
import golden_cases as GC
path = "/tmp/ct2x_test_tb"
if not os.path.exists(os.path.join(path, "model.bin")):
    mb.build_transformer(path, **GC.SYNTHETIC["tb"]["model"])
t = ct2.Translator(path, device="cuda", compute_type="int8")
rng = np.random.default_rng(3)
src = [rng.integers(4, 32000, 24).tolist() for _ in range(16)]
kw = dict(beam_size=1, max_decoding_length=48, min_decoding_length=48)
mode = sys.argv[1]
out = {}
if mode == "loop":       # translate for a while next to another process doing the same
    first, _ = t.translate_ids(src, **kw)
    open(sys.argv[2], "w").write("up")
    while not os.path.exists(sys.argv[3]):   # wait for the sibling to be up as well
        time.sleep(0.01)
    same, fused_calls, calls = True, 0, 0
    end = time.time() + 6.0
    while time.time() < end:
        hyps, _ = t.translate_ids(src, **kw)
        same = same and hyps == first
        calls += 1
        fused_calls += int(t.last_stats()["kernels_per_step"] <= 3)
    out = {"hyps": first, "same": same, "calls": calls, "fused_calls": fused_calls,
           "fallbacks": t.cluster_fallbacks()}
elif mode == "inject":   # a starved cluster: the call is decoded again on the launch-per-op step
    base, _ = t.translate_ids(src, **kw)
    k_fused = t.last_stats()["kernels_per_step"]
    t.debug_inject_cluster_timeout()
    again, _ = t.translate_ids(src, **kw)
    k_after = t.last_stats()["kernels_per_step"]
    later, _ = t.translate_ids(src, **kw)
    b4, s4 = t.translate_ids(src[:4], beam_size=4, max_decoding_length=12, return_scores=True)
    out = {"base": base, "again": again, "later": later, "k_fused": k_fused, "k_after": k_after,
           "fallbacks": t.cluster_fallbacks(), "beam": b4}
elif mode == "compact":   # 40 rows whose finished ones leave: the session crosses from the several-rows
    # form of the launch (per-cluster lines and counters) to the one-row form (per-row) at <= 32 rows
    rng = np.random.default_rng(17)
    many = [rng.integers(4, 32000, int(rng.integers(6, 30))).tolist() for _ in range(40)]
    long_kw = dict(beam_size=1, max_decoding_length=40, min_decoding_length=1, return_scores=True)
    free, _ = t.translate_ids(many, **long_kw)
    # an end token that stops about half of the rows early: the most frequent token of steps 2 .. 9
    from collections import Counter
    cnt = Counter(tok for h in free for tok in set(h[0][2:10]))
    end_tok, hits = cnt.most_common(1)[0]
    hyps, scores = t.translate_ids(many, end_token_id=int(end_tok), **long_kw)
    st = t.last_stats()
    out = {"end_token": int(end_tok), "rows_hit": int(hits), "hyps": hyps,
           "scores": np.asarray(scores, np.float32).view(np.uint32).tolist(),
           "compactions": st["compactions"], "kernels": st["kernels_per_step"]}
elif mode == "callback":  # a step callback that translates on another handle of the same device
    t2 = ct2.Translator(path, device="cuda", compute_type="int8")
    inner = []
    def cb(step):
        if not inner:
            inner.append(t2.translate_ids(src[:2], **kw)[0])
        return False
    base, _ = t.translate_ids(src[:2], **kw)
    hyps, _ = t.translate_ids(src[:2], callback=cb, **kw)
    out = {"base": base, "hyps": hyps, "inner": inner[0]}
print(json.dumps(out))
"""


def _worker(args, env=None, wait=True, timeout=600):
    p = subprocess.Popen([sys.executable, "-c", WORKER % {"root": ROOT, "here": HERE}] + args,
                         env=env or dict(os.environ), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if not wait:
        return p
    try:
        o, e = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        p.kill()
        raise
    assert p.returncode == 0, e[-3000:]
    return json.loads(o.strip().splitlines()[-1])


@pytest.mark.gpu
def test_starved_cluster_is_decoded_again_not_thrown(gpu):
    r = _worker(["inject"])
    assert r["k_fused"] <= 3, "the decoder-stack launch did not run"
    assert r["again"] == r["base"] and r["later"] == r["base"]
    assert r["fallbacks"] == 1
    assert r["k_after"] > 40, "after a starved cluster the handle stays on the launch-per-op step"
    ref = _worker(["inject"], env=dict(os.environ, CT2X_CLUSTER_STACK="0"))
    assert ref["base"] == r["base"] and ref["beam"] == r["beam"] and ref["fallbacks"] == 0


@pytest.mark.gpu
def test_two_processes_on_one_gpu_both_translate(gpu, tmp_path):
    """Two replica processes on one GPU: whoever holds the device's gate runs the decoder-stack
    launch, the other one the launch-per-op step; both return the same tokens as a lone process,
    nobody throws, nobody is voided."""
    alone = _worker(["inject"], env=dict(os.environ, CT2X_CLUSTER_STACK="0"))["base"]
    fa, fb = str(tmp_path / "a_up"), str(tmp_path / "b_up")
    pa = _worker(["loop", fa, fb], wait=False)
    pb = _worker(["loop", fb, fa], wait=False)
    outs = []
    for p in (pa, pb):
        o, e = p.communicate(timeout=900)
        assert p.returncode == 0, e[-3000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    for r in outs:
        assert r["hyps"] == alone and r["same"] and r["calls"] >= 3
        assert r["fallbacks"] == 0, "the gate, not the time-out, keeps two processes apart"
    # the gate is tried per call: over six seconds each process ran the fused launch at least once
    # and the two never held it together (no fallbacks above)
    assert outs[0]["fused_calls"] + outs[1]["fused_calls"] >= 1
    print("two processes: calls %s, on the decoder-stack launch %s" %
          ([r["calls"] for r in outs], [r["fused_calls"] for r in outs]))


@pytest.mark.gpu
def test_callback_may_translate_on_the_same_device(gpu):
    """(the round-5 gate was a mutex held across the user's callback: a callback that translated on
    the same device waited for its own caller)"""
    r = _worker(["callback"], timeout=180)
    assert r["hyps"] == r["base"] and r["inner"] == r["base"]


@pytest.mark.gpu
def test_compaction_across_the_32_row_boundary_of_the_launch(gpu):
    """A 40-row greedy session runs the several-rows form of the launch; once half of its rows have
    ended they are dropped (CT2X_COMPACT_MIN_ROWS=2 here) and the session continues on the one-row
    form, whose hand-over lines and counters are indexed differently (Engine::compact_rows clears
    them when the form changes). Same tokens and score bits as the launch-per-op step."""
    env = dict(os.environ, CT2X_COMPACT_MIN_ROWS="2")
    fused = _worker(["compact"], env=env)
    plain = _worker(["compact"], env=dict(env, CT2X_CLUSTER_STACK="0"))
    assert fused["end_token"] == plain["end_token"] and fused["rows_hit"] >= 12
    assert fused["compactions"] >= 1 and plain["compactions"] >= 1
    assert fused["hyps"] == plain["hyps"] and fused["scores"] == plain["scores"]
    lengths = sorted(len(h[0]) for h in fused["hyps"])
    assert lengths[0] < 12 and lengths[-1] == 40, lengths  # (rows really ended at different steps)
    assert fused["kernels"] <= 4 < plain["kernels"]
