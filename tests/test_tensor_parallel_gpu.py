"""-m gpu tensor-parallel parity: the model sharded over `world` ranks
(heads and FFN inner dimension split, row-parallel layers all-reduced) must reproduce the
unsharded engine. The ranks run as threads of this process on the one visible GPU, connected by
the in-process communicator (ct2x_translator_create_tp_local); the RCCL binding used on real
multi-GPU nodes is exercised by a one-rank communicator.

int8: bit-identical (the exchange carries exact int32 partial products and a max of absmax).
float32: partial fp32 products are summed in rank order - tokens equal, scores within 1e-4."""
import os
import threading

import numpy as np
import pytest

import golden_cases as GC

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]


def _model_dir(key):
    from ctranslate2_amd import model_builder as mb
    path = "/tmp/ct2x_test_%s" % key
    if not os.path.exists(os.path.join(path, "model.bin")):
        if key in GC.SYNTHETIC:
            mb.build_transformer(path, **GC.SYNTHETIC[key]["model"])
        else:
            mb.build_decoder_lm(path, **GC.SYNTHETIC_LM[key]["model"])
    return path


def _run_ranks(ranks, fn):
    """fn(translator) on every rank concurrently (each rank = one host thread)."""
    out = [None] * len(ranks)
    err = [None] * len(ranks)

    def work(i):
        try:
            out[i] = fn(ranks[i])
        except Exception as e:  # noqa: BLE001 - surfaced below
            err[i] = e

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(ranks))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=180)
    for e in err:
        if e is not None:
            raise e
    return out


def _sources(key):
    cfg = GC.SYNTHETIC[key]
    rng = np.random.default_rng(cfg["data_seed"])
    vocab = cfg["model"]["vocab_size"]
    return [rng.integers(4, vocab, n).tolist() for n in cfg["src_lengths"]]


@pytest.mark.parametrize("key,world", [("tiny", 2), ("small", 2), ("small", 4)])
@pytest.mark.parametrize("beam", [1, 4])
def test_translate_int8_bit_identical(gpu, key, world, beam):
    import ctranslate2_amd as ct2
    path = _model_dir(key)
    cfg = GC.SYNTHETIC[key]
    src = _sources(key)
    kw = dict(beam_size=beam, max_decoding_length=cfg["max_len"], min_decoding_length=cfg["min_len"],
              return_scores=True, use_graph=False)
    single = ct2.Translator(path, compute_type="int8")
    ref_h, ref_s = single.translate_ids(src, **kw)
    ranks = ct2.Translator.tensor_parallel_local(path, world, compute_type="int8")
    outs = _run_ranks(ranks, lambda t: t.translate_ids(src, **kw))
    for hyps, scores in outs:
        assert hyps == ref_h
        assert np.array_equal(scores, ref_s)
    for t in ranks:
        t.close()


def test_step_logits_int8_bit_identical(gpu):
    import ctranslate2_amd as ct2
    path = _model_dir("small")
    cfg = GC.SYNTHETIC["small"]
    src = [[1] + s + [2] for s in _sources("small")]  # step_logits takes ids WITH specials
    m = max(len(s) for s in src)
    src = [s for s in src if len(s) == m]
    rng = np.random.default_rng(11)
    tgt = rng.integers(4, cfg["model"]["vocab_size"], (len(src), 6)).astype(np.int32)
    single = ct2.Translator(path, compute_type="int8")
    ref = single.step_logits(src, tgt)
    ranks = ct2.Translator.tensor_parallel_local(path, 2, compute_type="int8")
    outs = _run_ranks(ranks, lambda t: t.step_logits(src, tgt))
    for logits in outs:
        assert np.array_equal(logits, ref)
    for t in ranks:
        t.close()


@pytest.mark.parametrize("compute_type", ["float32", "float16", "bfloat16"])
def test_translate_float_types(gpu, compute_type):
    import ctranslate2_amd as ct2
    path = _model_dir("small")
    cfg = GC.SYNTHETIC["small"]
    src = _sources("small")
    kw = dict(beam_size=1, max_decoding_length=cfg["max_len"], min_decoding_length=cfg["min_len"],
              return_scores=True, use_graph=False)
    single = ct2.Translator(path, compute_type=compute_type)
    ref_h, ref_s = single.translate_ids(src, **kw)
    ranks = ct2.Translator.tensor_parallel_local(path, 2, compute_type=compute_type)
    outs = _run_ranks(ranks, lambda t: t.translate_ids(src, **kw))
    for hyps, scores in outs:
        if compute_type == "float32":
            assert hyps == ref_h
            np.testing.assert_allclose(scores, ref_s, atol=1e-4)
        else:  # a different fp32 summation order ahead of a 16-bit rounding: allow near-tie flips
            same = [a == b for a, b in zip(hyps, ref_h)]
            assert sum(same) >= len(same) - 1
            np.testing.assert_allclose(scores[same], ref_s[same], atol=5e-3)
    for t in ranks:
        t.close()


def test_generate_decoder_only_int8_bit_identical(gpu):
    """Llama-style path under tensor parallelism: GQA (8 heads / 2 KV heads -> 4 / 1 per rank),
    RoPE, RMSNorm, gated FFN, sequence-mode prefill."""
    import ctranslate2_amd as ct2
    key = "lm_small"
    path = _model_dir(key)
    cfg = GC.SYNTHETIC_LM[key]
    rng = np.random.default_rng(cfg["data_seed"])
    prompts = [rng.integers(4, cfg["model"]["vocab_size"], cfg["prompt_len"]).tolist()
               for _ in range(cfg["batch"])]
    kw = dict(max_length=cfg["max_len"], min_length=cfg["max_len"], return_scores=True, use_graph=False)
    single = ct2.Translator(path, compute_type="int8")
    ref_o, ref_s = single.generate_ids(prompts, **kw)
    ranks = ct2.Translator.tensor_parallel_local(path, 2, compute_type="int8")
    outs = _run_ranks(ranks, lambda t: t.generate_ids(prompts, **kw))
    for o, s in outs:
        assert o == ref_o
        assert np.array_equal(s, ref_s)
    for t in ranks:
        t.close()


def test_indivisible_heads_are_rejected(gpu):
    import ctranslate2_amd as ct2
    path = _model_dir("tiny")  # 4 heads
    with pytest.raises(ct2.Ct2xError, match="not divisible"):
        ct2.Translator.tensor_parallel_local(path, 3, compute_type="int8")


def test_rccl_binding_one_rank(gpu):
    """The RCCL entry points used on multi-GPU nodes (ncclGetUniqueId, ncclCommInitRank with the
    id by value, ncclAllReduce int32-sum / float-max / float-sum on the engine's stream)."""
    from ctranslate2_amd import _lib
    L = _lib.lib()
    _lib.check(L.ct2x_tp_selftest_rccl(0))


@pytest.mark.parametrize("world", [2, 4])
def test_local_collectives(gpu, world):
    """ReduceAll (int32 sum, float max) and GatherAll of the in-process communicator, checked on
    every rank (src/ops/nccl_ops_gpu.cu:53-85)."""
    from ctranslate2_amd import _lib
    L = _lib.lib()
    _lib.check(L.ct2x_tp_selftest_local(0, world))


# (in-process ranks share ONE GPU and their exchange kernels wait for each other: they need separate
# hardware queues, which two streams of different priority guarantee - so the double has two ranks;
# more ranks belong on more GPUs)
@pytest.mark.parametrize("world,rows,depth,n", [(2, 32, 256, 512), (2, 16, 7168, 4096), (2, 128, 512, 512)])
def test_one_shot_exchanges(gpu, world, rows, depth, n):
    """The one-shot exchanges of a row-parallel int8 Dense (tp_oneshot.hip: granules {payload, tag}
    stored into peer mailboxes and polled inside the producer / consumer kernels, no collective
    call): every rank's Quantize output equals the unsharded row's, every rank's dequantised sum
    equals the unsharded product, bit for bit - and the loopback price of the exchange pair on one
    device is printed (the number DESIGN section 7 quotes)."""
    import ctypes as C
    from ctranslate2_amd import _lib
    L = _lib.lib()
    us = C.c_float()
    _lib.check(L.ct2x_tp_selftest_oneshot(0, world, C.c_int64(rows), C.c_int64(depth), C.c_int64(n), 50,
                                          C.byref(us)))
    print("one-shot exchange pair, world %d, [%d, %d] -> [%d, %d]: %.2f us per row-parallel Dense "
          "(two launches incl. Quantize and Dequantize, single-device loopback)"
          % (world, rows, depth, rows, n, us.value))
    assert 0 < us.value < 500


def test_one_shot_off_is_the_collective_path(gpu):
    """CT2X_TP_ONESHOT=0 keeps the all-reduce form (what a communicator without peer mapping runs):
    same bits."""
    import ctranslate2_amd as ct2
    path = _model_dir("small")
    src = _sources("small")
    single = ct2.Translator(path, compute_type="int8")
    ref, _ = single.translate_ids(src, beam_size=1, max_decoding_length=12, min_decoding_length=12)
    os.environ["CT2X_TP_ONESHOT"] = "0"
    try:
        ranks = ct2.Translator.tensor_parallel_local(path, 2, compute_type="int8")
    finally:
        del os.environ["CT2X_TP_ONESHOT"]
    outs = _run_ranks(ranks, lambda t: t.translate_ids(src, beam_size=1, max_decoding_length=12,
                                                       min_decoding_length=12)[0])
    assert outs[0] == ref and outs[1] == ref
    for t in ranks:
        t.close()


@pytest.mark.parametrize("key,beam,use_graph", [("tiny", 1, False), ("small", 1, True), ("small", 4, True)])
def test_two_processes_one_gpu_int8_bit_identical(gpu, key, beam, use_graph):
    """The CROSS-PROCESS exchange path on the one visible GPU: two rank processes, shared-memory
    bootstrap, hipIpc peer-mapped mailboxes, one-shot exchanges polled inside kernels of two
    different processes (ct2x_translator_create_tp_shm; RCCL would refuse two ranks on one device).
    Both ranks must return the unsharded engine's hypotheses and scores bit for bit, twice."""
    import json
    import subprocess
    import sys
    import tempfile
    import ctranslate2_amd as ct2
    path = _model_dir(key)
    cfg = GC.SYNTHETIC[key]
    src = _sources(key)
    kw = dict(beam_size=beam, max_decoding_length=cfg["max_len"], min_decoding_length=cfg["min_len"],
              return_scores=True, use_graph=use_graph)
    single = ct2.Translator(path, compute_type="int8")
    ref_h, ref_s = single.translate_ids(src, **kw)
    single.close()
    here = os.path.dirname(os.path.abspath(__file__))
    seg = "ct2x_tp_test_%d_%s_%d" % (os.getpid(), key, beam)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if key == "small" and beam == 1:
        # a STALE segment of the same name, as a crashed run leaves it: right size, `ready` word set.
        # A rank that opens it before rank 0 has replaced it must notice (its handshake token never
        # comes back) and open the name again instead of spinning on the dead segment's barrier.
        with open("/dev/shm/" + seg, "wb") as f:
            f.write(np.array([0x43543258], np.uint32).tobytes())
            f.truncate(4096 + 2 * (8 << 20))
    with tempfile.TemporaryDirectory() as tmp:
        outs = [os.path.join(tmp, "rank%d.json" % r) for r in range(2)]
        procs = [subprocess.Popen([sys.executable, os.path.join(here, "_tp_shm_rank.py"), path, str(r),
                                   "2", seg, "int8", outs[r], str(beam), str(int(use_graph))],
                                  env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
                 for r in range(2)]
        logs = []
        for p in procs:
            try:
                logs.append(p.communicate(timeout=240))
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
        for p, (so, se) in zip(procs, logs):
            assert p.returncode == 0, se[-3000:]
        for r in range(2):
            res = json.load(open(outs[r]))
            assert res["tp"] == {"rank": r, "world": 2, "mailboxes": True, "backend": "shared-memory"}
            for h, s in ((res["hyps"], res["scores"]), (res["hyps2"], res["scores2"])):
                assert h == ref_h
                assert np.array_equal(np.asarray(s, dtype=np.float32), ref_s)


def test_rccl_two_process_tensor_parallel():
    """The REAL multi-GPU path - one process per GPU, RCCL bootstrap, peer-mapped mailboxes over
    xGMI (hipIpc), one-shot exchanges - through bench.py --tp under torch.distributed.run. Needs two
    visible GPUs: collected everywhere, skipped on single-GPU boxes (the driver's GPU tier has one)."""
    import subprocess
    import sys
    import torch
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(root, "bench.py"),
           "--gpus", "2", "--tp", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-l8",
           "--no-profile", "--check-tp-parity"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    import json
    res = json.loads(line)
    assert res["config"]["tp_tokens_equal_unsharded"] is True


def test_fast_exchange_mode_is_close_not_exact(gpu):
    """CT2X_TP_FAST=1: the reference's own exchange (local Quantize scale per rank, ONE fp32
    all-reduce of the dequantised partial outputs, src/layers/common.cc:348-401) instead of the
    exact int32 exchange. One collective less per row-parallel layer; the logits then differ from
    the unsharded engine by int8 quantisation noise, not more."""
    import ctranslate2_amd as ct2
    path = _model_dir("small")
    cfg = GC.SYNTHETIC["small"]
    src = [[1] + s + [2] for s in _sources("small")]
    m = max(len(s) for s in src)
    src = [s for s in src if len(s) == m]
    rng = np.random.default_rng(12)
    tgt = rng.integers(4, cfg["model"]["vocab_size"], (len(src), 6)).astype(np.int32)
    single = ct2.Translator(path, compute_type="int8")
    ref = single.step_logits(src, tgt)
    os.environ["CT2X_TP_FAST"] = "1"
    try:
        ranks = ct2.Translator.tensor_parallel_local(path, 2, compute_type="int8")
    finally:
        del os.environ["CT2X_TP_FAST"]
    outs = _run_ranks(ranks, lambda t: t.step_logits(src, tgt))
    assert np.array_equal(outs[0], outs[1])          # every rank holds the same sum
    err = np.abs(outs[0] - ref)
    spread = ref.std()
    print("fast exchange: max |d| = %.3g (logit sigma %.3g), identical = %s"
          % (err.max(), spread, np.array_equal(outs[0], ref)))
    assert err.max() < 0.15 * spread                  # quantisation noise of one more scale choice
    assert np.mean(outs[0].argmax(-1) == ref.argmax(-1)) > 0.9
    for t in ranks:
        t.close()
