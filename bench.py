#!/usr/bin/env python
"""Benchmark of the hot path named by BASELINE.json: generated tokens/s of int8 Transformer-base
En->De greedy decoding (batch 32, beam 1) on MI355X, whole-job over N GPUs (one process per GPU,
independent replicas = the reference's data-parallel ReplicaPool, no collective on the data path).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

One "step" = one Translator.translate_batch call over one synthetic batch: 32 source sentences of
32 random token ids, 64 forced decoding steps (min = max decoding length) => exactly 2048 generated
tokens per step per GPU, encoder + decoder + sampling + host loop included, model load excluded
(the reference's own tokens/s accounting, include/ctranslate2/translator.h:176-209). Inputs are
host id arrays (a few KB); weights, KV cache and all activations are resident in HBM.

Rank 0 prints ONE JSON line. Besides the contract fields it carries
  roofline:     the dominant kernel of the step, timed with HIP events on the engine's stream in a
                profiling pass of the same workload (ct2x_profile_begin/end), algorithmic bytes from
                the engine's per-launch accounting (DESIGN.md "Roofline accounting").
  cpu_baseline: the REAL reference CPU library (oracle/_ref: CTranslate2 + MKL, int8) timed on the
                host cores on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)
INT8_MFMA_PEAK_TOPS = 5000.0   # dense int8 MFMA ~= 2x bf16 (same guide)

TB = dict(num_encoder_layers=6, num_decoder_layers=6, d_model=512, ffn_dim=2048, num_heads=8,
          vocab_size=32000, seed=0)


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--batch", type=int, default=32)
    p.add_argument("--beam", type=int, default=1)
    p.add_argument("--src-len", type=int, default=32)
    p.add_argument("--max-len", type=int, default=64)
    p.add_argument("--streams", type=int, default=0,
                   help="concurrent decode streams per GPU (0 = library default)")
    p.add_argument("--no-graph", action="store_true")
    p.add_argument("--no-session-reuse", action="store_true",
                   help="every translate call builds its own decoding session (CT2X_SESSION_CACHE=0): "
                        "the A/B of DESIGN 5.000 'Session reuse'")
    p.add_argument("--compute-type", default="int8",
                   help="int8 (the metric), float16, bfloat16 or float32: same model, weights "
                        "converted at load")
    p.add_argument("--tp", action="store_true",
                   help="tensor parallel over the --gpus ranks (one sharded model, every rank runs "
                        "the same batch, RCCL all-reduce after each row-parallel Dense) instead of "
                        "independent replicas; strong scaling")
    p.add_argument("--tp-model", default="tb", choices=["tb", "nllb"],
                   help="with --tp: the model the ranks share - tb (Transformer-base int8, the metric's model) or "
                        "nllb (NLLB-200-3.3B geometry, float16, batch 64 beam 4: BASELINE configs[3], which the "
                        "reference runs with TP = 4)")
    p.add_argument("--plan", action="store_true",
                   help="rendezvous (gloo when no GPU is visible), agree on the run and print it from rank 0 "
                        "WITHOUT touching a GPU: what the multi-GPU invocation would do (tests/test_distributed_cpu.py)")
    p.add_argument("--check-tp-parity", action="store_true",
                   help="with --tp: rank 0 also decodes the batch on an unsharded engine and reports "
                        "whether the tensor-parallel tokens are identical (config.tp_tokens_equal_unsharded)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-l8", action="store_true",
                   help="skip the Llama-3-8B-shape leg (BASELINE configs[2], reported as config.l8)")
    p.add_argument("--no-whisper", action="store_true",
                   help="skip the Whisper-large-v3-shaped bfloat16 leg (BASELINE configs[4] family, "
                        "reported as config.whisper)")
    p.add_argument("--no-nllb", action="store_true",
                   help="skip the NLLB-3.3B-shape float16 leg (BASELINE configs[3] geometry on one GPU, "
                        "reported as config.nllb)")
    p.add_argument("--no-profile", action="store_true")
    p.add_argument("--no-token-check", action="store_true",
                   help="skip config.tokens_equal_unfused (a second interpreter decodes the last timed "
                        "batch with every fusion of the step switched off)")
    p.add_argument("--tp-shm", action="store_true",
                   help="on ONE GPU: two PROCESSES share it and run the tensor-parallel engine through "
                        "the shared-memory communicator (ShmComm) - the multi-rank code path without a "
                        "second device; reported as config.tp_shm")
    p.add_argument("--unfused-tokens-of", type=int, default=-1, help=argparse.SUPPRESS)
    p.add_argument("--cpu-threads", type=int, default=0, help="host threads for the CPU baseline")
    return p.parse_args()


NLLB = dict(num_encoder_layers=24, num_decoder_layers=24, d_model=2048, ffn_dim=8192, num_heads=16,
            vocab_size=256206, seed=0, quantization="float16", direct_f16=True, share_embeddings=True)


def run_plan(args, rank, world):
    """What this invocation does, as data (printed by --plan, and the same decisions main() takes)."""
    tp = args.tp and world > 1
    model = args.tp_model if tp else "tb"
    return {"world": world, "rank": rank,
            "parallelism": ("tensor parallel x%d" % world) if tp else ("replicas x%d" % world),
            "model": {"tb": "Transformer-base int8", "nllb": "NLLB-200-3.3B geometry float16"}[model],
            "batch": 64 if model == "nllb" else args.batch, "beam": 4 if model == "nllb" else args.beam,
            "scaling": "strong" if tp else "weak",
            "before_the_timed_region": (["ct2x_tp_selftest_multi on every rank (RCCL unique id of rank 0 by "
                                         "torch.distributed broadcast)"] if world > 1 else []),
            "data_path_collective": ("all-reduce of int32 partial products / one-shot mailbox exchange after "
                                     "every row-parallel Dense") if tp else None}


def plan_only(args, rank, local_rank, world):
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group(backend="nccl" if torch.cuda.is_available() else "gloo")
    plan = run_plan(args, rank, world)
    # the 128 bytes every rank needs travel the way the real run sends them
    uid = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        uid = torch.arange(128, dtype=torch.uint8)
    if world > 1:
        dist.broadcast(uid, src=0)
        plans = [None] * world
        dist.all_gather_object(plans, plan)
    else:
        plans = [plan]
    ok = bool((uid == torch.arange(128, dtype=torch.uint8)).all()) and \
        all(p["parallelism"] == plans[0]["parallelism"] and p["model"] == plans[0]["model"] for p in plans) and \
        sorted(p["rank"] for p in plans) == list(range(world))
    if rank == 0:
        print(json.dumps({"plan": plans[0], "ranks_agree": ok, "id_bytes_received": int(uid.numel())}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("bench.py --plan: the ranks disagree")


def communicator_selftest(ct2, dist, torch, rank, local_rank, world):
    """ct2x_tp_selftest_multi on every rank; the unique id of rank 0 travels by broadcast."""
    uid = torch.zeros(128, dtype=torch.uint8, device=torch.device("cuda", local_rank))
    if rank == 0:
        uid.copy_(torch.frombuffer(bytearray(ct2.tp_unique_id()), dtype=torch.uint8))
    dist.broadcast(uid, src=0)
    try:
        out = ct2.tp_selftest_multi(local_rank, rank, world, unique_id=bytes(uid.cpu().numpy().tobytes()))
    except Exception as e:  # noqa: BLE001 - every rank must reach the gather below
        out = {"ok": False, "rank": rank, "error": "%s: %s" % (type(e).__name__, e)}
    outs = [None] * world
    dist.all_gather_object(outs, out)
    bad = [o for o in outs if not o.get("ok")]
    if bad:
        raise SystemExit("bench.py: the tensor-parallel communicator failed its self-test before anything was "
                         "timed: %s" % json.dumps(bad))
    return {"ok": True, "world": world, "backend": outs[0]["backend"], "mailboxes": all(o["mailboxes"] for o in outs),
            "oneshot_us_per_dense_max": max((o["oneshot_us_per_dense"] or 0.0) for o in outs)}


def make_sources(rank, batch, src_len, vocab):
    rng = np.random.default_rng(1 + rank)
    return [rng.integers(4, vocab, src_len).tolist() for _ in range(batch)]


def shard_info():
    """(rank, local_rank, world) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def aggregate_throughput(local_tokens, elapsed_s, dist=None, device=None):
    """Whole-job throughput: tokens of ALL ranks / max-over-ranks time (weak scaling: every rank
    processes its own batch). Works with any torch.distributed backend (gloo in the CPU tests)."""
    import torch
    if dist is None or not dist.is_initialized():
        return float(local_tokens) / elapsed_s, elapsed_s, int(local_tokens)
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    n = torch.tensor([float(local_tokens)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(n.item()) / float(t.item()), float(t.item()), int(n.item())


NOOP_LABEL = "noop[calibration]"


def pick_dominant(profile):
    """The (kernel, launch shape) with the largest accumulated device time. Profile labels carry the
    launch shape (gemm_s8_dequant[m=32,n=512,k=512], ...); the dominant LABEL is reported - one
    population, comparable with the same instantiation's line in rocprofv3's kernel stats - and the
    per-kernel totals stay visible in `profile.top`."""
    label = max(profile, key=lambda k: profile[k]["total_ms"])
    st = dict(profile[label])
    st["main_shape"] = label
    return label, st


def pmc_traffic(label):
    """HBM bytes per launch from the committed PMC passes (profiles/*_pmc_traffic.json: rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 correction applied) or None."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if name.endswith("_pmc_traffic.json"):
            with open(os.path.join(pdir, name)) as f:
                entry = json.load(f).get(label)
            if entry:
                best = entry["traffic_bytes"]
    return best


def roofline_entry(label, st):
    avg_ms = st["total_ms"] / max(st["launches"], 1)
    seconds = st["total_ms"] * 1e-3
    gbs = st["bytes"] / seconds / 1e9 if st["bytes"] > 0 and seconds > 0 else None
    tops = st["ops"] / seconds / 1e12 if st["ops"] > 0 and seconds > 0 else None
    entry = {"kernel": label, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": (gbs / HBM_PEAK_GBS) if gbs else None,
             "traffic": pmc_traffic(st.get("main_shape", label)),
             "launches": st["launches"], "avg_us": avg_ms * 1e3,
             "bytes_per_launch": st["bytes"] / max(st["launches"], 1),
             "main_shape": st.get("main_shape", label),
             "timing": "HIP events around every launch of an eager (graph-free) pass of the same "
                       "workload on the engine's stream; includes the event pair's own cost "
                       "(about 3-4 us per launch at these sizes: a decode step is latency bound, "
                       "see DESIGN.md section 4)"}
    if tops is not None:
        entry["int8_tops"] = tops
        entry["mfma_frac"] = tops / INT8_MFMA_PEAK_TOPS
    if label.startswith("decoder_stack"):
        # The decoder-stack launch (csrc/decoder_cluster.hip) is bound by neither HBM nor MFMA: every
        # cluster (one per batch row) streams its own copy of the weights through the L2 -> CU path
        # (~120 GB/s per CU, profiles/r05_row_resident_probe.txt) and the layers are chains of 6
        # hand-overs + 2 attention cores + 3 norm chains each. Both figures beside
        # the HBM fraction the contract asks for:
        rows = int(label.split("m=")[1].split(",")[0]) if "m=" in label else 32
        l2_bytes = entry["bytes_per_launch"] * rows
        entry["l2_to_cu"] = {"bytes_per_launch": l2_bytes,
                             "achieved_GBps": l2_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else None,
                             "measured_ceiling_GBps": 28770.0,
                             "note": "the same weight bytes once per cluster (batch row); ceiling = 256 CUs x "
                                     "112 GB/s measured with every workgroup walking one weight set"}
        entry["bound_note"] = ("latency: per layer 6 hand-overs between the 8 workgroups of a row, 2 attention cores, "
                               "3 norm statistics chains on the dependent path; the HBM figure above is the "
                               "contract's (weights once per launch / launch time)")
    return entry


def prefill_gemm_leg():
    """The MFMA-bound face of the same Dense (north_star's ">= 70 % of the int8-MFMA
    roofline for the projection GEMMs"): the fused Quantize'd-input GEMM + Dequantize (+ activation)
    launch at the two M >= 1024 shapes of the path - the Llama-3-8B prefill FFN-in projection
    [2048 x 4096] x [14336 x 4096]^T with the Swish epilogue, and Transformer-base's cross-attention
    K/V projection [1024 x 512] x [1024 x 512]^T - random int8 operands resident in HBM, fp32 result
    written to HBM, HIP events on the launch stream around `reps` back-to-back launches.
    algorithmic ops = 2 M N K; bytes = M K + N K + 4 N + 4 M + 4 M N."""
    import ctypes as C
    import torch
    from ctranslate2_amd import _lib
    L = _lib.lib()
    out = {}
    L.ct2x_thread_stream.restype = C.c_void_p
    stream = C.c_void_p(L.ct2x_thread_stream())
    notes = {"tb_cross_kv": "a 1 GOP launch: latency bound (launch + one K loop of 4 stages), NOT an MFMA datum",
             "tb_encoder_ffn_in_relu_8192": "K = 512: 17 GOP against 71 MB of fp32 output - bound by the result's "
                                            "HBM write (hbm_frac), not by the matrix pipe"}
    for name, (m, n, k, act, reps) in {"l8_prefill_ffn_in_swish": (2048, 14336, 4096, 2, 20),
                                       "tb_encoder_ffn_in_relu_8192": (8192, 2048, 512, 0, 50),
                                       "tb_cross_kv": (1024, 1024, 512, -1, 200)}.items():
        g = torch.Generator(device="cuda").manual_seed(m + n + k)
        qa = torch.randint(-127, 128, (m, k), dtype=torch.int8, device="cuda", generator=g)
        w = torch.randint(-127, 128, (n, k), dtype=torch.int8, device="cuda", generator=g)
        sa = torch.rand(m, device="cuda", generator=g) * 20 + 1
        sw = torch.rand(n, device="cuda", generator=g) * 200 + 50
        y = torch.empty((m, n), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()

        def launch():
            _lib.check(L.ct2x_gemm_s8_dequant(C.c_void_p(qa.data_ptr()), C.c_void_p(sa.data_ptr()),
                                              C.c_void_p(w.data_ptr()), C.c_void_p(sw.data_ptr()), None, None,
                                              C.c_int64(m), C.c_int64(n), C.c_int64(k), C.c_int(act),
                                              C.c_void_p(y.data_ptr()), C.c_int(0), stream))
        for _ in range(3):
            launch()
        e0, e1 = C.c_void_p(), C.c_void_p()
        _lib.check(L.ct2x_event_create(C.byref(e0)))
        _lib.check(L.ct2x_event_create(C.byref(e1)))
        _lib.check(L.ct2x_event_record(e0, stream))
        for _ in range(reps):
            launch()
        _lib.check(L.ct2x_event_record(e1, stream))
        ms = C.c_float()
        _lib.check(L.ct2x_event_elapsed_ms(e0, e1, C.byref(ms)))
        L.ct2x_event_destroy(e0)
        L.ct2x_event_destroy(e1)
        us = ms.value * 1e3 / reps
        ops = 2.0 * m * n * k
        nbytes = m * k + n * k + 4 * n + 4 * m + 4.0 * m * n
        tops = ops / (us * 1e-6) / 1e12
        out[name] = {"m": m, "n": n, "k": k, "activation": {2: "swish", 0: "relu", -1: "none"}[act],
                     "avg_us": us, "launches": reps,
                     "roofline": {"bound": "mfma", "achieved": tops, "peak": INT8_MFMA_PEAK_TOPS,
                                  "unit": "TOP/s", "frac": tops / INT8_MFMA_PEAK_TOPS,
                                  "ops_per_launch": ops, "bytes_per_launch": nbytes,
                                  "hbm_frac": nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": None}}
        if name in notes:
            out[name]["note"] = notes[name]
        del qa, w, y
    return out


def ragged_leg(translator, args):
    """What the padded source positions cost (the reference removes padding in the
    encoder and the cross-attention K/V projection for the int8 compute types, src/padder.cc:56-74;
    this backend computes all batch x longest rows). Batches of 32 and 256 sentences with lengths
    ~ U[8, 32] against the same batches at the full length 32: tokens/s, the encoder + memory
    projection time of a call (ct2x_last_stats.encode_ms), the share of padded rows, and the upper
    bound of what gathering the valid rows could return = encode_ms x padded share / call time."""
    out = {}
    for batch in (32, 256):
        rng = np.random.default_rng(100 + batch)
        lengths = rng.integers(8, 33, batch)
        ragged = [rng.integers(4, TB["vocab_size"], int(n)).tolist() for n in lengths]
        full = [rng.integers(4, TB["vocab_size"], 32).tolist() for _ in range(batch)]
        kw = dict(beam_size=1, max_decoding_length=args.max_len, min_decoding_length=args.max_len)
        entry = {}
        for name, src in (("full_length", full), ("ragged", ragged)):
            for _ in range(2):
                translator.translate_ids(src, **kw)
            t0 = time.perf_counter()
            reps = 5
            enc = 0.0
            for _ in range(reps):
                translator.translate_ids(src, **kw)
                enc += translator.last_stats()["encode_ms"]
            ms = (time.perf_counter() - t0) / reps * 1e3
            entry[name] = {"ms_per_call": ms, "encode_ms": enc / reps,
                           "tokens_per_s": batch * args.max_len / (ms * 1e-3)}
        padded = 1.0 - float(lengths.sum()) / (batch * int(lengths.max()))
        entry["padded_row_share"] = padded
        entry["padding_cost_upper_bound_pct"] = 100.0 * entry["ragged"]["encode_ms"] * padded / entry["ragged"]["ms_per_call"]
        out["batch%d" % batch] = entry
    out["note"] = ("lengths ~ U[8, 32]; padding removal could return at most padding_cost_upper_bound_pct of a call "
                   "(the decode steps are row-wise per sentence and never see the padding)")
    return out


def tp_local_leg(model_dir, sources, kw, ref_hyps):
    """Tensor parallelism on the ONE visible GPU: the model sharded over 2 ranks that live as threads
    of this process (in-process communicator), row-parallel layers exchanged through the one-shot
    mailbox kernels (tp_oneshot.hip) - the code path a multi-GPU node runs over xGMI, minus the
    links. Reports tokens/s of the sharded decode (NOT a scaling number: both ranks share one GPU),
    whether its tokens equal the unsharded engine's, and the loopback price of one exchange pair."""
    import ctypes as C
    import threading
    import ctranslate2_amd as ct2
    from ctranslate2_amd import _lib
    ranks = ct2.Translator.tensor_parallel_local(model_dir, 2, compute_type="int8")
    outs, errs = [None, None], [None, None]

    def work(i):
        try:
            ranks[i].translate_ids(sources, **kw)  # warm-up (graph capture)
            t0 = time.perf_counter()
            for _ in range(3):
                outs[i] = ranks[i].translate_ids(sources, **kw)[0]
            outs[i] = (outs[i], (time.perf_counter() - t0) / 3)
        except Exception as e:  # noqa: BLE001
            errs[i] = e
    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    for e in errs:
        if e is not None:
            raise e
    for t in ranks:
        t.close()
    (hyps0, dt0), (hyps1, dt1) = outs
    tokens = sum(len(h[0]) for h in hyps0)
    us = C.c_float()
    L = _lib.lib()
    _lib.check(L.ct2x_tp_selftest_oneshot(0, 2, C.c_int64(32), C.c_int64(256), C.c_int64(512), 200, C.byref(us)))
    return {"workload": "same model and batch, tensor parallel x2 as two in-process ranks on this one GPU "
                        "(one-shot mailbox exchanges; functional leg, not a scaling number)",
            "value": tokens / max(dt0, dt1), "unit": "tokens/s",
            "tokens_equal_unsharded": bool(hyps0 == ref_hyps and hyps1 == ref_hyps),
            "exchange_pair_us_loopback": us.value,
            "collective_calls_per_row_parallel_dense": 0, "device_side_exchanges_per_row_parallel_dense": 2}


def tp_shm_leg(model_dir, args, ref_hyps):
    """Tensor parallelism across two PROCESSES on the one visible GPU (tools/tp_shm_rank.py): the
    sharded engine over the shared-memory / hipIpc communicator - separate address spaces, mailboxes
    mapped through IPC handles, the bootstrap a multi-GPU node runs - minus the links. Functional
    leg: tokens/s of the sharded decode with both ranks on ONE GPU is not a scaling number."""
    import subprocess
    import tempfile
    seg = "/ct2x_bench_%d" % os.getpid()
    tmp = tempfile.mkdtemp(prefix="ct2x_tp_shm_")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(2):
        out = os.path.join(tmp, "rank%d.json" % r)
        procs.append((out, subprocess.Popen(
            [sys.executable, os.path.join(ROOT, "tools", "tp_shm_rank.py"), model_dir, str(r), "2", seg, out,
             str(args.batch), str(args.src_len), str(args.max_len)], env=env, stdout=subprocess.PIPE,
            stderr=subprocess.PIPE, text=True)))
    outs = []
    for out, p in procs:
        try:
            _, err = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            raise RuntimeError("rank timed out")
        if p.returncode != 0:
            raise RuntimeError("rank failed: " + err[-400:])
        with open(out) as f:
            outs.append(json.load(f))
    tokens = sum(len(h[0]) for h in outs[0]["hyps"])
    return {"workload": "same model and batch, tensor parallel x2 as two PROCESSES on this one GPU (ShmComm: "
                        "shared-memory bootstrap, IPC-mapped one-shot mailboxes; functional leg, not a scaling number)",
            "value": tokens / max(o["seconds_per_call"] for o in outs), "unit": "tokens/s",
            "tokens_equal_unsharded": bool(all(o["hyps"] == ref_hyps for o in outs)),
            "backend": outs[0]["tp"].get("backend"), "world": outs[0]["tp"].get("world")}


def physical_cores():
    """Distinct (socket, core) pairs of /proc/cpuinfo; falls back to os.cpu_count()."""
    try:
        pairs = set()
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    pairs.add((phys, core))
                phys = core = None
        if pairs:
            return len(pairs)
    except OSError:
        pass
    return os.cpu_count() or 1


def cpu_baseline(model_dir, sources, args):
    """The reference's own CPU path (MKL int8, oracle/_ref) on a bounded sample of the same
    workload, on ALL physical cores of this host, in the two settings SURVEY section 8(d) names:
    one replica with intra_threads = N, and the reference's throughput recipe inter_threads = N/4,
    intra_threads = 4 (docs/performance.md:8-10). `value` is the better of the two."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _refshim as R
    if not R.available():
        return {"value": None, "unit": "tokens/s", "kind": "reference",
                "error": "oracle/_ref or the MKL runtime is not present on this host"}
    logical = os.cpu_count() or 1
    cores = args.cpu_threads or physical_cores()
    settings = [(1, cores)]
    if cores >= 8:
        settings.append((max(1, cores // 4), 4))
    runs = []
    for inter, intra in settings:
        m = R.RefModel(model_dir, "int8", inter, intra)
        # every replica gets batches of 8 sentences; the sample is the bench batch repeated
        need = max(len(sources), inter * 8)
        sample = (sources * ((need + len(sources) - 1) // len(sources)))[:need]
        mbs = len(sources) if inter == 1 else 8
        kw = dict(beam_size=args.beam, max_len=args.max_len, min_len=args.max_len,
                  return_scores=False, max_batch_size=mbs)
        m.translate_ids(sample, **kw)  # warm-up (MKL packing, thread pools)
        t0 = time.perf_counter()
        reps = tokens = 0
        while True:
            hyps, _ = m.translate_ids(sample, **kw)
            tokens += sum(len(h[0]) for h in hyps)
            reps += 1
            if time.perf_counter() - t0 > 8.0 or reps >= 5:
                break
        dt = time.perf_counter() - t0
        m.close()
        runs.append({"inter_threads": inter, "intra_threads": intra, "value": tokens / dt,
                     "sample": "%d x translate_batch(%d sentences, %d forced steps, "
                               "max_batch_size=%d), %.1f s" % (reps, len(sample), args.max_len, mbs, dt)})
    best = max(runs, key=lambda r: r["value"])
    return {"value": best["value"], "unit": "tokens/s", "cores": cores, "kind": "reference",
            "sample": "inter_threads=%d intra_threads=%d: %s; compute_type=int8 (MKL u8s8)" %
                      (best["inter_threads"], best["intra_threads"], best["sample"]),
            "settings": runs, "host_logical_cpus": logical, "host_physical_cores": physical_cores()}


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` without a torch.distributed.run environment: start the N ranks
    ourselves (one process per GPU, rendezvous on 127.0.0.1) and hand over to them."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def unfused_tokens(args, which):
    """The tokens of rotation batch `which` from a SECOND interpreter in which every fusion of the
    decode step is off (separate sampler launch, launch-per-op layers, a session per call, eager
    launches): what the timed path must reproduce token for token."""
    import subprocess
    env = dict(os.environ)
    env.update(CT2X_FUSED_ARGMAX="0", CT2X_CLUSTER_STACK="0", CT2X_SESSION_CACHE="0", CT2X_GROUP_GRAPH="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.abspath(__file__), "--unfused-tokens-of", str(which), "--batch", str(args.batch),
           "--beam", str(args.beam), "--src-len", str(args.src_len), "--max-len", str(args.max_len),
           "--compute-type", args.compute_type]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        return None
    return json.loads(r.stdout.strip().splitlines()[-1])


def rotation_batches(args, rank, tp):
    sources = make_sources(0 if tp else rank, args.batch, args.src_len, TB["vocab_size"])
    return [sources] + [make_sources(1000 * (i + 1) + (0 if tp else rank), args.batch, args.src_len,
                                     TB["vocab_size"]) for i in range(3)]


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(args)
    if args.no_session_reuse:
        os.environ["CT2X_SESSION_CACHE"] = "0"  # (read by the library at its first translate call)
    rank, local_rank, world = shard_info()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    import torch
    import torch.distributed as dist
    if args.plan:
        return plan_only(args, rank, local_rank, world)
    if not torch.cuda.is_available():
        sys.stderr.write("bench.py needs an MI355X: no GPU is visible and there is no CPU fallback\n")
        sys.stderr.flush()
        if world > 1:
            # (a launcher tears the other ranks down the moment one exits: wait until every rank has
            # said the same thing - one marker file per rank, keyed by the rendezvous port - so that
            # the log shows every rank got this far even when their imports took very different times)
            import tempfile
            mark = os.path.join(tempfile.gettempdir(), "ct2x_bench_nogpu_%s_" % os.environ.get("MASTER_PORT", "0"))
            open(mark + str(rank), "w").close()
            deadline = time.time() + 180.0
            while time.time() < deadline and not all(os.path.exists(mark + str(r)) for r in range(world)):
                time.sleep(0.1)
            time.sleep(0.5)
            try:
                os.remove(mark + str(rank))
            except OSError:
                pass
        raise SystemExit(1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)

    import ctranslate2_amd as ct2
    from ctranslate2_amd import model_builder as mb

    # More than one GPU: the communicator first (every collective the tensor-parallel engine calls and
    # the peer-mailbox exchanges, bit-checked on every rank) - the first multi-GPU run names the piece
    # that fails instead of printing a number.
    selftest = None
    if world > 1:
        selftest = communicator_selftest(ct2, dist, torch, rank, local_rank, world)
    tp = args.tp and world > 1
    if tp and args.tp_model == "nllb":
        args.compute_type, args.batch, args.beam = "float16", 64, 4
        model_dir = "/tmp/ct2x_bench_nllb_L24_d2048"
        if rank == 0 and not os.path.exists(os.path.join(model_dir, "model.bin")):
            mb.build_transformer(model_dir, **NLLB)
        dist.barrier()
    else:
        model_dir = "/tmp/ct2x_bench_tb_r%d" % local_rank
        if not os.path.exists(os.path.join(model_dir, "model.bin")):
            mb.build_transformer(model_dir, **TB)
    translator = ct2.Translator(model_dir, device="cuda", device_index=local_rank, compute_type=args.compute_type,
                                tensor_parallel=tp)
    if args.unfused_tokens_of >= 0:
        # (child of unfused_tokens(): one eager call, tokens out, nothing else)
        src = rotation_batches(args, 0, False)[args.unfused_tokens_of]
        hyps, _ = translator.translate_ids(src, beam_size=args.beam, max_decoding_length=args.max_len,
                                           min_decoding_length=args.max_len, use_graph=False)
        print(json.dumps([h[0] for h in hyps]))
        return
    # replicas: every rank has its own batch; tensor parallel: ONE batch, computed by all ranks
    sources = make_sources(0 if tp else rank, args.batch, args.src_len, TB["vocab_size"])
    kw = dict(beam_size=args.beam, max_decoding_length=args.max_len, min_decoding_length=args.max_len,
              use_graph=not args.no_graph, num_streams=args.streams)

    # Every call of the timed region translates OTHER sentences of the same shape (four batches in
    # rotation; `sources` itself serves the later legs): the session reuse of the library
    #  skips allocation and graph capture, never a kernel of the encoder or decoder.
    rotation = rotation_batches(args, rank, tp)
    assert rotation[0] == sources
    calls = [0]
    last = {}

    def one_step():
        which = calls[0] % len(rotation)
        calls[0] += 1
        hyps, _ = translator.translate_ids(rotation[which], **kw)
        last["which"], last["tokens"] = which, hyps
        return sum(len(h[0]) for h in hyps)

    for _ in range(args.warmup):
        one_step()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    tokens = 0
    for _ in range(args.steps):
        tokens += one_step()   # translate_ids returns after the engine's stream has drained
    fence()
    elapsed = time.perf_counter() - t0
    value, max_elapsed, total_tokens = aggregate_throughput(tokens, elapsed, dist if world > 1 else None, device)
    if tp:  # the ranks produced the SAME tokens together
        total_tokens //= world
        value /= world
    stats = translator.last_stats()
    timed_last = dict(last)

    result = {
        "metric": "generated tokens/sec (%s, batch=%d, beam=%d), whole job" % (args.compute_type, args.batch, args.beam),
        "value": value,
        "unit": "tokens/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": max_elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if tp else "weak",
        "vs_baseline": None,
        "dtype": args.compute_type,
        "data": "synthetic",
        "config": {
            "workload": ("NLLB-200-3.3B geometry (24 + 24 layers, d_model 2048, FFN 8192, vocabulary 256206) "
                         "float16, batch=%d beam=%d, source length %d, %d forced decode steps, random-init "
                         "weights (BASELINE configs[3])" % (args.batch, args.beam, args.src_len, args.max_len))
                        if (tp and args.tp_model == "nllb") else
                        "Transformer-base En->De %s, batch=%d beam=%d, "
                        "source length %d, %d forced decode steps, random-init weights N(0,0.02)"
                        % ("int8 (int8 weights + int8 activations, fp32 accumulate/epilogue = compute "
                           "type int8_float32)" if args.compute_type == "int8" else args.compute_type,
                           args.batch, args.beam, args.src_len, args.max_len),
            "tokens_per_step_per_gpu": tokens // max(args.steps, 1),
            "parallelism": ("tensor parallel x%d (heads / FFN split, RCCL all-reduce of int32 partial "
                            "products after every row-parallel Dense)" % world) if tp else
                           "replicas x%d (one process per GPU, no data-path collective)" % world,
            "hip_graph": not args.no_graph,
            "streams_per_gpu": args.streams if args.streams > 0 else int(os.environ.get("CT2X_NUM_STREAMS", "1")),
            "kernels_per_decode_step": stats["kernels_per_step"],
            "communicator_selftest": selftest,
            "source_batches_in_rotation": len(rotation),
            "session_reuse_hits": stats.get("session_hits", 0),
            "decode_ms_last_call": stats["decode_ms"],
            "encode_ms_last_call": stats["encode_ms"],
        },
    }

    if rank == 0 and not tp and not args.no_token_check and timed_last:
        # the LAST call of the timed region (fused arg-max, decoder-stack launch, reused session,
        # grouped step graphs) against a second interpreter with all of that switched off
        ref = unfused_tokens(args, timed_last["which"])
        result["config"]["tokens_equal_unfused"] = (None if ref is None else
                                                    bool([h[0] for h in timed_last["tokens"]] == ref))

    if rank == 0 and not tp and not args.no_session_reuse and not args.no_profile and args.gpus == 1:
        # the same timed loop WITHOUT session reuse (every call builds, captures and releases its own
        # decoding session), in its own interpreter: the switch is read once per process
        try:
            import subprocess
            env = dict(os.environ, CT2X_SESSION_CACHE="0")
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--no-session-reuse", "--steps", "10",
                                  "--warmup", "2", "--batch", str(args.batch), "--beam", str(args.beam),
                                  "--src-len", str(args.src_len), "--max-len", str(args.max_len),
                                  "--compute-type", args.compute_type, "--no-cpu-baseline", "--no-profile",
                                  "--no-l8", "--no-whisper", "--no-nllb", "--no-token-check"],
                                 env=env, capture_output=True, text=True, timeout=300)
            nsr = json.loads(out.stdout.strip().splitlines()[-1])
            result["config"]["no_session_reuse"] = {"value": nsr["value"], "unit": nsr["unit"],
                                                    "ms_per_step": nsr["ms_per_step"],
                                                    "session_reuse_hits": nsr["config"]["session_reuse_hits"]}
        except Exception as e:  # never break the headline line
            result["config"]["no_session_reuse"] = {"error": "%s: %s" % (type(e).__name__, e)}

    if tp and args.check_tp_parity:
        hyps_tp, _ = translator.translate_ids(sources, **kw)
        if rank == 0:
            single = ct2.Translator(model_dir, device="cuda", device_index=local_rank, compute_type=args.compute_type)
            hyps_1, _ = single.translate_ids(sources, **kw)
            result["config"]["tp_tokens_equal_unsharded"] = bool(hyps_tp == hyps_1)
            single.close()
        dist.barrier()

    if rank == 0 and world == 1 and args.beam == 1 and not args.no_profile:
        # BASELINE.json configs[1] (batch 32 x beam 4) beside the metric's beam-1 line: a short,
        # separately timed leg after the measured region (5 calls), reported as `config.beam4`.
        kw4 = dict(kw, beam_size=4)
        translator.translate_ids(sources, **kw4)
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        tok4 = 0
        for _ in range(5):
            hyps4, _ = translator.translate_ids(sources, **kw4)
            tok4 += sum(len(h[0]) for h in hyps4)
        torch.cuda.synchronize()
        dt4 = time.perf_counter() - t4
        result["config"]["beam4"] = {"workload": "same model and batch, beam_size=4 (BASELINE configs[1])",
                                     "value": tok4 / dt4, "unit": "tokens/s", "ms_per_step": dt4 / 5 * 1e3}

    if rank == 0:
        if not args.no_profile:
            translator.profile_begin()
            one_step()
            profile = translator.profile_end()
            noop = profile.pop(NOOP_LABEL, None)  # the measurement's own calibration launches
            label, st = pick_dominant(profile)
            result["roofline"] = roofline_entry(label, st)
            # The step is latency bound: an event bracket around a single 3 us launch mostly measures
            # itself. The timed region gives one solid number - the wall time of a call replayed from
            # hipGraphs - so the dominant launch is priced by its SHARE of the eager pass applied to
            # that wall time (achieved / frac below); the bracket figures stay beside it.
            all_ms = sum(v["total_ms"] for v in profile.values())
            if all_ms > 0 and st["launches"] > 0 and st["bytes"] > 0:
                in_graph_us = elapsed / args.steps * 1e6 * (st["total_ms"] / all_ms) / st["launches"]
                r = result["roofline"]
                r["avg_us_event_bracket"] = r["avg_us"]
                r["achieved_event_bracket"] = r["achieved"]
                r["avg_us"] = in_graph_us
                r["achieved"] = st["bytes"] / st["launches"] / (in_graph_us * 1e-6) / 1e9
                r["frac"] = r["achieved"] / HBM_PEAK_GBS
                r["timing"] = ("wall time of the timed region (hipGraph replay, HIP-event / barrier "
                               "bracketed) x this launch's share of an eager profiling pass of the same "
                               "call (HIP events around every launch on the engine's stream) / its "
                               "launches per call; avg_us_event_bracket is the raw per-launch bracket, "
                               "noop_bracket_us the same bracket around an empty kernel")
            if noop and noop["launches"]:
                # the same event bracket around an empty kernel: what of avg_us is measurement and
                # dependent-launch cost rather than the kernel (rocprofv3's kernel trace has its own
                # constant: an empty kernel reads about 4 us there, see profiles/README.md)
                noop_us = noop["total_ms"] / noop["launches"] * 1e3
                result["roofline"]["noop_bracket_us"] = noop_us
                result["roofline"]["avg_us_above_noop"] = (
                    result["roofline"].get("avg_us_event_bracket", result["roofline"]["avg_us"]) - noop_us)
            total_ms = sum(v["total_ms"] for v in profile.values())
            top = sorted(profile.items(), key=lambda kv: -kv[1]["total_ms"])[:8]
            result["profile"] = {
                "device_ms_per_step_eager": total_ms,
                "top": [{"kernel": k, "launches": v["launches"], "total_ms": round(v["total_ms"], 4),
                         "share": round(v["total_ms"] / total_ms, 4),
                         "GBps": round(v["bytes"] / (v["total_ms"] * 1e-3) / 1e9, 1) if v["bytes"] and v["total_ms"] else None}
                        for k, v in top],
            }
        if world == 1 and not args.no_profile and args.beam == 1:
            try:
                result["config"]["ragged_sources"] = ragged_leg(translator, args)
            except Exception as e:  # noqa: BLE001
                result["config"]["ragged_sources"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_profile and args.compute_type == "int8" and args.beam == 1:
            try:
                ref_hyps, _ = translator.translate_ids(sources, **kw)
                result["config"]["tp_local"] = tp_local_leg(model_dir, sources, kw, ref_hyps)
            except Exception as e:  # never break the headline line
                result["config"]["tp_local"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and (args.tp_shm or not args.no_profile) and args.compute_type == "int8" and args.beam == 1:
            try:
                ref_hyps, _ = translator.translate_ids(sources, **kw)
                result["config"]["tp_shm"] = tp_shm_leg(model_dir, args, ref_hyps)
            except Exception as e:  # never break the headline line
                result["config"]["tp_shm"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_profile and args.compute_type == "int8":
            try:
                result["config"]["prefill_gemm"] = prefill_gemm_leg()
            except Exception as e:  # never break the headline line
                result["config"]["prefill_gemm"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_l8 and args.beam == 1 and args.compute_type == "int8":
            # BASELINE.json configs[2] (Llama-3 8B shape, int8, batch 16 greedy) under the same
            # clock: tools/bench_l8.py in its own process (7.5 GB of weights), two timed calls of
            # 16 prompts x 128 tokens + 128 forced new tokens; its decode step is the
            # HBM-roofline case of this backend (weights streamed once per step).
            try:
                import subprocess
                translator.close()
                out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_l8.py"),
                                      "--steps", "2", "--warmup", "1", "--no-profile"],
                                     capture_output=True, text=True, timeout=600)
                l8 = json.loads(out.stdout.strip().splitlines()[-1])
                result["config"]["l8"] = {
                    "workload": l8["config"]["workload"], "value": l8["value"], "unit": l8["unit"],
                    "ms_per_call": l8["ms_per_step"],
                    "prefill_ms": l8["config"]["prefill_ms_last_call"],
                    "decode_ms_per_step": l8["config"]["decode_ms_per_step"],
                    "kernels_per_decode_step": l8["config"]["kernels_per_decode_step"],
                    "roofline": l8["decode_roofline"],
                    "parity": "exact-order kernels on this path (attention_exact128.hip, RMSNorm, "
                              "int8 Dense): teacher-forced int8 logits max |d| = 0 and greedy tokens "
                              "identical to the reference's CPU path at this geometry (2 layers, "
                              "goldens from oracle/_ref): tests/test_generator_gpu.py::"
                              "test_llama3_geometry_teacher_forced_logits_int8, "
                              "::test_llama3_geometry_generate_matches_reference; the random weights "
                              "of this leg are not checked against anything",
                }
            except Exception as e:  # never break the headline line
                result["config"]["l8"] = {"value": None, "error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_nllb and args.beam == 1 and args.compute_type == "int8":
            # BASELINE.json configs[3]'s model and batch (NLLB-200 3.3B shape, float16, batch 64 beam 4)
            # on this one GPU: tools/bench_nllb.py in its own process (6.6 GB of weights).
            try:
                import subprocess
                translator.close()
                out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_nllb.py"),
                                      "--steps", "2", "--warmup", "1"],
                                     capture_output=True, text=True, timeout=900)
                nl = json.loads(out.stdout.strip().splitlines()[-1])
                result["config"]["nllb"] = {
                    "workload": nl["config"]["workload"], "value": nl["value"], "unit": nl["unit"],
                    "dtype": nl["dtype"], "ms_per_call": nl["ms_per_step"],
                    "encode_ms": nl["config"]["encode_ms_last_call"],
                    "decode_ms_per_step": nl["config"]["decode_ms_per_step"],
                    "roofline": nl["decode_roofline"],
                    "parity": "teacher-forced float16 logits within 1e-3 of the reference's float32 "
                              "evaluation at this geometry (2 + 2 layers): "
                              "tests/test_translator_gpu.py::test_nllb_geometry_float16_logits",
                }
            except Exception as e:  # never break the headline line
                result["config"]["nllb"] = {"value": None, "error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_whisper and args.beam == 1 and args.compute_type == "int8":
            # BASELINE.json configs[4]'s model family (Whisper-large-v3 shape, bfloat16, 8 windows of
            # 30 s, beam 5) on this one GPU: tools/bench_whisper.py in its own process.
            try:
                import subprocess
                out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_whisper.py"),
                                      "--steps", "2", "--warmup", "1"],
                                     capture_output=True, text=True, timeout=900)
                wl = json.loads(out.stdout.strip().splitlines()[-1])
                result["config"]["whisper"] = {
                    "workload": wl["config"]["workload"], "value": wl["value"], "unit": wl["unit"],
                    "dtype": wl["dtype"], "ms_per_call": wl["ms_per_step"],
                    "encode_ms": wl["config"]["encode_ms"],
                    "decode_ms_per_step": wl["config"]["decode_ms_per_step"],
                    "audio_seconds_per_second": wl["config"]["audio_seconds_per_second"],
                    "roofline": wl["encoder_roofline"],
                    "parity": "float32 tokens / language ranking / DTW paths identical to the reference's "
                              "models::Whisper on CPU, 16-bit arms within tolerance (synthetic 2 + 2 layer "
                              "model): tests/test_whisper_gpu.py",
                }
            except Exception as e:  # never break the headline line
                result["config"]["whisper"] = {"value": None, "error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline(model_dir, sources, args)
            except Exception as e:  # the baseline leg must never break the GPU measurement
                result["cpu_baseline"] = {"value": None, "unit": "tokens/s", "kind": "reference",
                                          "error": "%s: %s" % (type(e).__name__, e)}
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
