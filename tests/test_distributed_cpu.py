"""world_size-2 gloo test of the multi-GPU leg of bench.py (SURVEY 8(e) "replicas"): every rank
decodes its own shard of sentences with no data-path collective; the only exchanges are the
barrier and the max-over-ranks time / sum-of-tokens reductions. Runs on CPU (the GPU work is
replaced by a sleep of rank-dependent length)."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    import bench
    rank, local_rank, world = bench.shard_info()
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    sources = bench.make_sources(rank, 4, 5, 100)
    dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.05 * (rank + 1))          # rank 1 is the straggler
    elapsed = time.perf_counter() - t0
    tokens = 100 * (rank + 1)              # every rank processed its own units
    value, max_elapsed, total = bench.aggregate_throughput(tokens, elapsed, dist, torch.device("cpu"))
    print(json.dumps({"rank": rank, "value": value, "max_elapsed": max_elapsed, "total": total,
                      "elapsed": elapsed, "first_source": sources[0]}))
    dist.barrier()
    dist.destroy_process_group()
""" % ROOT)


def test_two_rank_weak_scaling_accounting(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = 29500 + os.getpid() % 2000
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=120)
        assert p.returncode == 0, err
        outs.append(json.loads(out.strip().splitlines()[-1]))
    outs.sort(key=lambda o: o["rank"])
    # whole-job value = all ranks' tokens / the slowest rank's time, identical on every rank
    assert outs[0]["total"] == outs[1]["total"] == 300
    assert abs(outs[0]["max_elapsed"] - outs[1]["max_elapsed"]) < 1e-9
    assert outs[0]["max_elapsed"] >= max(o["elapsed"] for o in outs) - 1e-9
    assert abs(outs[0]["value"] - 300 / outs[0]["max_elapsed"]) < 1e-6
    # shards are different sentences (independent units, no exchange)
    assert outs[0]["first_source"] != outs[1]["first_source"]


def test_single_process_path():
    sys.path.insert(0, ROOT)
    import bench
    value, elapsed, total = bench.aggregate_throughput(2048, 0.5)
    assert value == 4096 and elapsed == 0.5 and total == 2048
    label, st = bench.pick_dominant({"a": {"total_ms": 1.0, "launches": 2, "bytes": 1e6, "ops": 0},
                                     "b": {"total_ms": 3.0, "launches": 3, "bytes": 6e6, "ops": 3e9}})
    assert label == "b"
    r = bench.roofline_entry(label, st)
    assert abs(r["achieved"] - 2.0) < 1e-9 and r["bound"] == "hbm" and r["peak"] == 8000.0
    assert abs(r["frac"] - 2.0 / 8000.0) < 1e-12 and abs(r["avg_us"] - 1000.0) < 1e-9


TP_WORKER = textwrap.dedent("""
    import ctypes as C, json, os, sys
    sys.path.insert(0, %r)
    import numpy as np
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    rank, world = dist.get_rank(), dist.get_world_size()
    import ctranslate2_amd as ct2
    from ctranslate2_amd import _lib, model_builder as mb
    L = _lib.lib()
    # 1. the bootstrap of Translator(..., tensor_parallel=True): rank 0 draws the ncclUniqueId,
    #    every rank receives it over the process group
    uid = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        buf = C.create_string_buffer(128)
        _lib.check(L.ct2x_tp_unique_id(buf))
        uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
    dist.broadcast(uid, src=0)
    # 2. the partition this rank would load (fused Q | K | V of 8 / 2 heads of 16, host only)
    q, kv = 8 * 16, 2 * 16
    blocks = (C.c_int64 * 3)(q, kv, kv)
    first, count = (C.c_int64 * 3)(), (C.c_int64 * 3)()
    _lib.check(L.ct2x_tp_shard_rows(C.c_int64(q + 2 * kv), blocks, C.c_int32(3), C.c_int32(rank),
                                    C.c_int32(world), first, count))
    mine = torch.zeros(q + 2 * kv, dtype=torch.int32)
    for f, c in zip(first, count):
        mine[f:f + c] += 1
    dist.all_reduce(mine)
    # 3. the real constructor: bootstraps the same way, then must stop LOUDLY at the first HIP
    #    call on a box without a GPU (no CPU fallback, no hang)
    path = os.environ["CT2X_TP_MODEL"]
    if rank == 0 and not os.path.exists(os.path.join(path, "model.bin")):
        mb.build_transformer(path, num_encoder_layers=1, num_decoder_layers=1, d_model=32, num_heads=2, ffn_dim=64, vocab_size=40, seed=3)
    dist.barrier()
    err = None
    try:
        ct2.Translator(path, device="cuda", compute_type="int8", tensor_parallel=True)
    except Exception as e:
        err = "%%s: %%s" %% (type(e).__name__, e)
    print(json.dumps({"rank": rank, "uid": bytes(uid.numpy().tobytes()).hex(),
                      "cover": [int(mine.min()), int(mine.max())], "error": err}))
    dist.barrier()
    dist.destroy_process_group()
""" % ROOT)


def test_two_rank_tensor_parallel_bootstrap(tmp_path):
    """Two processes, gloo: the unique-id hand-out and the per-rank partition of the tensor-parallel
    constructor run on CPU; ct2x_translator_create_tp itself needs the device and must fail with the
    HIP error on every rank instead of falling back or hanging."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU box: the constructor would succeed (covered by tests/test_tensor_parallel_gpu.py)")
    script = tmp_path / "tp_worker.py"
    script.write_text(TP_WORKER)
    port = 31500 + os.getpid() % 2000
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   CT2X_TP_MODEL=str(tmp_path / "model"))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0, err[-2000:]
        outs.append(json.loads(out.strip().splitlines()[-1]))
    outs.sort(key=lambda o: o["rank"])
    assert outs[0]["uid"] == outs[1]["uid"] and set(outs[0]["uid"]) != {"0"}
    assert outs[0]["cover"] == outs[1]["cover"] == [1, 1]      # every row owned exactly once
    for o in outs:
        assert o["error"] is not None and ("hip" in o["error"].lower() or "device" in o["error"].lower()), o


def test_gpus_flag_starts_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT a torch.distributed.run environment (the way a driver may
    invoke it) must start two ranks itself, each seeing RANK / LOCAL_RANK / WORLD_SIZE = 2; and a
    launcher that started a different number of ranks than --gpus says is an error, not a silent
    one-rank run. No GPU here: every rank stops at bench.py's "needs an MI355X" line - after
    shard_info() and the world check, which is what this test is about."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1",
                        "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    # both ranks were started and both reached the GPU check
    assert p.stderr.count("bench.py needs an MI355X") >= 2, p.stderr[-2000:]
    # a launcher / flag mismatch is refused before anything else
    env2 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    q = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env2,
                       capture_output=True, text=True, timeout=120)
    assert q.returncode != 0 and "--gpus 2 but the launcher started 1 rank" in q.stderr


def _free_port():
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def test_plan_of_a_tensor_parallel_run_under_gloo():
    """`bench.py --gpus 2 --tp --tp-model nllb --plan` under torch.distributed.run with NO GPU: the two
    ranks rendezvous over gloo, the 128 bytes of rank 0 reach rank 1 the way the RCCL unique id travels,
    both take the same decisions (tensor parallel x2 over the NLLB-3.3B geometry, strong scaling, the
    communicator self-test before anything is timed) and rank 0 prints them."""
    port = _free_port()
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--tp", "--tp-model", "nllb", "--plan"],
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["ranks_agree"] and out["id_bytes_received"] == 128
    plan = out["plan"]
    assert plan["world"] == 2 and plan["parallelism"] == "tensor parallel x2" and plan["scaling"] == "strong"
    assert "NLLB" in plan["model"] and plan["batch"] == 64 and plan["beam"] == 4
    assert plan["before_the_timed_region"] and "ct2x_tp_selftest_multi" in plan["before_the_timed_region"][0]
    # replicas (the default): weak scaling, no data-path collective, the self-test all the same
    q = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plan"],
                       capture_output=True, text=True, timeout=300)
    assert q.returncode == 0, q.stderr[-2000:]
    rep = json.loads([l for l in q.stdout.splitlines() if l.startswith("{")][-1])["plan"]
    assert rep["parallelism"] == "replicas x2" and rep["scaling"] == "weak" and rep["data_path_collective"] is None
