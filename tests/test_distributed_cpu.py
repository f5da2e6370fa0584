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
