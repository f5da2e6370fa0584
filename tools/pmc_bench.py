#!/usr/bin/env python
"""HBM-side bytes per launch of the kernels of the BENCH'S OWN timed call (round 5: the
decoder-stack launch and the vocabulary projection with the fused arg-max), from PMC counters the
way /opt/skills/guides/MI355X_MICROARCH.md prescribes: `rocprofv3 --pmc FETCH_SIZE --kernel-trace`
and `--pmc WRITE_SIZE --kernel-trace` in SEPARATE passes (never combined with sys / hip / hsa
tracing) over `python bench.py --steps 2 --warmup 1` without the other legs; counters in KiB;
gfx950 correction: FETCH_SIZE tallies the 128-byte requests of 16-byte-per-lane streaming loads at
64 B (x2). The memory-side counters include Infinity-Cache hits and, for the decoder-stack launch,
the hand-over traffic (sc1 stores drop the line from L2, so every poll of a slot is a memory-side
request).

Run on the GPU box from the repo root:  python tools/pmc_bench.py gpurun_out/r5/pmc > profiles/r05_pmc_traffic.json
"""
import csv
import glob
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# kernel-name fragment -> the engine's profile label (bench.py roofline / pmc_traffic)
LABELS = {
    "decoder_cluster_kernel": "decoder_stack[m=32,n=512,k=512]",
    "false, true>(signed char": "gemm_s8_dequant[m=32,n=32000,k=512]",  # skinny kernel, ARGMAX form
}


def one_pass(counter, outdir):
    outdir = os.path.abspath(outdir)
    os.makedirs(outdir, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.check_call(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", outdir,
                           "--", sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1",
                           "--no-cpu-baseline", "--no-profile", "--no-l8", "--no-whisper", "--no-nllb",
                           "--no-token-check"], cwd="/tmp", env=env, stdout=subprocess.DEVNULL,
                          stderr=subprocess.DEVNULL)
    values = {}
    for path in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if r.get("Counter_Name") != counter:
                    continue
                for frag, label in LABELS.items():
                    if frag in r["Kernel_Name"]:
                        values.setdefault(label, []).append(float(r["Counter_Value"]))
    return values


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r5/pmc"
    fetch = one_pass("FETCH_SIZE", os.path.join(out, "fetch"))
    write = one_pass("WRITE_SIZE", os.path.join(out, "write"))
    result = {"_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace) over bench.py's own "
                       "timed calls (Transformer-base int8, batch 32, 64 steps), median per dispatch; raw counters in "
                       "KiB; traffic_bytes = 2 x FETCH (gfx950: 128-byte requests of 16-byte-per-lane loads counted "
                       "at 64 B) + WRITE; memory-side requests (Infinity-Cache hits and hand-over polls included)"}
    for label in LABELS.values():
        if label in fetch and label in write:
            f, w = statistics.median(fetch[label]), statistics.median(write[label])
            result[label] = {"fetch_kib_raw": f, "write_kib_raw": w, "dispatches": len(fetch[label]),
                             "fetch_kib_min_max": [min(fetch[label]), max(fetch[label])],
                             "traffic_bytes": (2 * f + w) * 1024}
    print(json.dumps(result, indent=1))


if __name__ == "__main__":
    main()
