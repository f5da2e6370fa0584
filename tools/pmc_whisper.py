#!/usr/bin/env python
"""HBM-side bytes per launch of the 16-bit kernels of the Whisper-large-v3-shaped leg
(tools/bench_whisper.py: 8 windows x 30 s, beam 5, bfloat16) from PMC counters, collected as
/opt/skills/guides/MI355X_MICROARCH.md prescribes: `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and
`--pmc WRITE_SIZE --kernel-trace` in SEPARATE passes; counters in KiB; gfx950 correction: FETCH_SIZE
tallies the 128-byte requests of 16-byte-per-lane streaming loads at 64 B (x2). Next to each kernel
its algorithmic bytes per launch (what one pass over its operands moves).

  python tools/pmc_whisper.py gpurun_out/r5/pmc_whisper > profiles/r05_pmc_whisper.json
"""
import csv
import glob
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, BEAM, S, D, H, FFN = 8, 5, 1500, 1280, 20, 5120
ROWS = B * BEAM
KERNELS = {
    # fragment of the kernel name -> (label, algorithmic bytes per launch)
    "xattn_h_kernel": ("xattn_h_kernel: encoder-decoder attention of a decode step (memory K/V in bfloat16)",
                       B * S * 2 * D * 2 + 2 * ROWS * D * 4),
    "attn_prefill_h_kernel": ("attn_prefill_h_kernel: encoder self-attention (fp32 Q, K, V in, context out)",
                              4 * B * S * D * 4),
    # the six Dense launches of a decoder layer differ in size (3.3 - 13.1 MB of weights): compared as
    # their MEAN - one sixth of a layer's weights + activations in and out + the K-slice partial tiles
    # (5 / 10 slices written through and read back once)
    "gemm_h_rows_kernel": ("gemm_h_rows_kernel: decode-row Dense, 40 rows (mean over the six launches of a decoder layer)",
                           ((4 * D * D + 2 * D * D + 2 * D * FFN) * 2                       # weights: qkv + out, cross q + out, ffn
                            + ROWS * 4 * (5 * D + FFN)                                       # activations in
                            + ROWS * 4 * (3 * D + 3 * D + FFN + D)                           # results out
                            + 2 * ROWS * 4 * (5 * (3 * D + 3 * D + FFN) + 10 * D)) // 6),    # partial tiles, out and back
}
MEAN = {"gemm_h_rows_kernel"}


def one_pass(counter, outdir):
    outdir = os.path.abspath(outdir)
    os.makedirs(outdir, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.check_call(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", outdir,
                           "--", sys.executable, os.path.join(ROOT, "tools", "bench_whisper.py"), "--steps", "1",
                           "--warmup", "0", "--new-tokens", "16"], cwd="/tmp", env=env,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    values = {}
    for path in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if r.get("Counter_Name") != counter:
                    continue
                for frag in KERNELS:
                    if frag in r["Kernel_Name"]:
                        values.setdefault(frag, []).append(float(r["Counter_Value"]))
    return values


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r5/pmc_whisper"
    fetch = one_pass("FETCH_SIZE", os.path.join(out, "fetch"))
    write = one_pass("WRITE_SIZE", os.path.join(out, "write"))
    result = {"_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace) over "
                       "tools/bench_whisper.py --steps 1 --warmup 0 --new-tokens 16; median per dispatch (mean where the label says so); raw counters "
                       "in KiB; traffic_bytes = 2 x FETCH (gfx950: 128-byte requests of 16-byte-per-lane loads counted "
                       "at 64 B) + WRITE; memory-side requests (Infinity-Cache hits included)"}
    for frag, (label, algo) in KERNELS.items():
        if frag in fetch and frag in write:
            mid = statistics.mean if frag in MEAN else statistics.median
            f, w = mid(fetch[frag]), mid(write[frag])
            result[label] = {"fetch_kib_raw": f, "write_kib_raw": w, "dispatches": len(fetch[frag]),
                             "traffic_bytes": (2 * f + w) * 1024, "algorithmic_bytes": algo,
                             "traffic_over_algorithmic": (2 * f + w) * 1024 / algo}
    print(json.dumps(result, indent=1))


if __name__ == "__main__":
    main()
