#!/usr/bin/env python
"""HBM bytes per launch of the decode-shape kernels from PMC counters, the way
/opt/skills/guides/MI355X_MICROARCH.md prescribes: `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and
`--pmc WRITE_SIZE --kernel-trace` in SEPARATE passes over tools/kernel_bench.cpp (never combined
with sys / hip / hsa tracing), counters in KiB, gfx950 correction: FETCH_SIZE counts the 128-byte
requests of 16-byte-per-lane streaming loads at 64 B (x2; calibrated on the 32 x 32000 x 512 GEMM,
whose 16.4 MB of weights read 0.505x).

Run on the GPU box from the repo root:  python tools/pmc_traffic.py gpurun_out/pmc_r02 > profiles/r02_pmc_traffic.json
"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M = 32
# kernel_bench's launch sequence (one entry per run() call) -> the engine's profile label
SEQUENCE = [
    "quantize_s8[m=32,n=0,k=512]", "quantize_s8[m=32,n=0,k=2048]", "norm_quantize[m=32,n=0,k=512]",
    "norm[m=32,n=0,k=512]",
    "gemm_s8_dequant[m=32,n=512,k=512]", "gemm_s8_dequant[m=32,n=1536,k=512]",
    "gemm_s8_dequant[m=32,n=2048,k=512]", "gemm_s8_dequant[m=32,n=512,k=2048]",
    "gemm_s8_dequant[m=32,n=32000,k=512]",
    "gemm_s8_dequant[m=32,n=512,k=512]/rotating", "gemm_s8_dequant[m=32,n=2048,k=512]/rotating",
    "norm_gemm_s8_dequant[m=32,n=512,k=512]", "norm_gemm_s8_dequant[m=32,n=1536,k=512]",
    "norm_gemm_s8_dequant[m=32,n=2048,k=512]",
    "gemm_s8_dequant[m=32,n=512,k=512]/residual",
    "attention_self_decode_q[m=32,n=8,k=64]/T=1", "attention_self_decode_q[m=32,n=8,k=64]/T=16",
    "attention_self_decode_q[m=32,n=8,k=64]/T=32", "attention_self_decode_q[m=32,n=8,k=64]/T=64",
    "softmax_log[m=32,depth=32000]",
]


def one_pass(counter, outdir):
    outdir = os.path.abspath(outdir)  # rocprofv3 runs from /tmp
    os.makedirs(outdir, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.check_call(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv",
                           "-d", outdir, "--", "/tmp/kb", str(M)], cwd="/tmp", env=env,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    files = glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True)
    assert files, "no counter_collection.csv under " + outdir
    rows = []
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") == counter and "plug" not in r["Kernel_Name"]:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], r.get("Grid_Size", ""),
                             float(r["Counter_Value"])))
    rows.sort()
    # runs of consecutive dispatches of one (kernel, grid): one run() call each (>= 400 launches)
    runs = []
    for _, name, grid, val in rows:
        if runs and runs[-1][0] == (name, grid):
            runs[-1][1].append(val)
        else:
            runs.append([(name, grid), [val]])
    # one run() call = 5 warm-up + 400 timed launches; consecutive calls of the same (kernel, grid)
    # (norm with / without Quantize, the four cache lengths of the attention) are split again
    out = []
    for k, v in runs:
        if len(v) < 300:
            continue  # (topk alternates two kernels: runs of one)
        per = 405
        if len(v) % per == 0:
            for i in range(0, len(v), per):
                out.append((k, v[i:i + per]))
        else:
            out.append((k, v))
    return out


def median(v):
    s = sorted(v)
    return s[len(s) // 2]


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pmc_r02")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-x", "hip",
                           os.path.join(ROOT, "tools", "kernel_bench.cpp"), "-I" + os.path.join(ROOT, "include"),
                           "-L" + os.path.join(ROOT, "ctranslate2_amd"), "-lct2_mi355",
                           "-Wl,-rpath," + os.path.join(ROOT, "ctranslate2_amd"), "-o", "/tmp/kb"],
                          stderr=subprocess.DEVNULL)
    fetch = one_pass("FETCH_SIZE", os.path.join(out, "fetch"))
    write = one_pass("WRITE_SIZE", os.path.join(out, "write"))
    result = {"_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace) on "
                       "tools/kernel_bench.cpp at 32 rows, median over >= 400 dispatches per shape; raw "
                       "counters in KiB; traffic_bytes = 2 x FETCH (gfx950: 128-byte requests of "
                       "16-byte-per-lane loads are counted at 64 B) + WRITE. Labels are the engine's "
                       "profile labels; '/...' suffixes name kernel_bench variants of the same launch.",
              "_kernels": [k[0][0][:80] + " grid " + k[0][1] for k in fetch]}
    if len(fetch) != len(SEQUENCE) or len(write) != len(SEQUENCE):
        result["_warning"] = "found %d / %d runs, expected %d: labels not assigned" % (
            len(fetch), len(write), len(SEQUENCE))
        result["_fetch_kib"] = [median(v) for _, v in fetch]
        result["_write_kib"] = [median(v) for _, v in write]
    else:
        for label, (_, fv), (_, wv) in zip(SEQUENCE, fetch, write):
            f_kib, w_kib = median(fv), median(wv)
            result[label] = {"fetch_kib_raw": f_kib, "write_kib_raw": w_kib,
                             "traffic_bytes": (2 * f_kib + w_kib) * 1024}
    print(json.dumps(result, indent=1))


if __name__ == "__main__":
    main()
