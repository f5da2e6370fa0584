#!/usr/bin/env python
"""HBM traffic of the decode GEMMs at the Llama shapes (persistent streaming kernel, weights rotated
through 600 MB) from rocprofv3 PMC counters, collected as /opt/skills/guides/MI355X_MICROARCH.md
prescribes: `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in SEPARATE passes with --kernel-trace only;
gfx950 counts the 128-byte requests of 16-byte-per-lane loads at 64 B, so traffic = 2 x FETCH + WRITE
(KiB counters). Prints one JSON object: per (kernel, grid) the median bytes per launch next to the
algorithmic bytes of that shape.
  python tools/pmc_stream_gemm.py outdir"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(16, 14336, 4096), (16, 4096, 14336), (16, 6144, 4096), (16, 4096, 4096)]


def one_pass(counter, outdir):
    os.makedirs(outdir, exist_ok=True)
    args = [str(v) for s in SHAPES for v in s]
    subprocess.check_call(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d",
                           outdir, "--", "/tmp/gb"] + args, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"),
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    files = glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True)
    vals = {}
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") == counter and "stream_kernel" in r["Kernel_Name"]:
                import re
                name = re.search(r"gemm_s8_stream_kernel<[^>]*>", r["Kernel_Name"]).group(0)
                vals.setdefault((name, r.get("Grid_Size", "")), []).append(float(r["Counter_Value"]))
    return {k: sorted(v)[len(v) // 2] for k, v in vals.items()}


def main():
    out = os.path.abspath(sys.argv[1])
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-x", "hip",
                           os.path.join(ROOT, "tools", "gemm_bench.cpp"), "-I" + os.path.join(ROOT, "include"),
                           "-L" + os.path.join(ROOT, "ctranslate2_amd"), "-lct2_mi355",
                           "-Wl,-rpath," + os.path.join(ROOT, "ctranslate2_amd"), "-o", "/tmp/gb"],
                          stderr=subprocess.DEVNULL)
    fetch = one_pass("FETCH_SIZE", os.path.join(out, "fetch"))
    write = one_pass("WRITE_SIZE", os.path.join(out, "write"))
    res = {"_note": __doc__.split("\n\n")[0].replace("\n", " "), "kernels": []}
    for k in fetch:
        res["kernels"].append({"kernel": k[0], "grid": k[1], "fetch_kib_raw": fetch[k],
                               "write_kib_raw": write.get(k), "traffic_bytes": 2 * fetch[k] * 1024 +
                               (write.get(k) or 0) * 1024})
    res["algorithmic_bytes"] = {"%dx%dx%d" % s: s[0] * s[2] + s[1] * s[2] + 4 * s[1] + 4 * s[0] + 4 * s[0] * s[1] + 4 * s[1]
                                for s in SHAPES}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
