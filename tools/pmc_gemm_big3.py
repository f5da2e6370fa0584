#!/usr/bin/env python
"""Which unit the MFMA-bound int8 GEMM (`gemm_s8_big*_kernel`, fused dequantise + Swish form, the
Llama-3-8B prefill FFN-in shape [2048, 14336] x 4096 by default) keeps busy: rocprofv3 PMC passes
the way /opt/skills/guides/MI355X_MICROARCH.md prescribes (every pass `--pmc ... --kernel-trace`
only; FETCH_SIZE and WRITE_SIZE in passes of their own; SQ counters at most 8 per pass).

Derived per dispatch (median over the bench's timed launches):
  mfma_busy_pct   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x CUs x 4 SIMDs)   (gfx94x MfmaUtil formula;
                    rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs: GUI / 8 / duration is the shader clock)
  mfma_issue_pct  = SQ_INSTS_MFMA x 16 cycles / (GRBM_GUI_ACTIVE / 8 x CUs x 4)        (v_mfma_i32_16x16x64_i8 = 16 cycles)
  valu_busy_pct   = SQ_ACTIVE_INST_VALU x 4 (quad-cycles) / (GRBM_GUI_ACTIVE / 8 x CUs x 4)
  clock_ghz       = GRBM_GUI_ACTIVE / 8 / kernel duration
  the wave-cycle split WAIT_ANY / WAIT_INST_ANY / ACTIVE_INST_ANY (quad-cycles, disjoint)
  LDS: SQ_WAIT_INST_LDS, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  traffic_bytes   = 2 x FETCH_SIZE + WRITE_SIZE (KiB counters; gfx950 tallies the 128-byte requests
                    of 16-byte-per-lane loads at 64 B)
  python tools/pmc_gemm_big3.py outdir [--act N] [M N K]"""
import csv
import glob
import json
import os
import re
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [
    ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAVE_CYCLES",
     "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"],
    ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
     "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS", "GRBM_GUI_ACTIVE"],
    ["SQ_INSTS_MFMA", "SQ_INSTS_VALU_MFMA_MOPS_I8", "SQ_ACTIVE_INST_MISC", "SQ_INSTS_SALU", "SQ_INSTS_VMEM",
     "SQ_ACTIVE_INST_VMEM", "SQ_WAVES", "GRBM_GUI_ACTIVE"],
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
    ["TCC_HIT_sum", "TCC_MISS_sum"],
]


def one_pass(counters, outdir, gb, gb_args):
    os.makedirs(outdir, exist_ok=True)
    cmd = ["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", outdir, "--", gb] + gb_args
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True)
    vals, durs = {}, {}
    for path in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                m = re.search(r"gemm_s8_big\w*_kernel<[^>]*>", row["Kernel_Name"])
                if not m:
                    continue
                vals.setdefault((m.group(0), row["Counter_Name"]), []).append(float(row["Counter_Value"]))
    for path in glob.glob(os.path.join(outdir, "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                m = re.search(r"gemm_s8_big\w*_kernel<[^>]*>", row["Kernel_Name"])
                if m:
                    durs.setdefault(m.group(0), []).append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    if not vals:
        return {"_error": r.stdout[-600:]}, durs
    return {k: statistics.median(v) for k, v in vals.items()}, durs


def main():
    out = os.path.abspath(sys.argv[1])
    rest = sys.argv[2:]
    gb_args = rest if rest else ["--act", "2", "2048", "14336", "4096"]
    gb = os.path.join(ROOT, "build", "tools", "gemm_bench")
    if not os.path.exists(gb):
        gb = "/tmp/gb"
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-x", "hip",
                               os.path.join(ROOT, "tools", "gemm_bench.cpp"), "-I" + os.path.join(ROOT, "include"),
                               "-L" + os.path.join(ROOT, "ctranslate2_amd"), "-lct2_mi355",
                               "-Wl,-rpath," + os.path.join(ROOT, "ctranslate2_amd"), "-o", gb])
    res = {"_note": " ".join(__doc__.split("\n\n")[0].split()), "command": "gemm_bench " + " ".join(gb_args),
           "kernels": {}}
    for i, counters in enumerate(PASSES):
        vals, durs = one_pass(counters, os.path.join(out, "pass%d" % i), gb, gb_args)
        if "_error" in vals:
            res.setdefault("_failed_passes", []).append({"counters": counters, "tail": vals["_error"]})
            continue
        for (kernel, counter), v in vals.items():
            k = res["kernels"].setdefault(kernel, {"counters": {}, "duration_ns_by_pass": {}})
            k["counters"][counter if counter != "GRBM_GUI_ACTIVE" else "GRBM_GUI_ACTIVE@pass%d" % i] = v
            if kernel in durs:
                k["duration_ns_by_pass"]["pass%d" % i] = statistics.median(durs[kernel])
    cus, xcds = 256, 8
    for kernel, k in res["kernels"].items():
        c = k["counters"]
        d = {}
        gui0 = c.get("GRBM_GUI_ACTIVE@pass0")
        if gui0 and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            d["mfma_busy_pct"] = 100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui0 / xcds * cus * 4)
            if "pass0" in k["duration_ns_by_pass"]:
                d["clock_ghz_pass0"] = gui0 / xcds / k["duration_ns_by_pass"]["pass0"]
        gui2 = c.get("GRBM_GUI_ACTIVE@pass2")
        if gui2 and "SQ_INSTS_MFMA" in c:
            d["mfma_issue_pct_16cyc"] = 100.0 * c["SQ_INSTS_MFMA"] * 16 / (gui2 / xcds * cus * 4)
        gui1 = c.get("GRBM_GUI_ACTIVE@pass1")
        if gui1 and "SQ_ACTIVE_INST_VALU" in c:
            d["valu_busy_pct"] = 100.0 * c["SQ_ACTIVE_INST_VALU"] * 4 / (gui1 / xcds * cus * 4)
        if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c and c["SQ_WAVES"]:
            d["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
        if "SQ_WAIT_ANY" in c and "SQ_WAIT_INST_ANY" in c and "SQ_ACTIVE_INST_ANY" in c:
            tot = c["SQ_WAIT_ANY"] + c["SQ_WAIT_INST_ANY"] + c["SQ_ACTIVE_INST_ANY"]
            d["wave_cycle_split_pct"] = {n: round(100.0 * c[n] / tot, 1) for n in
                                         ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")}
        if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_conflict_over_active_pct"] = 100.0 * c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
        if "SQ_WAIT_INST_LDS" in c and "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
            d["wait_inst_lds_over_wave_cycles_pct"] = 100.0 * c["SQ_WAIT_INST_LDS"] / c["SQ_WAVE_CYCLES"]
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            d["traffic_bytes"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            d["l2_hit_pct"] = 100.0 * c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        k["derived"] = d
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
