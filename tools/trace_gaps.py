#!/usr/bin/env python
"""Per-kernel duration AND the idle gap in front of every dispatch, from a rocprofv3
--kernel-trace CSV (Start_Timestamp / End_Timestamp of each dispatch on the device clock).

The decode step is a chain of dependent launches: what a kernel "costs" the step is its duration
plus the gap between the previous kernel's end and its own start (dispatch latency, cache
write-back / invalidate at the boundary, and - under this profiler - its per-dispatch overhead).
Kernels are grouped by (name, grid); the summary covers the dispatches of the second half of the
trace (steady state: hipGraph replay of the decode steps).

  rocprofv3 --kernel-trace --output-format csv -d out -- python bench.py ...
  python tools/trace_gaps.py out/**/*kernel_trace.csv
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"ct2x::", "", name)
    m = re.match(r"([A-Za-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def main(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            try:
                s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            except (KeyError, ValueError):
                continue
            grid = r.get("Grid_Size_X") or r.get("Grid_Size") or ""
            wg = r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or ""
            rows.append((s, e, short(r["Kernel_Name"]), grid, wg))
    rows.sort()
    half = rows[len(rows) // 2:]
    stats = defaultdict(lambda: [0, 0.0, 0.0])
    prev_end = None
    total_dur = total_gap = 0.0
    for s, e, name, grid, wg in half:
        gap = (s - prev_end) if prev_end is not None else 0
        prev_end = max(e, prev_end or e)
        if gap > 200000 or gap < 0:  # host-side pause between calls, or overlap: not a boundary
            gap = 0
        st = stats[(name, grid, wg)]
        st[0] += 1
        st[1] += (e - s) / 1e3
        st[2] += gap / 1e3
        total_dur += (e - s) / 1e3
        total_gap += gap / 1e3
    print("%-58s %8s %6s %8s %9s %9s %7s" % ("kernel", "grid", "wg", "calls", "dur us", "gap us", "share"))
    tot = total_dur + total_gap
    for (name, grid, wg), (n, d, g) in sorted(stats.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        print("%-58s %8s %6s %8d %9.2f %9.2f %6.1f%%" % (name[:58], grid, wg, n, d / n, g / n, 100 * (d + g) / tot))
    print("dispatches %d: kernels busy %.1f ms, gaps %.1f ms (%.1f%% of the chain)" %
          (len(half), total_dur / 1e3, total_gap / 1e3, 100 * total_gap / tot))


if __name__ == "__main__":
    main(sys.argv[1])
