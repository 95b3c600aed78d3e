#!/usr/bin/env python3
"""Where does a small-batch step spend its time?  Reads a rocprofv3 kernel_trace.csv (one row per dispatch with start / end
timestamps) and prints, for the last `--calls` repetitions of the traced call: busy time per kernel name, the idle time between
consecutive dispatches (launch gaps), and the longest dispatches.  usage: trace_gaps.py <kernel_trace.csv> [--skip-frac 0.3]"""
import csv, json, sys, re
from collections import defaultdict

def short(n):
    n = re.sub(r"^void\s+", "", n)
    n = re.sub(r"\(.*$", "", n)
    n = n.replace("mvq::", "").replace("(anonymous namespace)::", "")
    return n[:110]

def main():
    path = sys.argv[1]
    skip = 0.3
    if "--skip-frac" in sys.argv:
        skip = float(sys.argv[sys.argv.index("--skip-frac") + 1])
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 0)) or 0), int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)))
    rows.sort()
    rows = rows[int(len(rows) * skip):]            # drop model load / warm-up dispatches
    busy = defaultdict(lambda: [0, 0])
    gap_total, span = 0, rows[-1][1] - rows[0][0]
    last_end = rows[0][0]
    gaps = []
    for s, e, n, wg, grid in rows:
        busy[short(n)][0] += e - s
        busy[short(n)][1] += 1
        if s > last_end:
            gap_total += s - last_end
            gaps.append(s - last_end)
        last_end = max(last_end, e)
    tot_busy = sum(v[0] for v in busy.values())
    out = {"dispatches": len(rows), "span_ms": span * 1e-6, "sum_kernel_ms": tot_busy * 1e-6, "idle_between_dispatches_ms": gap_total * 1e-6,
           "idle_frac_of_span": gap_total / span, "median_gap_us": sorted(gaps)[len(gaps) // 2] * 1e-3 if gaps else 0.0,
           "kernels": [{"kernel": k, "ms": v[0] * 1e-6, "calls": v[1], "avg_us": v[0] / v[1] * 1e-3, "share_of_busy": v[0] / tot_busy}
                       for k, v in sorted(busy.items(), key=lambda kv: -kv[1][0])[:40]]}
    print(json.dumps(out, indent=1))

if __name__ == "__main__":
    main()
