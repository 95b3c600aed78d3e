#!/usr/bin/env python3
"""Effective shader clock per kernel from one rocprofv3 pass: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / the dispatch's duration
(kernel trace of the same run).  usage: pmc_clock.py <rocprofv3 output dir>   -- prints kernel, launches, mean us, GHz, MFMA-busy share."""
import collections, csv, glob, re, sys
d = sys.argv[1]
dur = {}
for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        did = r["Dispatch_Id"]
        if did not in dur:
            continue
        n = re.sub(r"^void\s+", "", dur[did][1]).replace("mvq::", ""); n = re.sub(r"\(.*$", "", n)
        a = acc[n]
        a[r["Counter_Name"]] += float(r["Counter_Value"])
        a["ns@" + r["Counter_Name"]] += dur[did][0]; a["n@" + r["Counter_Name"]] += 1
for n, a in sorted(acc.items(), key=lambda kv: -kv[1].get("ns@GRBM_GUI_ACTIVE", 0)):
    if "GRBM_GUI_ACTIVE" not in a or a["ns@GRBM_GUI_ACTIVE"] < 2e6:
        continue
    ghz = a["GRBM_GUI_ACTIVE"] / 8.0 / a["ns@GRBM_GUI_ACTIVE"]
    busy = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * a["GRBM_GUI_ACTIVE"] / 8.0) if "SQ_VALU_MFMA_BUSY_CYCLES" in a else float("nan")
    print(f"{n[:64]:64s} x{int(a['n@GRBM_GUI_ACTIVE']):3d}  {a['ns@GRBM_GUI_ACTIVE'] / a['n@GRBM_GUI_ACTIVE'] / 1e3:8.1f} us  clock {ghz:5.2f} GHz  MFMA busy {busy:5.2f}")
