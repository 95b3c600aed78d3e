#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection CSVs per kernel into one JSON.  usage: pmc_sum.py out.json dir [dir ...]"""
import collections, csv, glob, json, sys
out = collections.defaultdict(lambda: collections.defaultdict(float))
for d in sys.argv[2:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            out[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
            out[r["Kernel_Name"]]["launches@" + r["Counter_Name"]] += 1
json.dump({k: dict(v) for k, v in out.items() if "conv1d_mfma" in k or "residual_unit" in k or "conv1d_lat" in k}, open(sys.argv[1], "w"), indent=1)
