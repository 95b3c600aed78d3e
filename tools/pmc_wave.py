#!/usr/bin/env python3
"""Per-kernel wave-state breakdown from rocprofv3 --pmc passes (SQ counters).  usage: pmc_wave.py dir [dir ...]"""
import collections, csv, glob, re, sys
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = re.sub(r"^void\s+", "", r["Kernel_Name"]).replace("mvq::", ""); n = re.sub(r"\(.*$", "", n)
            tot[n][r["Counter_Name"]] += float(r["Counter_Value"]); tot[n]["n@" + r["Counter_Name"]] += 1
for k, v in tot.items():
    if "mfma" not in k and "residual" not in k: continue
    wc = v.get("SQ_WAVE_CYCLES", 0)
    print(k)
    for c in sorted(v):
        if c.startswith("n@"): continue
        print(f"   {c:28s} {v[c]:.4g}  per launch {v[c] / v['n@' + c]:.4g}" + (f"   /WAVE_CYCLES {v[c] / wc:.3f}" if wc and c.startswith('SQ_') else ""))
