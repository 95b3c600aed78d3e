#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as the MI355X guide prescribes) of bench.py
into profiles/pmc_traffic.json: HBM bytes per launch for every kernel.  gfx950 corrections: counters are in KiB;
FETCH_SIZE reports half of the bytes of wide (16 B/lane) coalesced reads -> doubled; WRITE_SIZE is exact for 16-byte
streaming stores.   usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json>"""
import collections, csv, glob, json, re, sys


def load(d, counter):
    f = sorted(glob.glob(f"{d}/**/*counter_collection.csv", recursive=True))[-1]
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"^void\s+", "", r["Kernel_Name"]).replace("mvq::", "")
        name = re.sub(r"\(.*$", "", name)
        tot[name] += float(r["Counter_Value"]); cnt[name] += 1
    return tot, cnt


fetch, nf = load(sys.argv[1], "FETCH_SIZE")
write, nw = load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in fetch:
    n = nf[k]
    rd = 2.0 * fetch[k] * 1024 / n
    wr = write.get(k, 0.0) * 1024 / max(nw.get(k, 1), 1)
    out[k] = {"launches": n, "fetch_bytes_per_launch_corrected": rd, "write_bytes_per_launch": wr,
              "hbm_bytes_per_launch": rd + wr}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]:
    print(f"{k[:70]:70s} x{v['launches']:4d}  rd {v['fetch_bytes_per_launch_corrected']/1e6:9.1f} MB  wr {v['write_bytes_per_launch']/1e6:9.1f} MB")
