#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as the MI355X guide prescribes) of bench.py
into profiles/pmc_traffic.json: HBM bytes per launch for every kernel.  gfx950 corrections: counters are in KiB;
FETCH_SIZE reports half of the bytes of wide (16 B/lane) coalesced reads -> doubled; WRITE_SIZE is exact for 16-byte
streaming stores.  The table is stamped (``_meta``) with the commit it was profiled at and a digest of csrc/, so that
bench.py can say whether the kernels changed since.

  usage: pmc_traffic.py <fetch_dir> <write_dir> <out.json> [commit] [command...]"""
import collections
import csv
import glob
import hashlib
import json
import re
import sys
from pathlib import Path


def csrc_digest(root) -> str:
    """sha256 (first 16 hex digits) over the kernel sources, in name order."""
    h = hashlib.sha256()
    d = Path(root) / "multimodal_vqvae_compression_audio_tactile_amd" / "csrc"
    for f in sorted(list(d.glob("*.hip")) + list(d.glob("*.hpp"))):
        h.update(f.name.encode()); h.update(f.read_bytes())
    return h.hexdigest()[:16]


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


def _flag(cmd, name, default):
    return cmd[cmd.index(name) + 1] if name in cmd and cmd.index(name) + 1 < len(cmd) else default


def main(argv):
    fetch, nf = load(argv[1], "FETCH_SIZE")
    write, nw = load(argv[2], "WRITE_SIZE")
    root = Path(__file__).resolve().parent.parent
    out = {"_meta": {"commit": argv[4] if len(argv) > 4 else None, "csrc_sha16": csrc_digest(root),
                     "command": " ".join(argv[5:]) or None,
                     # what the per-launch bytes describe: bench.py only quotes them for a run of the SAME workload / batch / arithmetic
                     "workload": _flag(argv[5:], "--workload", "joint"), "batch": int(_flag(argv[5:], "--batch", "256")),
                     "arith": _flag(argv[5:], "--arith", "f32"),
                     "corrections": "KiB -> bytes; FETCH_SIZE x2 (gfx950 wide coalesced reads); WRITE_SIZE as is"}}
    for k in fetch:
        n = nf[k]
        rd = 2.0 * fetch[k] * 1024 / n
        wr = write.get(k, 0.0) * 1024 / max(nw.get(k, 1), 1)
        out[k] = {"launches": n, "fetch_bytes_per_launch_corrected": rd, "write_bytes_per_launch": wr,
                  "hbm_bytes_per_launch": rd + wr}
    json.dump(out, open(argv[3], "w"), indent=1)
    rows = [(k, v) for k, v in out.items() if k != "_meta"]
    for k, v in sorted(rows, key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]:
        print(f"{k[:70]:70s} x{v['launches']:4d}  rd {v['fetch_bytes_per_launch_corrected']/1e6:9.1f} MB  wr {v['write_bytes_per_launch']/1e6:9.1f} MB")


if __name__ == "__main__":
    main(sys.argv)
