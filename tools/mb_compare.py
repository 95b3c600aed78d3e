#!/usr/bin/env python3
"""Side-by-side view of tools/conv_microbench.py outputs: mb_compare.py base.txt new.txt [more.txt ...] (TFLOP/s per layer; ms saved per step)."""
import re, sys

def load(f):
    d = {}
    for l in open(f):
        m = re.match(r"(\S+)\s+.*?\s+([\d.]+) ms x\s*(\d+) =\s+([\d.]+) ms\s+([\d.]+) TF", l)
        if m:
            d[m.group(1)] = (float(m.group(2)), int(m.group(3)), float(m.group(5)))
    return d

files = sys.argv[1:]
tabs = [load(f) for f in files]
tot = [0.0] * len(tabs)
for k in tabs[0]:
    if not all(k in t for t in tabs):
        continue
    ms0, c, _ = tabs[0][k]
    row = f"{k:18s} x{c:<2d}"
    for i, t in enumerate(tabs):
        ms, _, tf = t[k]
        tot[i] += ms * c
        row += f" | {ms:7.3f} ms {tf:6.1f} TF" + (f" {c * (ms0 - ms):+6.2f}" if i else "")
    print(row)
print("sum over shared layers:", "  ".join(f"{x:.1f}" for x in tot))
