#!/usr/bin/env python3
"""Print the headline numbers and the per-kernel table of one or two bench.py JSON lines (second = baseline to diff against)."""
import json, sys
def load(f): return json.loads(open(f).read().strip().splitlines()[-1])
d = load(sys.argv[1]); b = load(sys.argv[2]) if len(sys.argv) > 2 else None
print(f"value {d['value']:.0f} {d['unit']}  ms/step {d['ms_per_step']:.2f}  path_tflops {d.get('path_tflops', 0):.2f}" +
      (f"   (baseline {b['ms_per_step']:.2f} ms, {b.get('path_tflops', 0):.2f})" if b else ""))
if "roofline" in d: print("roofline", d["roofline"]["kernel"], f"{d['roofline']['achieved']:.1f} frac {d['roofline']['frac']:.3f}")
if "latency_b1" in d: print("latency", {k: round(v, 3) for k, v in d["latency_b1"].items() if isinstance(v, float)})
ks = d.get("conv_stack", {}).get("kernels", {}); bk = (b or {}).get("conv_stack", {}).get("kernels", {})
tot = 0.0
for k, v in ks.items():
    tot += v["ms_per_step"]
    o = bk.get(k)
    print(f"{k[:52]:52s} {v['tflops']:7.1f} TF {v['ms_per_step']:7.2f} ms x{v['launches_per_step']:3d}" + (f"   was {o['tflops']:7.1f} TF {o['ms_per_step']:7.2f} ms" if o else ""))
print("conv total ms", round(tot, 2), "step", round(d["ms_per_step"], 2))
