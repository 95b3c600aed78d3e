#!/usr/bin/env python3
"""A/B of the AR loop alone: the persistent kernel (csrc/ar_fused.hip), the same stages as one host call of stand-alone launches
(mvq_ar_latents_staged_f32) and the Python loop, by batch size.
   usage (GPU box): python3 tools/ar_fused_ab.py [B ...]   -> one JSON line per batch size (ms per call, HIP events, 20 calls)"""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from multimodal_vqvae_compression_audio_tactile_amd import build_proposed, synth  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n


def main():
    dev = torch.device("cuda:0")
    sd = synth.proposed_model_state(3, rvq_books=8, rvq_embed=512)
    net = build_proposed(sd, rvq_books=8, rvq_embed=512, device=dev)
    for B in [int(v) for v in sys.argv[1:]] or [1, 2, 4, 6, 8, 12, 16, 24]:
        g = torch.Generator().manual_seed(B)
        zt = (torch.randn(B, 1024, 75, generator=g) * 0.7).to(dev)
        qa = (torch.randn(B, 1024, 75, generator=g) * 0.7).to(dev)
        row = {"batch": B}
        outs = {}
        for name, fused_cap, staged_cap in (("fused_ms", 1 << 20, 0), ("staged_one_host_call_ms", 0, 8), ("python_loop_ms", 0, 0)):
            if name.startswith("staged") and B > 8:
                continue
            net.AR_FUSED_MAX_BATCH, net.AR_STAGED_MAX_BATCH = fused_cap, staged_cap
            row[name] = round(timed(lambda: net._ar_latents(qa, zt)), 4)
            outs[name] = net._ar_latents(qa, zt)[0]
        row["bit_equal"] = all(bool(torch.equal(v, outs["python_loop_ms"])) for v in outs.values())
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
