#!/usr/bin/env python3
"""Batch independence of the exact path, stage by stage: the first segment(s) of a batch of B must give the same bits whatever B is
(bench.py's batch_sweep enforces it on forward_eval's output; this says WHERE a difference starts).
   usage (GPU box): python3 tools/batch_invariance.py [B ...]      (default 1 6 64)"""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from multimodal_vqvae_compression_audio_tactile_amd import build_proposed, synth  # noqa: E402


def stages(net, a, t):
    za = net.A_ENC(a)
    qa = net.A_QUANT(za)[0]
    zt = net.T_ENC(t)
    z_run = net._ar_latents(qa, zt, None)[0]
    y = net.T_DEC(z_run)
    return {"za": za, "qa": qa, "zt": zt, "z_run": z_run, "y": y}


def main():
    dev = torch.device("cuda:0")
    sd = synth.proposed_model_state(7, rvq_books=8, rvq_embed=512)
    net = build_proposed(sd, rvq_books=8, rvq_embed=512, device=dev)
    sizes = [int(v) for v in sys.argv[1:]] or [1, 6, 64]
    a_all = synth.audio_segments(max(sizes), seed=7).to(dev)
    t_all = synth.tactile_segments(max(sizes), seed=7).to(dev)
    with torch.no_grad():
        outs = {B: stages(net, a_all[:B].contiguous(), t_all[:B].contiguous()) for B in sizes}
    # the two-stream form of the branches (proposed.py: _encode_branches) against the one-stream stages above, call after call
    for B in sizes:
        a, t = a_all[:B].contiguous(), t_all[:B].contiguous()
        bad = {"qa": 0, "zt": 0, "y": 0}
        first_bad = None
        with torch.no_grad():
            for rep in range(12):
                qa, zt = net._encode_branches(a, t)
                y = net.forward_eval(a, t)
                for k, v in (("qa", qa), ("zt", zt), ("y", y)):
                    if not torch.equal(v, outs[B][k]):
                        bad[k] += 1
                        if first_bad is None:
                            w = outs[B][k]
                            first_bad = f"rep {rep} {k}: {int((v != w).sum())} of {v.numel()} differ, max |d| {float((v - w).abs().max()):.3e}"
        print(json.dumps({"two_stream_vs_one_stream": f"B={B}", "calls": 12, "mismatching_calls": bad, "first": first_bad}), flush=True)
    base = sizes[0]
    for B in sizes[1:]:
        row = {"compare": f"B={base} rows vs B={B}"}
        for k, v in outs[base].items():
            w = outs[B][k][:base]
            row[k] = "equal" if torch.equal(v, w) else f"{int((v != w).sum())} of {v.numel()} differ (max |d| {float((v - w).abs().max()):.3e})"
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
