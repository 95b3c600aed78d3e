#!/usr/bin/env python3
"""B = 1 latency in the reference's own protocol (Evaluation/dac_vcpwq_proposed6_latency.py:300-332,489-525):
1 s of zeros at 24 kHz, 3 warm-ups, 10 repeats, device synchronised before every clock read; encode
(`encode_latents` / `DAC.encode`) and decode (`T_DEC` / `DAC.decode`) timed separately.  Published reference numbers
(unstated CUDA GPU, AMP): proposed encode 12.8-16.3 ms, decode 2.75-2.86 ms; DAC encode 1.89-3.63 ms, decode 2.86-3.10 ms."""
import json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
import multimodal_vqvae_compression_audio_tactile_amd as mvq
from multimodal_vqvae_compression_audio_tactile_amd import synth

dev = torch.device("cuda:0")
sync = torch.cuda.synchronize


def timed(fn, repeats=10):
    ts = []
    for _ in range(repeats):
        t0 = time.perf_counter(); out = fn(); sync(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.mean(ts)), float(np.min(ts)), out


res = {}
for books, K in ((1, 256), (10, 256), (8, 512)):
    sd = synth.proposed_model_state(7, rvq_books=books, rvq_embed=K, with_pe=True)
    net = mvq.build_proposed(sd, rvq_books=books, rvq_embed=K, device=dev)
    a = torch.zeros(1, 1, 24000, device=dev); t = torch.zeros(1, 1, 24000, device=dev)
    for _ in range(3):
        z = net.encode_latents(a, t, books_use=books); net.T_DEC(z)
    sync()
    enc_ms, enc_min, z = timed(lambda: net.encode_latents(a, t, books_use=books))
    dec_ms, dec_min, _ = timed(lambda: net.T_DEC(z))
    res[f"proposed_B{books}_K{K}"] = {"enc_ms": enc_ms, "enc_min_ms": enc_min, "dec_ms": dec_ms, "dec_min_ms": dec_min}
    # same calls replayed as hipGraphs (one host call each)
    from multimodal_vqvae_compression_audio_tactile_amd.graphs import GraphedCall
    g_enc = GraphedCall(lambda aa, tt: net.encode_latents(aa, tt, books_use=books), a, t)
    g_dec = GraphedCall(lambda zz: net.T_DEC(zz), z)
    zg = g_enc(a, t); sync()
    assert torch.equal(zg, z)
    e_ms, e_min, _ = timed(lambda: g_enc(a, t))
    d_ms, d_min, _ = timed(lambda: g_dec(z))
    res[f"proposed_B{books}_K{K}"].update({"graph_enc_ms": e_ms, "graph_enc_min_ms": e_min, "graph_dec_ms": d_ms, "graph_dec_min_ms": d_min})
    del net, g_enc, g_dec
mdl = mvq.DAC(); mdl.load_state_dict(synth.dac_state(7), strict=True); mdl = mdl.to(dev).eval()
x = torch.zeros(1, 1, 24000, device=dev)
for n_q in (1, 8):
    for _ in range(3):
        z, *_ = mdl.encode(x, n_quantizers=n_q); mdl.decode(z)
    sync()
    enc_ms, enc_min, zz = timed(lambda: mdl.encode(x, n_quantizers=n_q)[0])
    dec_ms, dec_min, _ = timed(lambda: mdl.decode(zz))
    res[f"dac24_nq{n_q}"] = {"enc_ms": enc_ms, "enc_min_ms": enc_min, "dec_ms": dec_ms, "dec_min_ms": dec_min}
print(json.dumps(res, indent=1))
