#!/usr/bin/env python3
"""B = 1 encode (default) or decode of the latency protocol, 20 repeats, for `rocprofv3 --kernel-trace --stats`
(where do the milliseconds go?).  usage: latency_trace.py [enc|dec] [books] [embed]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import multimodal_vqvae_compression_audio_tactile_amd as mvq
from multimodal_vqvae_compression_audio_tactile_amd import synth
what = sys.argv[1] if len(sys.argv) > 1 else "enc"
books = int(sys.argv[2]) if len(sys.argv) > 2 else 1
embed = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device("cuda:0")
net = mvq.build_proposed(synth.proposed_model_state(7, rvq_books=books, rvq_embed=embed), rvq_books=books, rvq_embed=embed, device=dev)
a = torch.zeros(1, 1, 24000, device=dev); t = torch.zeros(1, 1, 24000, device=dev)
z = net.encode_latents(a, t)
f = (lambda: net.encode_latents(a, t)) if what == "enc" else (lambda: net.T_DEC(z))
for _ in range(3): f()
torch.cuda.synchronize()
for _ in range(20): f()
torch.cuda.synchronize()
