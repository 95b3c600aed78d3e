#!/usr/bin/env python3
"""B = 1 encode of the latency protocol, 20 repeats, for `rocprofv3 --kernel-trace --stats` (where do the ~5 ms go?)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import multimodal_vqvae_compression_audio_tactile_amd as mvq
from multimodal_vqvae_compression_audio_tactile_amd import synth
dev = torch.device("cuda:0")
net = mvq.build_proposed(synth.proposed_model_state(7, rvq_books=1, rvq_embed=256), rvq_books=1, rvq_embed=256, device=dev)
a = torch.zeros(1, 1, 24000, device=dev); t = torch.zeros(1, 1, 24000, device=dev)
for _ in range(3): net.encode_latents(a, t)
torch.cuda.synchronize()
for _ in range(20): net.encode_latents(a, t)
torch.cuda.synchronize()
