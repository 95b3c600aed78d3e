#!/usr/bin/env python3
"""Where the training step's device memory goes (one MI355X).  usage: train_mem_probe.py [B]
Prints allocated / peak-allocated GB around every phase of one training step (encoders, AR head, saving decoder forward,
losses, backward of the losses, decoder backward, head backward)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import multimodal_vqvae_compression_audio_tactile_amd as mvq
from multimodal_vqvae_compression_audio_tactile_amd import synth, dac

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
sd = synth.proposed_model_state(7, rvq_books=8, rvq_embed=512)
net = mvq.build_proposed(sd, rvq_books=8, rvq_embed=512, device=dev, cls=mvq.AllPredAR)
net.train()
crit = mvq.TrainingLoss()
a, t = synth.audio_segments(B, seed=7).to(dev), synth.tactile_segments(B, seed=7).to(dev)
GB = 1e9


def mark(tag):
    torch.cuda.synchronize()
    print(f"{tag:44s} allocated {torch.cuda.memory_allocated() / GB:7.2f} GB   peak since last mark {torch.cuda.max_memory_allocated() / GB:7.2f} GB", flush=True)
    torch.cuda.reset_peak_memory_stats()


def wrap(obj, name, tag):
    f = getattr(obj, name)
    def g(*a_, **k_):
        mark(f"  before {tag}")
        r = f(*a_, **k_)
        mark(f"  after  {tag}")
        return r
    setattr(obj, name, g)


wrap(net, "_encode_branches", "encoders + A_QUANT")
wrap(net, "_ar_latents", "AR head")
wrap(net.T_DEC, "forward_saving", "decoder forward (saving)")
wrap(net.T_DEC, "backward_input", "decoder backward")
mark("model + inputs resident")
for step in range(2):
    print(f"--- step {step}")
    with torch.enable_grad():
        out = net.forward_step(a, t)
        mark("forward_step done")
        total = crit(out["y_hat"], out["tgt"])
        mark("losses done")
        total.backward()
        mark("backward done")
    del out, total
    mark("step tensors released")
