#!/usr/bin/env python3
"""Host-only experiment: a wide DecoderBlock's three ResidualUnits issued per batch SLICE, so that a slice's
intermediates (k7 output, raw / Snake outputs of the 1x1) are still in the 256 MB Infinity Cache when the next launch reads
them, against the same launches over the whole batch.  usage: slice_probe.py [C] [T] [B]   (default 192 11996 256)"""
import math, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from multimodal_vqvae_compression_audio_tactile_amd import ops

C = int(sys.argv[1]) if len(sys.argv) > 1 else 192
T = int(sys.argv[2]) if len(sys.argv) > 2 else 11996
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
units = []
for d in (1, 3, 9):
    w7 = ops.pack_conv1d((torch.randn(C, C, 7, generator=g) / math.sqrt(7 * C)).to(dev))
    w1 = ops.pack_conv1d((torch.randn(C, C, 1, generator=g) / math.sqrt(C)).to(dev))
    units.append((d, w7, torch.randn(C, generator=g).to(dev), w1, torch.randn(C, generator=g).to(dev),
                  (torch.rand(C, generator=g) + 0.5).to(dev), (torch.rand(C, generator=g) + 0.5).to(dev)))
h = torch.randn(B, C, T, device=dev)
hs = torch.randn(B, C, T, device=dev)
a_next = (torch.rand(C, device=dev) + 0.5)


def chain(h, hs, out=None):
    for i, (d, w7, b7, w1, b1, a_mid, a_dual) in enumerate(units):
        t = ops.conv1d(hs, w7, C, 7, bias=b7, dil=d, pad=3 * d, alpha_out=a_mid)
        if i < 2:
            h, hs = ops.conv1d(t, w1, C, 1, bias=b1, residual=h, alpha_dual=a_dual)
        else:
            h = ops.conv1d(t, w1, C, 1, bias=b1, residual=h, alpha_out=a_next, out=out)
    return h


def timed(f, n=3):
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); y = f(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best, y


ms_full, y_full = timed(lambda: chain(h, hs))
flops = 3 * 2.0 * C * C * 8 * T * B
print(f"C={C} T={T} B={B}: whole batch {ms_full:8.3f} ms  {flops / ms_full * 1e-9:6.1f} TF")
for S in (2, 4, 8, 16, 32):
    if S > B:
        break
    y = torch.empty_like(h)
    def sliced():
        for s in range(0, B, S):
            chain(h[s:s + S], hs[s:s + S], out=y[s:s + S])
        return y
    ms, ys = timed(sliced)
    print(f"   slices of {S:3d}: {ms:8.3f} ms  {flops / ms * 1e-9:6.1f} TF   bit-equal {torch.equal(ys, y_full)}   working set {5 * S * C * T * 4 / 1e6:.0f} MB")
