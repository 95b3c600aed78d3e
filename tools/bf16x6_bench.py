#!/usr/bin/env python3
"""Per-layer timing of the opt-in bf16x6 7-tap conv against the exact fp32 kernel (HIP events, one GPU).
usage: bf16x6_bench.py [B]   -- prints TFLOP/s (fp32-equivalent: 2*Cin*7*Cout*T*B / time) per wide layer."""
import math, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from multimodal_vqvae_compression_audio_tactile_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
def timeit(f, n=10):
    f(); f(); torch.cuda.synchronize()
    best = float("inf")
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best
for name, C, T in (("enc.b2", 256, 3000), ("enc.b3", 512, 600), ("dec.b0", 768, 600), ("dec.b1", 384, 3000), ("dec.b2", 192, 11996)):
    for dil in (1, 3, 9):
        x = torch.randn(B, C, T, device=dev)
        w = torch.randn(C, C, 7, device=dev) / math.sqrt(7 * C)
        bias = torch.randn(C, device=dev); alpha = torch.rand(C, device=dev) + 0.5
        wp = ops.pack_conv1d(w); wq = ops.pack_conv1d_k7_bf16x3(w)
        xs = ops.bf16x3_split(x)
        flops = 2.0 * C * C * 7 * T * B
        t_exact = timeit(lambda: ops.conv1d(x, wp, C, 7, bias=bias, dil=dil, pad=3 * dil, alpha_out=alpha))
        t_split = timeit(lambda: ops.bf16x3_split(x))
        t_n = timeit(lambda: ops.conv1d_k7_bf16x6(xs, wq, B, C, T, C, dil, bias=bias, alpha_out=alpha))
        xh, xam = ops.f16x2_split(x); wh, wam = ops.pack_conv1d_k7_f16x2(w)
        t_hs = timeit(lambda: ops.f16x2_split(x))
        t_h = timeit(lambda: ops.conv1d_k7_f16x3(xh, xam, wh, wam, B, C, T, C, dil, bias=bias, alpha_out=alpha))
        print(f"{name}.k7d{dil}  C {C:4d} T {T:5d}  exact {t_exact:7.3f} ms {flops/t_exact*1e-9:6.1f} TF | split {t_split:6.3f} ms | "
              f"bf16x6 {t_n:7.3f} ms {flops/t_n*1e-9:6.1f} TF | with the split {flops/(t_n+t_split)*1e-9:6.1f} TF || f16x3 {t_h:7.3f} ms {flops/t_h*1e-9:6.1f} TF, split {t_hs:6.3f} ms, together {flops/(t_h+t_hs)*1e-9:6.1f} TF", flush=True)
        del x, xs, xh
