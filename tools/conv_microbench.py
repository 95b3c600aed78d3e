#!/usr/bin/env python3
"""Per-layer timing of every conv shape on the path (HIP events, one GPU).  usage: conv_microbench.py [B] [filter]

A/B against another build of the SAME sources: MVQ_LIB_PATH=/path/to/libmvq_other.so.  Timing builds (pieces of a kernel
compiled out -- wrong results by construction, only the clock is read) are made with
    make -C multimodal_vqvae_compression_audio_tactile_amd/csrc OBJDIR=build_exp OUT=../libmvq_exp.so EXTRA="-DMVQ_TIMING_BUILD -DMVQ_EXP=4"
and load only with MVQ_ALLOW_TIMING_BUILD=1 in the environment (mvq_build_flags() != 0; include/mvq.h); the same variable is
needed when an A/B knob (MVQ_NO_DMA, MVQ_ROWFAST_MAX_KB, MVQ_NO_TOKEN_RVQ) is set."""
import math, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from multimodal_vqvae_compression_audio_tactile_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
flt = sys.argv[2] if len(sys.argv) > 2 else ""
dev = torch.device("cuda:0")
layers = []   # (name, kind, Cin, Cout, ks, stride, dil, Tin, alpha_in, residual, alpha_out, count)
import os
UNFUSE = [int(c) for c in os.environ.get("MVQ_MB_UNFUSE", "").split(",") if c]     # time these widths as two launches
def ru(prefix, C, T, n):
    if C <= 128 and C not in UNFUSE:
        for d in (1, 3, 9):
            layers.append((f"{prefix}.RUd{d}", "r", C, C, 7, 1, d, T, True, True, False, n))
        return
    for d in (1, 3, 9):
        layers.append((f"{prefix}.k7d{d}", "c", C, C, 7, 1, d, T, False, False, True, n))     # wide units get a pre-snaked input
    # the production calls of the 1x1: skip input + dual output (raw + Snake for the next unit) twice per block, skip + the
    # block's trailing Snake once
    layers.append((f"{prefix}.k1dual", "d", C, C, 1, 1, 1, T, False, True, False, 2 * n))
    layers.append((f"{prefix}.k1", "c", C, C, 1, 1, 1, T, False, True, True, n))
T = 24000; C = 64
for i, s in enumerate((2, 4, 5, 8)):
    ru(f"enc.b{i}", C, T, 2)
    layers.append((f"enc.b{i}.down", "c", C, 2 * C, 2 * s, s, 1, T, False, False, False, 2))
    T //= s; C *= 2
layers.append(("enc.out.k3", "c", 1024, 1024, 3, 1, 1, 75, False, False, False, 2))
layers.append(("dec.in.k7", "c", 1024, 1536, 7, 1, 1, 75, False, False, True, 1))
T = 75; C = 1536
for i, s in enumerate((8, 5, 4, 2)):
    layers.append((f"dec.b{i}.up", "t", C, C // 2, 2 * s, s, 1, T, False, False, False, 1))
    T = (T - 1) * s - 2 * math.ceil(s / 2) + 2 * s; C //= 2
    ru(f"dec.b{i}", C, T, 1)
layers.append(("pred.linear1024", "f", 1024, 1024, 1, 1, 1, 16, False, False, False, 20))
layers.append(("pred.ffn1", "f", 1024, 2048, 1, 1, 1, 16, False, False, False, 5))
layers.append(("pred.ffn3", "f", 2048, 1024, 1, 1, 1, 16, False, True, False, 5))
layers.append(("proj_down", "f", 1024, 96, 1, 1, 1, 16, False, False, False, 5))
layers.append(("proj_up", "f", 96, 1024, 1, 1, 1, 16, False, True, False, 5))

tot = 0.0
print(f"B={B}")
for (name, kind, cin, cout, ks, st, dil, tin, ai, res, ao, count) in layers:
    if flt and flt not in name: continue
    if kind == "f":
        x = torch.randn(1, cin, B * tin, device=dev)
    else:
        x = torch.randn(B, cin, tin, device=dev)
    Bx, _, Tx = x.shape
    tv = 0
    if kind in ("c", "d") and st == 1 and Tx % 4 != 0:            # zero-padded rows (include/mvq.h): what the decoder really runs
        tv = Tx
        x = torch.nn.functional.pad(x, (0, 4 - Tx % 4)); Tx = x.shape[-1]
    if kind == "r":
        w7 = ops.pack_conv1d(torch.randn(cout, cin, 7, device=dev) / math.sqrt(cin * 7))
        w1 = ops.pack_conv1d(torch.randn(cout, cin, 1, device=dev) / math.sqrt(cin))
        b7 = torch.randn(cout, device=dev); b1 = torch.randn(cout, device=dev)
        aa = torch.rand(cin, device=dev) + 0.5; ab = torch.rand(cin, device=dev) + 0.5
        if os.environ.get("MVQ_MB_RU", "presnaked") == "presnaked":     # the production call: pre-snaked input + dual output
            xs = torch.randn_like(x); a2 = torch.rand(cin, device=dev) + 0.5
            f = lambda: ops.residual_unit(x, w7, b7, aa, ab, w1, b1, dil, x_snaked=xs, alpha_dual=a2)
        else:                                                             # round-2 call: Snake on load, single output
            f = lambda: ops.residual_unit(x, w7, b7, aa, ab, w1, b1, dil)
        flops = 2.0 * cin * cout * 8 * Tx * Bx
        kname = "       residual_unit_kernel"
    elif kind == "t":
        w = torch.randn(cin, cout, ks, device=dev) / math.sqrt(cin * 2)
        wp = ops.pack_conv_transpose1d(w, st)
        f = lambda: ops.conv_transpose1d(x, wp, cout, st, math.ceil(st / 2))
        flops = 2.0 * cin * cout * ks * Tx * Bx
        kname = ops.conv_kernel_name(cin, cout, ks, st, 1, True, tin=Tx, batch=Bx)
    else:
        w = torch.randn(cout, cin, ks, device=dev) / math.sqrt(cin * ks)
        wp = ops.pack_conv1d(w)
        pad = (ks - 1) * dil // 2 if st == 1 else math.ceil(st / 2)
        tout = ops.conv1d_out_len(Tx, ks, st, dil, pad)
        a_in = torch.rand(cin, device=dev) + 0.5 if ai else None
        a_out = torch.rand(cout, device=dev) + 0.5 if ao else None
        r = torch.randn(Bx, cout, tout, device=dev) if res else None
        bias = torch.randn(cout, device=dev)
        a_dual = torch.rand(cout, device=dev) + 0.5 if kind == "d" else None
        f = lambda: ops.conv1d(x, wp, cout, ks, bias=bias, stride=st, dil=dil, pad=pad, alpha_in=a_in, residual=r, alpha_out=a_out,
                               alpha_dual=a_dual, tvalid=tv)
        flops = 2.0 * cin * cout * ks * tout * Bx
        kname = ops.conv_kernel_name(cin, cout, ks, st, dil, tin=Tx, batch=Bx)
    f(); f(); torch.cuda.synchronize()
    # best of three groups of n back-to-back launches (single groups of 5 scattered by up to 10 % on a shared box)
    n = max(5, min(20, int(30.0 / max(1e-3, flops * 1e-12 / 100.0 * 1e3))))      # ~30 ms of work per group
    ms = float("inf")
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        ms = min(ms, e0.elapsed_time(e1) / n)
    if os.environ.get("MVQ_MB_EVENTS") == "1":        # device time of the kernel(s) alone (per-launch HIP events inside the library):
        ops.profile_begin()                            # what a latency-regime launch costs without the Python / ctypes call around it
        for _ in range(n): f()
        d = ops.profile_end()
        ms = 1e3 * sum(v["seconds"] for v in d.values()) / n
        kname = " " * 18 + "+".join(k.replace("conv1d_mfma_kernel", "mfma").replace("conv1d_lat_kernel", "LAT").replace("residual_unit_kernel", "ru") for k in d)
    tot += ms * count
    print(f"{name:18s} {kname[18:60]:42s} Cin {cin:5d} Cout {cout:5d} T {Tx:7d} {ms:8.3f} ms x{count:2d} = {ms*count:7.2f} ms  {flops/ms*1e-9:6.1f} TF")
print(f"sum over path: {tot:.1f} ms per step")
