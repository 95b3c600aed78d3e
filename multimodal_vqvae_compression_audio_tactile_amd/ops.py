"""Tensor-level wrappers over the C ABI (include/mvq.h).

torch is plumbing here: device memory (caching allocator), the current HIP stream and nothing else.
Every op takes fp32 tensors on a HIP device, launches on ``torch.cuda.current_stream()`` and returns
freshly allocated outputs.  Inputs arriving in another dtype (e.g. fp16 under the reference's CUDA AMP
autocast, Evaluation/compare_dacvsproposal_5_eval.py:441) are widened to fp32: the path computes in fp32.
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import _lib
from ._lib import MvqError, check


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(t: torch.Tensor, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor")
    if t.device.type != "cuda":
        raise MvqError(f"{name}: tensor is on {t.device}; the MI355X path has no CPU fallback")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _p(t):
    return None if t is None else t.data_ptr()


def conv1d_out_len(tin, ks, stride=1, dil=1, pad=0):
    span = tin + 2 * pad - dil * (ks - 1) - 1
    return 0 if span < 0 else span // stride + 1


def conv_kernel_name(cin, cout, ks, stride=1, dil=1, transposed=False, tin=24000, batch=64) -> str:
    """Kernel instantiation the library launches for this conv shape (as rocprofv3 names it)."""
    import ctypes
    buf = ctypes.create_string_buffer(160)
    check(_lib.lib().mvq_conv_kernel_name(batch, cin, cout, ks, stride, dil, int(transposed), tin, buf, 160),
          "mvq_conv_kernel_name")
    return buf.value.decode()


def weight_norm(v: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """w = v * (g / ||v||) over dim 0 rows (old-style torch weight_norm fold)."""
    v = _dev(v, "v"); g = _dev(g, "g")
    w = torch.empty_like(v)
    rows = v.shape[0]
    check(_lib.lib().mvq_weight_norm_f32(v.data_ptr(), g.data_ptr(), w.data_ptr(), rows, v.numel() // rows, _stream()),
          "mvq_weight_norm_f32")
    return w


def pack_conv1d(w: torch.Tensor) -> torch.Tensor:
    """torch layout w[Cout, Cin, ks] -> K-major packed image used by conv1d()."""
    w = _dev(w, "w")
    cout, cin, ks = w.shape
    n = _lib.lib().mvq_conv1d_packed_floats(cin, cout, ks)
    wp = torch.empty(n, device=w.device, dtype=torch.float32)
    check(_lib.lib().mvq_conv1d_pack_f32(w.data_ptr(), wp.data_ptr(), cin, cout, ks, _stream()), "mvq_conv1d_pack_f32")
    return wp


def pack_conv_transpose1d(w: torch.Tensor, stride: int) -> torch.Tensor:
    """torch layout w[Cin, Cout, 2*stride] -> polyphase packed image used by conv_transpose1d()."""
    w = _dev(w, "w")
    cin, cout, ks = w.shape
    if ks != 2 * stride:
        raise MvqError(f"conv_transpose1d: kernel {ks} != 2*stride ({stride}) is outside the path")
    n = _lib.lib().mvq_conv_transpose1d_packed_floats(cin, cout, stride)
    wp = torch.empty(n, device=w.device, dtype=torch.float32)
    check(_lib.lib().mvq_conv_transpose1d_pack_f32(w.data_ptr(), wp.data_ptr(), cin, cout, stride, _stream()),
          "mvq_conv_transpose1d_pack_f32")
    return wp


def conv1d(x, wp, cout, ks, bias=None, stride=1, dil=1, pad=0, alpha_in=None, residual=None, alpha_out=None,
           tanh=False, out=None, alpha_dual=None, tvalid=0, gelu=False):
    """alpha_dual: also return snake(y_raw, alpha_dual) (the next ResidualUnit's Snake, hoisted): -> (y, y2).
    tvalid: rows are zero-padded beyond column tvalid (see include/mvq.h "Zero-padded rows"); 0 = plain tensors."""
    x = _dev(x, "x")
    B, cin, tin = x.shape
    tout = conv1d_out_len(tin, ks, stride, dil, pad)
    if out is None:
        out = torch.empty(B, cout, tout, device=x.device, dtype=torch.float32)
    y2 = torch.empty_like(out) if alpha_dual is not None else None
    if residual is not None:
        residual = _dev(residual, "residual")
        if tuple(residual.shape) != (B, cout, tout):
            raise MvqError(f"conv1d: residual shape {tuple(residual.shape)} != {(B, cout, tout)}")
    check(_lib.lib().mvq_conv1d_padded_f32(x.data_ptr(), wp.data_ptr(), _p(bias), _p(alpha_in), _p(residual),
                                           _p(alpha_out), out.data_ptr(), _p(y2), _p(alpha_dual), B, cin, tin, cout, ks,
                                           stride, dil, pad, 1 if tanh else (2 if gelu else 0), int(tvalid), _stream()), "mvq_conv1d_f32")
    return out if alpha_dual is None else (out, y2)


def build_flags() -> int:
    """mvq_build_flags() (include/mvq.h): 0 = product build with no A/B override in the environment."""
    return _lib.build_flags()


def profile_begin() -> None:
    """Start bracketing every conv / residual-unit kernel launch with HIP events on its launch stream (include/mvq.h)."""
    check(_lib.lib().mvq_profile_begin(), "mvq_profile_begin")


def profile_reserve(n_launches: int) -> None:
    """Pre-create the event pairs of `n_launches` kernel launches (keeps hipEventCreate out of a timed region)."""
    check(_lib.lib().mvq_profile_reserve(int(n_launches)), "mvq_profile_reserve")


def profile_end() -> dict:
    """Stop and collect: {kernel instantiation name: {"seconds", "flops", "launches"}} (synchronises)."""
    import ctypes
    cap = 1024
    buf = (_lib.ProfileEntry * cap)()
    n, total = ctypes.c_int(0), ctypes.c_int(0)
    check(_lib.lib().mvq_profile_end2(buf, cap, ctypes.byref(n), ctypes.byref(total)), "mvq_profile_end2")
    if total.value > n.value:        # never hand a truncated table to the roofline arithmetic
        raise MvqError(f"mvq_profile_end2: {total.value} kernel instantiations recorded, buffer holds {cap}")
    return {buf[i].kernel.decode(): {"seconds": buf[i].seconds, "flops": buf[i].flops, "launches": buf[i].launches}
            for i in range(n.value)}


def residual_unit_kernel_name(c, dil) -> str:
    import ctypes
    buf = ctypes.create_string_buffer(160)
    check(_lib.lib().mvq_residual_unit_kernel_name(c, dil, buf, 160), "mvq_residual_unit_kernel_name")
    return buf.value.decode()


def residual_unit_fused(x, w7p, b7, alpha_a, alpha_b, w1p, b1, dil, alpha_next=None, alpha_dual=None, tvalid=0, x_snaked=None):
    """The single-launch form (C in {64, 96, 128}).  x_snaked = snake_a(x) (the producer's dual output): staged as is."""
    B, C, T = x.shape
    y = torch.empty_like(x)
    y2 = torch.empty_like(x) if alpha_dual is not None else None
    if x_snaked is not None and x_snaked.shape != x.shape:
        raise MvqError("residual_unit: x_snaked must have the shape of x")
    check(_lib.lib().mvq_residual_unit_padded_f32(x.data_ptr(), _p(x_snaked), w7p.data_ptr(), _p(b7), alpha_a.data_ptr(),
                                                  alpha_b.data_ptr(), w1p.data_ptr(), _p(b1), _p(alpha_next), y.data_ptr(),
                                                  _p(y2), _p(alpha_dual), None, B, C, T, dil, int(tvalid), _stream()),
          "mvq_residual_unit_f32")
    return y if alpha_dual is None else (y, y2)


def residual_unit(x, w7p, b7, alpha_a, alpha_b, w1p, b1, dil, alpha_next=None, x_snaked=None, alpha_dual=None, tvalid=0, w7q=None):
    """x + conv1(snake(conv7_dil(snake(x)))) (+ the next Snake1d): one fused launch for C in {64,96,128}, else the
    two conv launches (Snake on load / on store in the first, skip + next Snake in the second's epilogue).
    x_snaked: snake_a(x) already produced by the previous layer's dual output (skips the staging-time Snake);
    alpha_dual: also return snake(y_raw, alpha_dual) for the next unit -> (y, y2).
    w7q: the layer's packed_mode() image of the 7-tap weights -- given only in an opt-in arithmetic mode (set_arith) for a unit the
    mode claims; None (the default) keeps every launch the exact kernel."""
    x = _dev(x, "x")
    B, C, T = x.shape
    # the fused single-launch form, unless an opt-in arithmetic mode claims the unit's 7-tap conv (then: split + matrix-core conv +
    # the exact 1x1 with its skip, as for the wide units)
    unfuse = w7q is not None and x_snaked is not None and not _KEEP_FUSED
    if _lib.lib().mvq_residual_unit_scratch_floats(B, C, T, dil) == 0 and not unfuse:
        return residual_unit_fused(x, w7p, b7, alpha_a, alpha_b, w1p, b1, dil, alpha_next, alpha_dual, tvalid, x_snaked)
    if x_snaked is not None and w7q is not None:      # opt-in modes (set_arith): non-parity, fp32-class
        h = conv1d_k7_mode(x_snaked, w7q, C, dil, bias=b7, alpha_out=alpha_b, tvalid=tvalid)
    elif x_snaked is not None:
        h = conv1d(x_snaked, w7p, C, 7, bias=b7, dil=dil, pad=3 * dil, alpha_out=alpha_b, tvalid=tvalid)
    else:
        h = conv1d(x, w7p, C, 7, bias=b7, dil=dil, pad=3 * dil, alpha_in=alpha_a, alpha_out=alpha_b, tvalid=tvalid)
    return conv1d(h, w1p, C, 1, bias=b1, residual=x, alpha_out=alpha_next, alpha_dual=alpha_dual, tvalid=tvalid)


# ---- opt-in, NON-PARITY arithmetic mode "bf16x6" (include/mvq.h; csrc/conv_k7_bf16.hip) ------------------------------------------
# The default ("f32") computes every conv as the exact fp32 fma chain of the arithmetic contract.  set_arith("bf16x6") routes the
# 7-tap convs of the wide ResidualUnits (C a multiple of 128, or of 96: C = 192) through the three-piece bf16 split: fp32-accurate,
# not bit-identical; everything else keeps the exact path.
_ARITH = "f32"
# A/B knobs of the opt-in modes, read ONCE at import and reported (dac.plan_overrides / bench.py `plan_overrides`)
ARITH_ENV_SEEN = {k: os.environ[k] for k in ("MVQ_ARITH_KEEP_FUSED", "MVQ_BF16X6_NO96") if k in os.environ}
_KEEP_FUSED = ARITH_ENV_SEEN.get("MVQ_ARITH_KEEP_FUSED") == "1"
_NO96 = ARITH_ENV_SEEN.get("MVQ_BF16X6_NO96") == "1"


def set_arith(mode: str) -> None:
    global _ARITH
    if mode not in ("f32", "bf16x6", "f16x3"):
        raise MvqError(f"set_arith: unknown mode {mode!r} (f32 | bf16x6 | f16x3)")
    _ARITH = mode


def get_arith() -> str:
    return _ARITH


class arith:
    """``with ops.arith("f16x3"): ...`` -- an opt-in arithmetic mode for the duration of a block (restored on exit, also on an
    exception).  The mode is process-global state read at LAUNCH time: a hipGraph captured under one mode keeps it when replayed."""

    def __init__(self, mode: str):
        self.mode, self.prev = mode, None

    def __enter__(self):
        self.prev = get_arith()
        set_arith(self.mode)
        return self

    def __exit__(self, *exc):
        set_arith(self.prev)
        return False


def bf16x6_eligible(c: int) -> bool:
    if _ARITH not in ("bf16x6", "f16x3") or c % 16 != 0:
        return False
    if _NO96:                                             # A/B knob: 128-row tiles only
        return c % 128 == 0
    return c % 128 == 0 or c % 96 == 0


def bf16x3_split(x):
    """x[B, C, T] fp32 -> the three-piece bf16 image [B][C/8][3][T][8] (a flat int16 tensor)."""
    x = _dev(x, "x")
    B, C, T = x.shape
    xs = torch.empty(B * C * T * 3, device=x.device, dtype=torch.int16)
    check(_lib.lib().mvq_bf16x3_split_f32(x.data_ptr(), xs.data_ptr(), B, C, T, _stream()), "mvq_bf16x3_split_f32")
    return xs


def pack_conv1d_k7_bf16x3(w, dgrad=False):
    """Folded weights w[Cout, Cin, 7] fp32 -> the packed three-piece bf16 image of mvq_conv1d_k7_bf16x6_f32.
    dgrad: the input-gradient image instead (rows = Cin, k-channels = Cout, taps reversed)."""
    w = _dev(w, "w").contiguous()
    cout, cin, ks = w.shape
    if dgrad:
        cout, cin = cin, cout
    n = _lib.lib().mvq_conv1d_k7_bf16x3_packed_bytes(cout, cin)
    if ks != 7 or n == 0:
        raise MvqError(f"pack_conv1d_k7_bf16x3: needs [Cout % 128 == 0 or % 96 == 0, Cin % 16 == 0, 7], got {tuple(w.shape)}")
    wq = torch.empty(n // 2, device=w.device, dtype=torch.int16)
    check(_lib.lib().mvq_conv1d_k7_pack_bf16x3(w.data_ptr(), wq.data_ptr(), cout, cin, 1 if dgrad else 0, _stream()), "mvq_conv1d_k7_pack_bf16x3")
    return wq


def conv1d_k7_bf16x6(xs, wq, batch, cin, t, cout, dil, bias=None, alpha_out=None, tvalid=0, dual=False, dsn_src=None, dsn_alpha=None,
                     residual=None):
    """y[B, cout, t] = snake_out(conv7_dil(xs) + bias) on the bf16x6 matrix path; xs from bf16x3_split, wq from pack_conv1d_k7_bf16x3.
    dual: -> (conv + bias, snake_out of it).  dsn_src / dsn_alpha / residual: the input-gradient epilogue (wq packed with dgrad=True)."""
    y = torch.empty(batch, cout, t, device=xs.device, dtype=torch.float32)
    y2 = torch.empty_like(y) if dual else None
    check(_lib.lib().mvq_conv1d_k7_bf16x6_f32(xs.data_ptr(), wq.data_ptr(), _p(bias), _p(alpha_out), y.data_ptr(), _p(y2), _p(dsn_src),
                                              _p(dsn_alpha), _p(residual), batch, cin, t, cout, dil, int(tvalid), _stream()),
          "mvq_conv1d_k7_bf16x6_f32")
    return (y, y2) if dual else y


def f16x2_split(x):
    """x[B, C, T] fp32 -> (two-piece fp16 image [B][C/8][2][T][8] as a flat int16 tensor, per-item |x| maxima as int32 bit patterns)."""
    x = _dev(x, "x")
    B, C, T = x.shape
    xs = torch.empty(B * C * T * 2, device=x.device, dtype=torch.int16)
    xamax = torch.empty(max(B, 1), device=x.device, dtype=torch.int32)
    check(_lib.lib().mvq_f16x2_split_f32(x.data_ptr(), xs.data_ptr(), xamax.data_ptr(), B, C, T, _stream()), "mvq_f16x2_split_f32")
    return xs, xamax


def pack_conv1d_k7_f16x2(w, dgrad=False):
    """Folded weights w[Cout, Cin, 7] fp32 -> (packed two-piece fp16 image, the tensor's |w| maximum as an int32 bit pattern).
    dgrad: the input-gradient image instead (rows = Cin, k-channels = Cout, taps reversed)."""
    w = _dev(w, "w").contiguous()
    cout, cin, ks = w.shape
    if dgrad:
        cout, cin = cin, cout
    n = _lib.lib().mvq_conv1d_k7_f16x2_packed_bytes(cout, cin)
    if ks != 7 or n == 0:
        raise MvqError(f"pack_conv1d_k7_f16x2: needs [Cout % 128 == 0 or % 96 == 0, Cin % 16 == 0, 7], got {tuple(w.shape)}")
    wq = torch.empty(n // 2, device=w.device, dtype=torch.int16)
    wamax = torch.empty(1, device=w.device, dtype=torch.int32)
    check(_lib.lib().mvq_conv1d_k7_pack_f16x2(w.data_ptr(), wq.data_ptr(), wamax.data_ptr(), cout, cin, 1 if dgrad else 0, _stream()),
          "mvq_conv1d_k7_pack_f16x2")
    return wq, wamax


def conv1d_k7_f16x3(xs, xamax, wq, wamax, batch, cin, t, cout, dil, bias=None, alpha_out=None, tvalid=0, dual=False, dsn_src=None,
                    dsn_alpha=None, residual=None):
    """y[B, cout, t] = snake_out(conv7_dil(xs) + bias) with two fp16 pieces per operand (three piece products); dual / dsn_* /
    residual as in conv1d_k7_bf16x6."""
    y = torch.empty(batch, cout, t, device=xs.device, dtype=torch.float32)
    y2 = torch.empty_like(y) if dual else None
    check(_lib.lib().mvq_conv1d_k7_f16x3_f32(xs.data_ptr(), xamax.data_ptr(), wq.data_ptr(), wamax.data_ptr(), _p(bias), _p(alpha_out),
                                             y.data_ptr(), _p(y2), _p(dsn_src), _p(dsn_alpha), _p(residual), batch, cin, t, cout, dil,
                                             int(tvalid), _stream()), "mvq_conv1d_k7_f16x3_f32")
    return (y, y2) if dual else y


def conv1d_k7_mode(x, wq, cout, dil, bias=None, alpha_out=None, tvalid=0, dual=False, dsn_src=None, dsn_alpha=None, residual=None):
    """The current opt-in mode's 7-tap conv on an fp32 input x[B, C, T]: split pass + matrix-core conv.  wq: what the layer's
    packed_mode() / packed_mode_dgrad() returned for this mode."""
    x = _dev(x, "x")
    B, C, T = x.shape
    kw = dict(bias=bias, alpha_out=alpha_out, tvalid=tvalid, dual=dual, dsn_src=dsn_src, dsn_alpha=dsn_alpha, residual=residual)
    if _ARITH == "f16x3":
        xs, xamax = f16x2_split(x)
        return conv1d_k7_f16x3(xs, xamax, wq[0], wq[1], B, C, T, cout, dil, **kw)
    if _ARITH == "bf16x6":
        return conv1d_k7_bf16x6(bf16x3_split(x), wq, B, C, T, cout, dil, **kw)
    raise MvqError("conv1d_k7_mode: no opt-in arithmetic mode is set")


def conv_transpose1d(x, wp, cout, stride, pad, bias=None, alpha_in=None, alpha_out=None, alpha_dual=None, tout_rows=0,
                     tvalid=0, output_padding=0):
    """tout_rows / tvalid: zero-padded rows (include/mvq.h): row length of the output and its true length.
    output_padding: torch's ConvTranspose1d argument (that many more samples at the end of each row)."""
    x = _dev(x, "x")
    B, cin, tin = x.shape
    tout = int(tout_rows) if tout_rows else (tin - 1) * stride - 2 * pad + 2 * stride + int(output_padding)
    out = torch.empty(B, cout, max(tout, 0), device=x.device, dtype=torch.float32)
    y2 = torch.empty_like(out) if alpha_dual is not None else None
    check(_lib.lib().mvq_conv_transpose1d_op_f32(x.data_ptr(), wp.data_ptr(), _p(bias), _p(alpha_in), _p(alpha_out),
                                                 out.data_ptr(), _p(y2), _p(alpha_dual), B, cin, tin, cout, stride,
                                                 pad, int(output_padding), int(tout_rows), int(tvalid), _stream()),
          "mvq_conv_transpose1d_f32")
    return out if alpha_dual is None else (out, y2)


def pack_segments(z, seg_per_row: int, seg_period: int):
    """z[B, C, T] -> zeros[ceil(B / seg_per_row), C, seg_per_row * seg_period] with item b in columns
    [(b % seg_per_row) * seg_period, ... + T) of row b // seg_per_row (the PACKED latent-rate layout, include/mvq.h)."""
    B, C, T = z.shape
    G = (B + seg_per_row - 1) // seg_per_row
    zp = torch.zeros(G, C, seg_per_row, seg_period, device=z.device, dtype=torch.float32)
    full = B // seg_per_row
    if full:
        zp[:full, :, :, :T] = z[:full * seg_per_row].reshape(full, seg_per_row, C, T).permute(0, 2, 1, 3)
    if B > full * seg_per_row:
        rest = B - full * seg_per_row
        zp[full, :, :rest, :T] = z[full * seg_per_row:].permute(1, 0, 2)
    return zp.reshape(G, C, seg_per_row * seg_period)


def conv1d_packed_rows(xp, wp, cout, ks, seg_per_row, seg_period, seg_valid, bias=None, dil=1, pad=0, residual=None,
                       alpha_out=None, alpha_dual=None, tanh=False):
    """Stride-1 'same' conv on PACKED rows xp[G, cin, seg_per_row * seg_period]; the gap columns of the output are zeros."""
    xp = _dev(xp, "xp")
    G, cin, L = xp.shape
    if L != seg_per_row * seg_period:
        raise MvqError("conv1d_packed_rows: row length != seg_per_row * seg_period")
    out = torch.empty(G, cout, L, device=xp.device, dtype=torch.float32)
    y2 = torch.empty_like(out) if alpha_dual is not None else None
    check(_lib.lib().mvq_conv1d_packed_rows_f32(xp.data_ptr(), wp.data_ptr(), _p(bias), _p(residual), _p(alpha_out), out.data_ptr(),
                                                _p(y2), _p(alpha_dual), G, cin, cout, ks, dil, pad, 1 if tanh else 0,
                                                seg_per_row, seg_period, seg_valid, _stream()), "mvq_conv1d_packed_rows_f32")
    return out if alpha_dual is None else (out, y2)


def vpacked_geometry(tin_rows, tin_valid, ks, stride, dil, pad, follow_pad=0):
    """(tout, tout_rows, per_in) of mvq_conv1d_vpacked_f32 for a conv over x[..., tin_rows] with tin_valid data columns, or None
    when the shape does not qualify.  tout_rows leaves `follow_pad` zero columns behind the valid outputs (the padding the NEXT
    conv needs when it runs on these rows virtually packed too)."""
    tout = conv1d_out_len(tin_valid, ks, stride, dil, pad)
    if tin_rows % 4 or tout <= 0:
        return None
    overhang = (tout - 1) * stride - pad + dil * (ks - 1) - (tin_valid - 1)
    gap = max(pad, overhang, 0)
    tout_rows = (max(tout + follow_pad, -(-(tin_valid + gap) // stride), -(-tin_rows // stride)) + 3) // 4 * 4
    return tout, tout_rows, stride * tout_rows


def conv1d_vpacked(x, wp, cout, ks, seg_per_row, tin_valid, tout_rows, bias=None, stride=1, dil=1, pad=0, residual=None,
                   alpha_out=None, alpha_dual=None, tanh=False):
    """Conv over VIRTUALLY packed rows (include/mvq.h): x[B, cin, tin_rows] with tin_valid data columns + zero tail ->
    y[B, cout, tout_rows] (valid outputs, then zeros).  No copy is made: the kernel's LDS-DMA sources and its stores are remapped."""
    x = _dev(x, "x")
    B, cin, tin_rows = x.shape
    per_in = stride * tout_rows
    out = torch.empty(B, cout, tout_rows, device=x.device, dtype=torch.float32)
    y2 = torch.empty_like(out) if alpha_dual is not None else None
    if residual is not None and tuple(residual.shape) != tuple(out.shape):
        raise MvqError(f"conv1d_vpacked: residual shape {tuple(residual.shape)} != {tuple(out.shape)}")
    check(_lib.lib().mvq_conv1d_vpacked_f32(x.data_ptr(), wp.data_ptr(), _p(bias), _p(residual), _p(alpha_out), out.data_ptr(),
                                            _p(y2), _p(alpha_dual), B, cin, tin_rows, int(tin_valid), cout, ks, stride, dil, pad,
                                            1 if tanh else 0, int(seg_per_row), per_in, tout_rows, _stream()), "mvq_conv1d_vpacked_f32")
    return out if alpha_dual is None else (out, y2)


def conv_transpose1d_packed_rows(xp, wp, cout, stride, pad, seg_per_row, seg_period, seg_valid, batch_out, bias=None,
                                 alpha_in=None, alpha_out=None, alpha_dual=None):
    """ConvTranspose1d(kernel 2*stride, stride, pad) reading PACKED rows, writing the unpacked y[batch_out, cout, Tseg]."""
    xp = _dev(xp, "xp")
    G, cin, L = xp.shape
    if L != seg_per_row * seg_period:
        raise MvqError("conv_transpose1d_packed_rows: row length != seg_per_row * seg_period")
    tout = (seg_valid - 1) * stride - 2 * pad + 2 * stride
    out = torch.empty(batch_out, cout, max(tout, 0), device=xp.device, dtype=torch.float32)
    y2 = torch.empty_like(out) if alpha_dual is not None else None
    check(_lib.lib().mvq_conv_transpose1d_packed_rows_f32(xp.data_ptr(), wp.data_ptr(), _p(bias), _p(alpha_in), _p(alpha_out),
                                                          out.data_ptr(), _p(y2), _p(alpha_dual), G, cin, cout, stride, pad,
                                                          seg_per_row, seg_period, seg_valid, batch_out, _stream()),
          "mvq_conv_transpose1d_packed_rows_f32")
    return out if alpha_dual is None else (out, y2)


def rvq_ema_forward(z, books, n_books_use=None, return_indices=False):
    """ResidualVQEMA.forward.  z[B,D,T]; books[nb,K,D] (stacked).  -> q[B,D,T] (, idx[nb_use, B*T] int64)."""
    z = _dev(z, "z"); books = _dev(books, "books")
    B, D, T = z.shape
    nb_all, K, D2 = books.shape
    if D2 != D:
        raise MvqError(f"rvq: book dim {D2} != token dim {D}")
    nb = nb_all if n_books_use is None else max(0, min(int(n_books_use), nb_all))
    q = torch.empty_like(z)
    idx = torch.empty(nb, B * T, device=z.device, dtype=torch.int32) if return_indices else None
    check(_lib.lib().mvq_rvq_ema_forward_f32(z.data_ptr(), books.data_ptr(), q.data_ptr(), _p(idx), B, D, T, nb, K,
                                             _stream()), "mvq_rvq_ema_forward_f32")
    if return_indices:
        return q, idx.long()
    return q


def rvq_ema_step_(z_tokens, books, decay=0.99):
    """ResidualVQEMA.ema_step: updates `books` [nb,K,D] in place."""
    z = _dev(z_tokens, "z_tokens")
    if books.device.type != "cuda" or books.dtype != torch.float32 or not books.is_contiguous():
        raise MvqError("rvq_ema_step_: books must be a contiguous fp32 HIP tensor (updated in place)")
    B, D, T = z.shape
    nb, K, _ = books.shape
    nbytes = _lib.lib().mvq_rvq_ema_step_scratch_bytes(B, T, nb, K, D)
    scratch = torch.empty(max(nbytes, 4), device=z.device, dtype=torch.uint8)
    check(_lib.lib().mvq_rvq_ema_step_f32(z.data_ptr(), books.data_ptr(), scratch.data_ptr(), B, D, T, nb, K,
                                          float(decay), _stream()), "mvq_rvq_ema_step_f32")
    return books


def dac_rvq_prepare(codebook):
    """F.normalize(codebook[nq, K, Dc]) and its squared norms, once (model load): -> (cb_normalised, cb_norm2) for dac_rvq."""
    codebook = _dev(codebook, "codebook")
    nq, K, Dc = codebook.shape
    cbn = torch.empty_like(codebook)
    cn2 = torch.empty(nq, K, device=codebook.device, dtype=torch.float32)
    check(_lib.lib().mvq_dac_rvq_prepare_f32(codebook.data_ptr(), cbn.data_ptr(), cn2.data_ptr(), nq, K, Dc, _stream()),
          "mvq_dac_rvq_prepare_f32")
    return cbn, cn2


def dac_rvq(z, in_w, in_b, codebook, out_w, out_b, n_q, nq_item=None, prepared=None):
    """DAC ResidualVectorQuantize -> (z_q, codes int64 [B,nq,T], latents [B,nq*Dc,T]).  ``nq_item`` (int32 [B] on the
    device): train-mode quantiser dropout -- item b's z_q sums only its first nq_item[b] stages.  ``prepared`` =
    dac_rvq_prepare(codebook): the normalised codebook computed once instead of in every block (same results)."""
    z = _dev(z, "z")
    B, C, T = z.shape
    _, K, Dc = codebook.shape
    zq = torch.empty_like(z)
    codes = torch.empty(B, n_q, T, device=z.device, dtype=torch.int32)
    lat = torch.empty(B, n_q * Dc, T, device=z.device, dtype=torch.float32)
    if nq_item is not None and (nq_item.dtype != torch.int32 or nq_item.numel() != B or not nq_item.is_cuda):
        raise MvqError("dac_rvq: nq_item must be an int32 HIP tensor with one entry per batch item")
    cbn, cn2 = prepared if prepared is not None else (None, None)
    check(_lib.lib().mvq_dac_rvq_prepared_f32(z.data_ptr(), in_w.data_ptr(), in_b.data_ptr(), codebook.data_ptr(), _p(cbn), _p(cn2),
                                              out_w.data_ptr(), out_b.data_ptr(), zq.data_ptr(), codes.data_ptr(),
                                              lat.data_ptr(), _p(nq_item), B, C, T, n_q, K, Dc, _stream()), "mvq_dac_rvq_f32")
    return zq, codes.long(), lat


def layernorm_c(x, gamma, beta, pe=None, eps=1e-5, do_tanh=False, post_scale=1.0, folded_batch=None, sub=None):
    """LayerNorm over channels of x[B,C,T] (of x - sub when ``sub`` is given); with ``folded_batch=B`` x is the
    token-folded [1,C,B*T] layout."""
    x = _dev(x, "x")
    if sub is not None:
        sub = _dev(sub, "sub")
        if sub.shape != x.shape:
            raise MvqError("layernorm_c: sub must have the shape of x")
    B, C, T = x.shape
    sb = sc = 0
    if folded_batch is not None:
        if B != 1 or T % folded_batch:
            raise MvqError("layernorm_c: folded tensor must be [1, C, B*T]")
        B, T = folded_batch, T // folded_batch
        sb, sc = T, B * T
    y = torch.empty_like(x)
    if pe is not None and (pe.shape[0] < T or pe.shape[1] != C):
        raise MvqError(f"layernorm_c: pe table {tuple(pe.shape)} too small for T={T}, C={C}")
    check(_lib.lib().mvq_layernorm_c_sub_f32(x.data_ptr(), _p(sub), _p(pe), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(),
                                             B, C, T, sb, sc, float(eps), int(do_tanh), float(post_scale), _stream()),
          "mvq_layernorm_c_f32")
    return y


def attention(q, k, v, heads, folded_batch=None):
    q = _dev(q, "q"); k = _dev(k, "k"); v = _dev(v, "v")
    B, C, Tq = q.shape
    Tk = k.shape[2]
    strides = (0, 0, 0, 0)
    if folded_batch is not None:
        B = folded_batch
        Tq, Tk = Tq // B, Tk // B
        strides = (Tq, B * Tq, Tk, B * Tk)
    ctx = torch.empty_like(q)
    check(_lib.lib().mvq_attention_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), ctx.data_ptr(), B, heads, C // heads,
                                       Tq, Tk, *strides, _stream()), "mvq_attention_f32")
    return ctx


def attention_kv_slice(q, k_all, v_all, heads, folded_batch, s, tk):
    """Attention of token-folded q[1,C,B*Tq] against columns [s, s+tk) of token-folded K/V over the WHOLE sequence
    (k_all, v_all: [1, C, B*Ta], column b*Ta + t): the kernel takes strides, nothing is copied."""
    q = _dev(q, "q"); k_all = _dev(k_all, "k"); v_all = _dev(v_all, "v")
    B = folded_batch
    C, Tq, Ta = q.shape[1], q.shape[2] // B, k_all.shape[2] // B
    if s < 0 or tk < 0 or s + tk > Ta:
        raise MvqError("attention_kv_slice: slice outside the key/value sequence")
    ctx = torch.empty_like(q)
    check(_lib.lib().mvq_attention_f32(q.data_ptr(), k_all.data_ptr() + 4 * s, v_all.data_ptr() + 4 * s, ctx.data_ptr(),
                                       B, heads, C // heads, Tq, tk, Tq, B * Tq, Ta, B * Ta, _stream()), "mvq_attention_f32")
    return ctx


_AR_CHECKED = set()


def ar_latents_fused(zt, z_run, *, k_all, v_all, t_audio, pe, ln_q, wq, wo, ln_f, w1, b1, w3, b3, ln_eps, tok, tok_eps, scale,
                     wd, bd, wu, bu, books, books_use, heads, c_ff, code_dim, r_tokens=None, idx_out=None, tactile_only=False, chunk=16,
                     staged=False):
    """The whole chunked AR loop as ONE persistent kernel (csrc/ar_fused.hip: mvq_ar_latents_f32): zt[B,C,Tlat] -> z_run[B,C,Tlat]
    (written in place), optionally r_tokens[B,96,Tlat] and idx_out[nb,B,Tlat] (int32).  ``k_all`` / ``v_all``: token-folded K / V
    of all chunks ([1,C,B*t_audio], CrossPredictor.keys_values) or None; the w* are K-major packed 1x1 weights (pack_conv1d);
    ln_q / ln_f / tok = (weight, bias).  ``staged``: the same stages as stand-alone launches issued by ONE host call
    (mvq_ar_latents_staged_f32; batch <= 8) instead of the persistent kernel.  Same bits as the Python loop either way
    (tests/test_gpu_ar_fused.py)."""
    zt = _dev(zt, "zt")
    B, C, Tl = zt.shape
    nb_all, K = (books.shape[0], books.shape[1]) if books is not None else (0, 1)
    nb = nb_all if books_use is None else max(0, min(int(books_use), nb_all))
    a = _lib.ArArgs()
    a.batch, a.t_lat, a.t_audio, a.tactile_only, a.books_use, a.rvq_k = B, Tl, int(t_audio), int(bool(tactile_only)), nb, K
    a.c_lat, a.c_ff, a.code_dim, a.heads, a.chunk = C, int(c_ff), int(code_dim), int(heads), int(chunk)
    a.ln_eps, a.tok_eps, a.scale = float(ln_eps), float(tok_eps), float(scale)
    a.zt, a.z_run, a.r_tokens, a.idx_out = zt.data_ptr(), z_run.data_ptr(), _p(r_tokens), _p(idx_out)
    a.k_all, a.v_all, a.pe = _p(k_all), _p(v_all), _p(pe)
    (a.lnq_g, a.lnq_b), (a.lnf_g, a.lnf_b), (a.tok_g, a.tok_b) = map(lambda t: (_p(t[0]), _p(t[1])), (ln_q, ln_f, tok))
    a.wq, a.wo, a.w1, a.b1, a.w3, a.b3 = _p(wq), _p(wo), _p(w1), _p(b1), _p(w3), _p(b3)
    a.wd, a.bd, a.wu, a.bu, a.books = _p(wd), _p(bd), _p(wu), _p(bu), _p(books)
    L = _lib.lib()
    nbytes = L.mvq_ar_workspace_bytes(B, Tl)
    ws = torch.empty(max(nbytes, 4), device=zt.device, dtype=torch.uint8)
    if staged:
        check(L.mvq_ar_latents_staged_f32(ctypes.byref(a), ws.data_ptr(), nbytes, _stream()), "mvq_ar_latents_staged_f32")
        return z_run
    check(L.mvq_ar_latents_f32(ctypes.byref(a), ws.data_ptr(), nbytes, _stream()), "mvq_ar_latents_f32")
    key = (B, Tl, zt.device.index, bool(tactile_only))
    if key not in _AR_CHECKED:            # co-residency of the persistent grid is a property of the shape: verified on its first use
        check(L.mvq_ar_check(ws.data_ptr(), _stream()), "mvq_ar_check")
        _AR_CHECKED.add(key)
    return z_run


def gelu(x):
    x = _dev(x, "x")
    y = torch.empty_like(x)
    check(_lib.lib().mvq_gelu_f32(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "mvq_gelu_f32")
    return y


def fold_time_slice(a, s, e):
    """a[B,C,T][..., s:e] -> token-folded [1, C, B*(e-s)] (column b*(e-s)+i = token i of batch element b)."""
    a = _dev(a, "a")
    B, C, T = a.shape
    n = e - s
    y = torch.empty(1, C, B * n, device=a.device, dtype=torch.float32)
    check(_lib.lib().mvq_copy3d_f32(a.data_ptr() + 4 * s, C * T, T, y.data_ptr(), n, B * n, B, C, n, _stream()),
          "mvq_copy3d_f32")
    return y


def unfold_into_(dst, s, src, batch):
    """dst[B,C,T][..., s:s+n] = unfold(src[1, C, B*n])."""
    B, C, T = dst.shape
    n = src.shape[2] // batch
    check(_lib.lib().mvq_copy3d_f32(src.data_ptr(), n, B * n, dst.data_ptr() + 4 * s, C * T, T, B, C, n, _stream()),
          "mvq_copy3d_f32")
    return dst


def fold_column_into_(dst_folded, col, src, t, batch):
    """dst_folded[1,C,B*n][:, :, b*n + col] = src[B,C,T][b, :, t]."""
    B, C, T = src.shape
    n = dst_folded.shape[2] // batch
    check(_lib.lib().mvq_copy3d_f32(src.data_ptr() + 4 * t, C * T, T, dst_folded.data_ptr() + 4 * col, n, B * n,
                                    B, C, 1, _stream()), "mvq_copy3d_f32")
    return dst_folded


def sub(a, b):
    """a - b for equal-shaped contiguous tensors."""
    a = _dev(a, "a"); b = _dev(b, "b")
    y = torch.empty_like(a)
    n = a.numel()
    check(_lib.lib().mvq_sub3d_f32(a.data_ptr(), 0, 0, b.data_ptr(), 0, 0, y.data_ptr(), 0, 0, 1, 1, n, _stream()),
          "mvq_sub3d_f32")
    return y


def align_xcorr(ref, est, max_shift=200):
    """Correlations c(s), s in [-max_shift, max_shift], of equal-length 1-D signals and the best shift (device int32)."""
    ref = _dev(ref, "ref").reshape(-1); est = _dev(est, "est").reshape(-1)
    if ref.numel() != est.numel():
        raise MvqError("align_xcorr: crop_match the signals first (equal lengths)")
    n = 2 * max_shift + 1
    corr = torch.empty(n, device=ref.device, dtype=torch.float32)
    scratch = torch.empty(n, device=ref.device, dtype=torch.int32)
    best = torch.zeros(1, device=ref.device, dtype=torch.int32)
    check(_lib.lib().mvq_align_xcorr_f32(ref.data_ptr(), est.data_ptr(), ref.numel(), max_shift, corr.data_ptr(),
                                         scratch.data_ptr(), best.data_ptr(), _stream()), "mvq_align_xcorr_f32")
    return corr, best


def align_xcorr_batch(ref, est, max_shift=200):
    """Best shift per item for ref[B,T] / est[B,T] (equal lengths): ONE launch pair for the whole batch (grid dimension =
    item), NO host sync; returns the int32 device tensor of shifts."""
    ref = _dev(ref, "ref"); est = _dev(est, "est")
    if ref.shape != est.shape or ref.dim() != 2:
        raise MvqError("align_xcorr_batch: ref and est must both be [B, T]")
    B, T = ref.shape
    n = 2 * max_shift + 1
    corr = torch.empty(B, n, device=ref.device, dtype=torch.float32)
    scratch = torch.empty(B, n, device=ref.device, dtype=torch.int32)
    best = torch.zeros(B, device=ref.device, dtype=torch.int32)
    check(_lib.lib().mvq_align_xcorr_batch_f32(ref.data_ptr(), est.data_ptr(), B, T, max_shift, corr.data_ptr(),
                                               scratch.data_ptr(), best.data_ptr(), _stream()), "mvq_align_xcorr_batch_f32")
    return best


def resample_ragged(x, kern, off, length, orig, newf, width, lout_pitch):
    """Rows x[b, off[b] : off[b] + length[b]] resampled by the filter bank kern[newf, 2*width + orig] in one launch.
    off / length: int32 device tensors [B].  -> (y[B, lout_pitch] zero past each row's output length, lout int32 [B])."""
    x = _dev(x, "x")
    if x.dim() != 2 or off.dtype != torch.int32 or length.dtype != torch.int32 or not off.is_cuda or not length.is_cuda:
        raise MvqError("resample_ragged: x must be [B, L]; off / length int32 HIP tensors")
    B, L = x.shape
    y = torch.empty(B, lout_pitch, device=x.device, dtype=torch.float32)
    lout = torch.empty(B, device=x.device, dtype=torch.int32)
    check(_lib.lib().mvq_resample_ragged_f32(x.data_ptr(), kern.data_ptr(), y.data_ptr(), off.data_ptr(), length.data_ptr(),
                                             lout.data_ptr(), B, L, lout_pitch, orig, newf, width, kern.shape[1], _stream()),
          "mvq_resample_ragged_f32")
    return y, lout


# ---------------------------------------------------------------------------------- backward (row f1)
def pack_conv1d_dgrad(w: torch.Tensor) -> torch.Tensor:
    """Conv1d weight w[Cout,Cin,ks] -> packed image of its input-gradient conv (flip + transpose)."""
    w = _dev(w, "w")
    cout, cin, ks = w.shape
    wp = torch.empty(_lib.lib().mvq_conv1d_dgrad_packed_floats(cin, cout, ks), device=w.device, dtype=torch.float32)
    check(_lib.lib().mvq_conv1d_pack_dgrad_f32(w.data_ptr(), wp.data_ptr(), cin, cout, ks, _stream()),
          "mvq_conv1d_pack_dgrad_f32")
    return wp


def pack_conv_transpose1d_dgrad(w: torch.Tensor) -> torch.Tensor:
    """ConvTranspose1d weight w[Cin,Cout,ks] -> packed image of its input-gradient (a strided conv)."""
    w = _dev(w, "w")
    cin, cout, ks = w.shape
    wp = torch.empty(_lib.lib().mvq_conv1d_dgrad_packed_floats(cin, cout, ks), device=w.device, dtype=torch.float32)
    check(_lib.lib().mvq_conv_transpose1d_pack_dgrad_f32(w.data_ptr(), wp.data_ptr(), cin, cout, ks, _stream()),
          "mvq_conv_transpose1d_pack_dgrad_f32")
    return wp


def conv1d_dgrad(gy, wp_dgrad, cin, tin, ks, stride=1, dil=1, pad=0, dsnake_src=None, dsnake_alpha=None, residual=None):
    """gx[B,cin,tin] = dgrad-conv(gy) * dsnake(dsnake_src) + residual; layer described by its FORWARD geometry."""
    gy = _dev(gy, "gy")
    B, cout, tout = gy.shape
    gx = torch.empty(B, cin, tin, device=gy.device, dtype=torch.float32)
    check(_lib.lib().mvq_conv1d_dgrad_f32(gy.data_ptr(), wp_dgrad.data_ptr(), _p(dsnake_src), _p(dsnake_alpha),
                                          _p(residual), gx.data_ptr(), B, cin, tin, cout, tout, ks, stride, dil, pad,
                                          _stream()), "mvq_conv1d_dgrad_f32")
    return gx


def mul_dtanh(g, y):
    g = _dev(g, "g"); y = _dev(y, "y")
    out = torch.empty_like(g)
    check(_lib.lib().mvq_mul_dtanh_f32(g.data_ptr(), y.data_ptr(), out.data_ptr(), g.numel(), _stream()), "mvq_mul_dtanh_f32")
    return out


# ------------------------------------------------------- backward of the reference-owned trainable modules (row f1)
def _fold_dims(x, folded_batch):
    B, C, T = x.shape
    if folded_batch is None:
        return B, C, T, 0, 0
    if B != 1 or T % folded_batch:
        raise MvqError("folded tensor must be [1, C, B*T]")
    T //= folded_batch
    return folded_batch, C, T, T, folded_batch * T


def layernorm_c_bwd(x, gamma, g, pe=None, eps=1e-5, folded_batch=None, need_gx=True):
    """-> (gx or None, dgamma, dbeta) for y = LayerNorm_C(x + pe)."""
    x = _dev(x, "x"); g = _dev(g, "g")
    B, C, T, sb, sc = _fold_dims(x, folded_batch)
    gx = torch.empty_like(x) if need_gx else None
    dgamma = torch.zeros(C, device=x.device, dtype=torch.float32)
    dbeta = torch.zeros(C, device=x.device, dtype=torch.float32)
    stats = torch.empty(2 * max(B * T, 1), device=x.device, dtype=torch.float32)
    check(_lib.lib().mvq_layernorm_c_bwd_f32(x.data_ptr(), _p(pe), gamma.data_ptr(), g.data_ptr(), _p(gx),
                                             dgamma.data_ptr(), dbeta.data_ptr(), stats.data_ptr(), B, C, T, sb, sc,
                                             float(eps), _stream()), "mvq_layernorm_c_bwd_f32")
    return gx, dgamma, dbeta


def gelu_bwd(x, g):
    x = _dev(x, "x"); g = _dev(g, "g")
    gx = torch.empty_like(x)
    check(_lib.lib().mvq_gelu_bwd_f32(x.data_ptr(), g.data_ptr(), gx.data_ptr(), x.numel(), _stream()), "mvq_gelu_bwd_f32")
    return gx


def scale_tanh(u, scale: float):
    u = _dev(u, "u")
    y = torch.empty_like(u)
    check(_lib.lib().mvq_scale_tanh_f32(u.data_ptr(), float(scale), y.data_ptr(), u.numel(), _stream()), "mvq_scale_tanh_f32")
    return y


def scale_tanh_bwd(u, g, scale: float):
    """-> (gu, d/dscale as a 0-d device tensor)."""
    u = _dev(u, "u"); g = _dev(g, "g")
    gu = torch.empty_like(u)
    nblk = max(1, min(1024, (u.numel() + 255) // 256))
    partial = torch.zeros(nblk, device=u.device, dtype=torch.float32)
    check(_lib.lib().mvq_scale_tanh_bwd_f32(u.data_ptr(), g.data_ptr(), float(scale), gu.data_ptr(), partial.data_ptr(),
                                            nblk, u.numel(), _stream()), "mvq_scale_tanh_bwd_f32")
    dscale = torch.empty(1, device=u.device, dtype=torch.float32)
    check(_lib.lib().mvq_rowsum_f32(partial.data_ptr(), dscale.data_ptr(), 1, nblk, 0, _stream()), "mvq_rowsum_f32")
    return gu, dscale.reshape(())


def attention_bwd(q, k, v, g, heads, folded_batch=None):
    q = _dev(q, "q"); k = _dev(k, "k"); v = _dev(v, "v"); g = _dev(g, "g")
    B, C, Tq = q.shape
    Tk = k.shape[2]
    strides = (0, 0, 0, 0)
    if folded_batch is not None:
        B = folded_batch
        Tq, Tk = Tq // B, Tk // B
        strides = (Tq, B * Tq, Tk, B * Tk)
    gq, gk, gv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    check(_lib.lib().mvq_attention_bwd_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), g.data_ptr(), gq.data_ptr(),
                                           gk.data_ptr(), gv.data_ptr(), B, heads, C // heads, Tq, Tk, *strides,
                                           _stream()), "mvq_attention_bwd_f32")
    return gq, gk, gv


def mul_scaled(a, b, scale=1.0):
    a = _dev(a, "a"); b = _dev(b, "b")
    out = torch.empty_like(a)
    check(_lib.lib().mvq_mul_scaled_f32(a.data_ptr(), b.data_ptr(), float(scale), out.data_ptr(), a.numel(), _stream()),
          "mvq_mul_scaled_f32")
    return out


def transpose2d(x):
    x = _dev(x, "x")
    r, c = x.shape
    out = torch.empty(c, r, device=x.device, dtype=torch.float32)
    check(_lib.lib().mvq_transpose2d_f32(x.data_ptr(), out.data_ptr(), r, c, _stream()), "mvq_transpose2d_f32")
    return out


def rowsum(x):
    x = _dev(x, "x")
    r, c = x.shape
    out = torch.empty(r, device=x.device, dtype=torch.float32)
    check(_lib.lib().mvq_rowsum_f32(x.data_ptr(), out.data_ptr(), r, c, 0, _stream()), "mvq_rowsum_f32")
    return out


def linear_wgrad(g, x):
    """dW[O,I] = g[O,N] x[I,N]^T over the token axis N, as a k=1 conv on the MFMA kernel with K = tokens:
    packed 'weights' = g^T (K-major), input = x^T [N, I].  N is padded with zero tokens to a multiple of 32."""
    g = _dev(g, "g"); x = _dev(x, "x")
    O, N = g.shape
    I, N2 = x.shape
    if N != N2:
        raise MvqError("linear_wgrad: token counts differ")
    if N % 32:
        padn = 32 - N % 32
        g = torch.nn.functional.pad(g, (0, padn)); x = torch.nn.functional.pad(x, (0, padn)); N += padn
    wp = pack_conv1d(g.reshape(O, N, 1))
    xt = transpose2d(x).reshape(1, N, I)
    return conv1d(xt, wp, O, 1).reshape(O, I)


# ---------------------------------------------------------------------------------------- training losses (row f2)
def stft_frames(x, window, out, col0, n_fft, hop):
    """x[B,T] -> out[n_fft, ncols][:, col0 : col0 + B*nframes] (windowed, reflect-padded, center=True)."""
    B, T = x.shape
    nfr = 1 + T // hop
    check(_lib.lib().mvq_stft_frames_f32(x.data_ptr(), window.data_ptr(), out.data_ptr(), B, T, n_fft, hop, nfr,
                                         out.shape[1], col0, _stream()), "mvq_stft_frames_f32")


def spec_mag(S, F, Fp, eps):
    ncols = S.shape[-1]
    mag = torch.empty(Fp, ncols, device=S.device, dtype=torch.float32)
    check(_lib.lib().mvq_spec_mag_f32(S.data_ptr(), mag.data_ptr(), F, Fp, ncols, float(eps), _stream()), "mvq_spec_mag_f32")
    return mag


def spec_loss_sums(mag, F, B, nfr):
    """-> [3, B]: per item sums of (X-Y)^2, Y^2, |X-Y|."""
    P = 16
    partial = torch.empty(3 * B, P, device=mag.device, dtype=torch.float32)
    check(_lib.lib().mvq_spec_loss_partial_f32(mag.data_ptr(), partial.data_ptr(), P, F, B, nfr, mag.shape[1], _stream()),
          "mvq_spec_loss_partial_f32")
    return rowsum(partial).reshape(3, B)


def spec_grad(S, mag, coef_a, coef_b, extra, F, Fp, B, nfr, eps):
    G = torch.empty(2 * Fp, B * nfr, device=S.device, dtype=torch.float32)
    check(_lib.lib().mvq_spec_grad_f32(S.data_ptr(), mag.data_ptr(), _p(coef_a), float(coef_b), _p(extra), G.data_ptr(),
                                       F, Fp, B, nfr, mag.shape[1], float(eps), _stream()), "mvq_spec_grad_f32")
    return G


def overlap_add_(dy, dframes, window, n_fft, hop):
    B, T = dy.shape
    check(_lib.lib().mvq_overlap_add_f32(dframes.data_ptr(), window.data_ptr(), dy.data_ptr(), B, T, n_fft, hop,
                                         1 + T // hop, _stream()), "mvq_overlap_add_f32")
    return dy


def l1_loss_sum(y, tgt, dy=None, coef=0.0):
    """sum |finite_or_zero(y) - finite_or_zero(tgt)| (0-d device tensor); dy += coef*sign(...) when dy is given."""
    n = y.numel()
    P = max(1, min(1024, (n + 255) // 256))
    partial = torch.empty(1, P, device=y.device, dtype=torch.float32)
    check(_lib.lib().mvq_l1_loss_f32(y.data_ptr(), tgt.data_ptr(), partial.data_ptr(), P, _p(dy), float(coef), n, _stream()),
          "mvq_l1_loss_f32")
    return rowsum(partial).reshape(())


def mel_max(M, n_mels, B, nfr):
    maxv = torch.empty(2 * B, device=M.device, dtype=torch.float32)
    argm = torch.empty(2 * B, device=M.device, dtype=torch.int32)
    check(_lib.lib().mvq_mel_max_f32(M.data_ptr(), maxv.data_ptr(), argm.data_ptr(), n_mels, B, nfr, M.shape[-1], _stream()),
          "mvq_mel_max_f32")
    return maxv, argm


def mel_cos(M, maxv, n_mels, B, nfr, eps, coef=None, use_log=True):
    """-> cos[B*nfr] (, dM[n_mels, B*nfr], dden[B*nfr] when coef = dL/dcos is given)."""
    cosv = torch.empty(B * nfr, device=M.device, dtype=torch.float32)
    dM = torch.empty(n_mels, B * nfr, device=M.device, dtype=torch.float32) if coef is not None else None
    dden = torch.empty(B * nfr, device=M.device, dtype=torch.float32) if coef is not None else None
    check(_lib.lib().mvq_mel_cos_f32(M.data_ptr(), maxv.data_ptr(), cosv.data_ptr(), _p(dM), _p(dden),
                                     float(coef or 0.0), n_mels, B, nfr, M.shape[-1], float(eps), int(use_log), _stream()),
          "mvq_mel_cos_f32")
    return cosv, dM, dden


def mel_max_grad_(dM, dden, maxv, argm, B, nfr, eps):
    check(_lib.lib().mvq_mel_max_grad_f32(dden.data_ptr(), maxv.data_ptr(), argm.data_ptr(), dM.data_ptr(), B, nfr, float(eps),
                                          _stream()), "mvq_mel_max_grad_f32")
    return dM


# ---------------------------------------------------------------------------------------- whole stacks (include/mvq.h, ABI 3)
class Stack:
    """mvq_stack handle of one dac Encoder / Decoder: launch plan + folded / packed weights in one device blob.

    ``Stack.encoder(d_model, strides, d_latent)`` / ``Stack.decoder(input_channel, channels, rates, d_out, output_padding)`` make a
    description-only handle (``param_names()`` lists the upstream state-dict keys it expects, in order); ``bind(tensors)`` makes
    the working handle from the parameter tensors in that order.  ``encoder_fwd`` / ``decoder_fwd`` / ``decoder_fwd_saving`` /
    ``decoder_bwd_input`` are ONE C call each (mvq_encoder_fwd_f32, ...): the per-layer plan lives in the library."""

    _BY_ID = {}                      # torch.ops.mi355x_vqvae.encoder_fwd / decoder_fwd / decoder_bwd_input name a stack by this id

    def __init__(self, kind, desc, handle, blob=None):
        import weakref
        self.kind, self.desc, self.handle, self.blob = kind, desc, handle, blob
        self.id = id(self)
        Stack._BY_ID[self.id] = weakref.ref(self)

    def __deepcopy__(self, memo):            # a handle is not copied with its module: the copy rebuilds its own on first use
        return None

    def __reduce__(self):                    # ... nor pickled (torch.save(module)): it pickles as None
        return (type(None), ())

    @classmethod
    def by_id(cls, i):
        ref = cls._BY_ID.get(int(i))
        st = ref() if ref else None
        if st is None:
            raise MvqError(f"no live Stack with id {i}")
        return st

    @classmethod
    def _make(cls, kind, desc, params=None):
        import ctypes
        L = _lib.lib()
        create = L.mvq_encoder_create if kind == "encoder" else L.mvq_decoder_create
        h = ctypes.c_void_p()
        if params is None:
            check(create(ctypes.byref(h), ctypes.byref(desc), None, None, 0, None), f"mvq_{kind}_create")
            return cls(kind, desc, h)
        probe = cls._make(kind, desc)
        names = probe.param_names()
        if len(params) != len(names):
            raise MvqError(f"Stack.bind: {len(params)} tensors for {len(names)} parameters")
        ts = [_dev(p.detach(), n) for p, n in zip(params, names)]
        for t, (n, shp) in zip(ts, probe.param_info()):
            if t.numel() != shp[0] * shp[1] * shp[2]:
                raise MvqError(f"Stack.bind: {n} has {t.numel()} elements, expected shape {shp}")
        blob = torch.empty(int(L.mvq_stack_weights_bytes(probe.handle)), device=ts[0].device, dtype=torch.uint8)
        arr = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        check(create(ctypes.byref(h), ctypes.byref(desc), arr, blob.data_ptr(), blob.numel(), _stream()), f"mvq_{kind}_create")
        return cls(kind, desc, h, blob)

    @classmethod
    def encoder(cls, d_model, strides, d_latent, params=None):
        d = _lib.EncoderDesc()
        d.d_model, d.n_strides, d.d_latent = int(d_model), len(strides), int(d_latent)
        for i, s in enumerate(strides):
            d.strides[i] = int(s)
        return cls._make("encoder", d, params)

    @classmethod
    def decoder(cls, input_channel, channels, rates, d_out=1, output_padding=False, params=None):
        d = _lib.DecoderDesc()
        d.input_channel, d.channels, d.n_rates, d.d_out, d.output_padding = int(input_channel), int(channels), len(rates), int(d_out), int(bool(output_padding))
        for i, s in enumerate(rates):
            d.rates[i] = int(s)
        return cls._make("decoder", d, params)

    def bind(self, params):
        return Stack._make(self.kind, self.desc, list(params))

    def __del__(self):
        try:
            Stack._BY_ID.pop(getattr(self, "id", None), None)
            if self.handle:
                _lib.lib().mvq_stack_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def param_info(self):
        import ctypes
        L = _lib.lib()
        out = []
        for i in range(L.mvq_stack_param_count(self.handle)):
            buf = ctypes.create_string_buffer(128)
            dims = (ctypes.c_int * 3)()
            check(L.mvq_stack_param_info(self.handle, i, buf, 128, dims), "mvq_stack_param_info")
            out.append((buf.value.decode(), tuple(dims)))
        return out

    def param_names(self):
        return [n for n, _ in self.param_info()]

    def set_plan(self, vpack_min_batch=0, pack_min_batch=0):
        check(_lib.lib().mvq_stack_set_plan(self.handle, int(vpack_min_batch), int(pack_min_batch)), "mvq_stack_set_plan")

    def out_len(self, t):
        L = _lib.lib()
        return int(L.mvq_encoder_out_len(self.handle, int(t)) if self.kind == "encoder" else L.mvq_decoder_out_len(self.handle, int(t)))

    def _ws(self, x, B, t):
        L = _lib.lib()
        n = L.mvq_encoder_workspace_bytes(self.handle, B, t) if self.kind == "encoder" else L.mvq_decoder_workspace_bytes(self.handle, B, t)
        return torch.empty(max(int(n), 256), device=x.device, dtype=torch.uint8)

    def encoder_fwd(self, x):
        """x[B, 1, T] -> z[B, d_latent, Tl]: mvq_encoder_fwd_f32."""
        x = _dev(x, "x")
        B, _, T = x.shape
        z = torch.empty(B, self.desc.d_latent, max(self.out_len(T), 0), device=x.device, dtype=torch.float32)
        if z.numel():
            ws = self._ws(x, B, T)
            check(_lib.lib().mvq_encoder_fwd_f32(self.handle, x.data_ptr(), z.data_ptr(), ws.data_ptr(), ws.numel(), B, T, _stream()), "mvq_encoder_fwd_f32")
        return z

    def decoder_fwd(self, z):
        """z[B, C, t] -> y[B, d_out, Tout] (tanh applied): mvq_decoder_fwd_f32."""
        z = _dev(z, "z")
        B, _, t = z.shape
        y = torch.empty(B, self.desc.d_out, max(self.out_len(t), 0), device=z.device, dtype=torch.float32)
        if y.numel():
            ws = self._ws(z, B, t)
            check(_lib.lib().mvq_decoder_fwd_f32(self.handle, z.data_ptr(), y.data_ptr(), ws.data_ptr(), ws.numel(), B, t, _stream()), "mvq_decoder_fwd_f32")
        return y

    def decoder_fwd_saving(self, z):
        """-> (y, saved): the training forward; `saved` (one uint8 tensor) goes to decoder_bwd_input."""
        z = _dev(z, "z")
        B, _, t = z.shape
        L = _lib.lib()
        y = torch.empty(B, self.desc.d_out, self.out_len(t), device=z.device, dtype=torch.float32)
        saved = torch.empty(int(L.mvq_decoder_saved_bytes(self.handle, B, t)), device=z.device, dtype=torch.uint8)
        ws = self._ws(z, B, t)
        check(L.mvq_decoder_fwd_saving_f32(self.handle, z.data_ptr(), y.data_ptr(), saved.data_ptr(), saved.numel(), ws.data_ptr(), ws.numel(), B, t,
                                           _stream()), "mvq_decoder_fwd_saving_f32")
        return y, saved

    def decoder_bwd_input(self, saved, gy, batch, t):
        """dL/dz [batch, C, t] from dL/dy and the saved forward: mvq_decoder_bwd_input_f32."""
        gy = _dev(gy, "gy")
        gz = torch.empty(batch, self.desc.input_channel, t, device=gy.device, dtype=torch.float32)
        ws = self._ws(gy, batch, t)
        check(_lib.lib().mvq_decoder_bwd_input_f32(self.handle, saved.data_ptr(), saved.numel(), gy.data_ptr(), gz.data_ptr(), ws.data_ptr(), ws.numel(),
                                                   batch, t, _stream()), "mvq_decoder_bwd_input_f32")
        return gz
