"""CPU oracle for the encode -> VQ -> decode hot path (TEST INFRASTRUCTURE ONLY).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package; the product package never does.  numpy in / numpy out; the arithmetic lives in
``oracle/c/oracle.c`` (fp32 fma chains in a stated order + ``det_math.h``), this file is the glue that
mirrors the reference's Python structure.

Reference rows restated here (paths relative to /root/reference):
  * ``AllPredAR.forward_step`` / ``ProposedEval.encode_latents`` / ``forward_eval``
      Training/compare_dacvsproposal_5.py:292-326, Evaluation/dac_vcpwq_proposed6_latency.py:451-487
  * ``CrossPredictor.forward``, ``TokenNorm``, ``PosEnc1D``   Training/compare_dacvsproposal_5.py:214-244
  * ``ResidualVQEMA.forward`` / ``ema_step``                   Training/compare_dacvsproposal_5.py:246-277
  * ``psnr_batch`` Evaluation/compare_dacvsproposal_5_eval.py:180-185,
    ``psnr_global_peak_db`` Evaluation/dac_vcpwq_proposed6_latency.py:204-214
  * the third-party ``dac.DAC`` (24 kHz) encoder / quantizer / decoder the reference calls as black
    boxes (Training/compare_dacvsproposal_5.py:294-296,322,329-338).  ``descript-audio-codec`` is not
    in /root/reference and not installed: its architecture is restated from SURVEY.md section 8a.
    PARITY UNPINNED for those rows (no upstream code or golden vectors available offline).
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_SO = _HERE / "_build" / "liboracle.so"
_lib = None

f32p = ctypes.POINTER(ctypes.c_float)
i32p = ctypes.POINTER(ctypes.c_int32)


def build(force: bool = False) -> Path:
    """Compile oracle/c/oracle.c with gcc (make)."""
    if force or not _SO.exists() or _SO.stat().st_mtime < max(
            (_HERE / "c" / "oracle.c").stat().st_mtime, (_HERE / "c" / "det_math.h").stat().st_mtime):
        subprocess.run(["make", "-C", str(_HERE / "c"), "-B"], check=True, capture_output=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(str(_SO))
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(f32p)


def _opt(a):
    if a is None:
        return None, None
    return _f(a)


# ------------------------------------------------------------------------------------------ primitives
def weight_norm(v: np.ndarray, g: np.ndarray) -> np.ndarray:
    v_, vp = _f(v)
    g_, gp = _f(np.reshape(g, -1))
    out = np.empty_like(v_)
    rows = v_.shape[0]
    lib().orc_weight_norm(vp, gp, out.ctypes.data_as(f32p), rows, int(v_.size // rows))
    return out


def conv1d_out_len(Tin, ks, stride=1, dil=1, pad=0):
    span = Tin + 2 * pad - dil * (ks - 1) - 1
    return 0 if span < 0 else span // stride + 1


def conv1d(x, w, bias=None, stride=1, dil=1, pad=0, alpha_in=None, residual=None, alpha_out=None,
           tanh=False):
    x_, xp = _f(x)
    w_, wp = _f(w)
    B, Cin, Tin = x_.shape
    Cout, Cin2, ks = w_.shape
    assert Cin2 == Cin
    Tout = conv1d_out_len(Tin, ks, stride, dil, pad)
    y = np.zeros((B, Cout, Tout), np.float32)
    b_, bp = _opt(bias)
    ai_, aip = _opt(None if alpha_in is None else np.reshape(alpha_in, -1))
    ao_, aop = _opt(None if alpha_out is None else np.reshape(alpha_out, -1))
    r_, rp = _opt(residual)
    if r_ is not None:
        assert r_.shape == y.shape
    lib().orc_conv1d(xp, wp, bp, y.ctypes.data_as(f32p), B, Cin, Tin, Cout, ks, stride, dil, pad,
                     aip, rp, aop, 1 if tanh else 0)
    return y


def conv_transpose1d(x, w, bias=None, stride=1, pad=0, alpha_in=None, alpha_out=None, output_padding=0):
    x_, xp = _f(x)
    w_, wp = _f(w)
    B, Cin, Tin = x_.shape
    Cin2, Cout, ks = w_.shape
    assert Cin2 == Cin and 0 <= output_padding < max(stride, 1)
    Tout = (Tin - 1) * stride - 2 * pad + ks + output_padding
    y = np.zeros((B, Cout, Tout), np.float32)
    b_, bp = _opt(bias)
    ai_, aip = _opt(None if alpha_in is None else np.reshape(alpha_in, -1))
    ao_, aop = _opt(None if alpha_out is None else np.reshape(alpha_out, -1))
    lib().orc_conv_transpose1d_op(xp, wp, bp, y.ctypes.data_as(f32p), B, Cin, Tin, Cout, ks, stride, pad, int(output_padding),
                                  aip, aop)
    return y


def snake(x, alpha):
    x_, xp = _f(x)
    a_, ap = _f(np.reshape(alpha, -1))
    y = np.empty_like(x_)
    B, C, T = x_.shape
    lib().orc_snake(xp, ap, y.ctypes.data_as(f32p), B, C, T)
    return y


def _unary(name, x):
    x_, xp = _f(x)
    y = np.empty_like(x_)
    fn = getattr(lib(), name)
    fn.argtypes = [f32p, f32p, ctypes.c_size_t]
    fn(xp, y.ctypes.data_as(f32p), x_.size)
    return y


def gelu(x): return _unary("orc_gelu", x)
def tanh(x): return _unary("orc_tanh", x)
def sin(x): return _unary("orc_sin", x)
def sin2_turns(t): return _unary("orc_sin2_turns", t)     # sin(pi t)^2
def sin_turns(t): return _unary("orc_sin_turns", t)       # sin(pi t)
def exp(x): return _unary("orc_exp", x)
def erf(x): return _unary("orc_erf", x)


def rvq_ema_forward(z, books, n_books_use=None, return_residual=False):
    """ResidualVQEMA.forward on z[B,D,T]; books = sequence of [K,D].  Returns (q[B,D,T], idx[nb,B*T])."""
    z_ = np.ascontiguousarray(z, np.float32)
    B, D, T = z_.shape
    nb = len(books) if n_books_use is None else min(int(n_books_use), len(books))
    x = np.ascontiguousarray(z_.transpose(0, 2, 1).reshape(B * T, D))
    bk = np.ascontiguousarray(np.stack([np.asarray(b, np.float32) for b in books[:nb]]) if nb else
                              np.zeros((0, 1, D), np.float32))
    K = bk.shape[1]
    idx = np.zeros((nb, B * T), np.int32)
    qs = np.zeros((B * T, D), np.float32)
    res = np.zeros((B * T, D), np.float32)
    lib().orc_rvq_ema_forward(x.ctypes.data_as(f32p), bk.ctypes.data_as(f32p), nb, K, D, B * T,
                              idx.ctypes.data_as(i32p), qs.ctypes.data_as(f32p), res.ctypes.data_as(f32p))
    q = np.ascontiguousarray(qs.reshape(B, T, D).transpose(0, 2, 1))
    if return_residual:
        return q, idx, res
    return q, idx


def rvq_margins(x_tokens, emb):
    x_, xp = _f(x_tokens)
    e_, ep = _f(emb)
    m = np.zeros(x_.shape[0], np.float32)
    lib().orc_rvq_margins(xp, ep, e_.shape[0], e_.shape[1], x_.shape[0], m.ctypes.data_as(f32p))
    return m


def rvq_ema_step(z_tokens, books, decay=0.99):
    """ResidualVQEMA.ema_step; returns (new_books[nb,K,D], idx[nb,N])."""
    z_ = np.ascontiguousarray(z_tokens, np.float32)
    B, D, T = z_.shape
    X = np.ascontiguousarray(z_.transpose(0, 2, 1).reshape(B * T, D))
    bk = np.ascontiguousarray(np.stack([np.asarray(b, np.float32) for b in books])).copy()
    nb, K, _ = bk.shape
    idx = np.zeros((nb, B * T), np.int32)
    fn = lib().orc_rvq_ema_step
    fn.argtypes = [f32p, f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, i32p]
    fn(X.ctypes.data_as(f32p), bk.ctypes.data_as(f32p), nb, K, D, B * T, float(np.float32(decay)),
       idx.ctypes.data_as(i32p))
    return bk, idx


def layernorm_c(x, gamma, beta, eps=1e-5, do_tanh=False, post_scale=1.0):
    x_, xp = _f(x)
    g_, gp = _f(gamma)
    b_, bp = _f(beta)
    B, C, T = x_.shape
    y = np.empty_like(x_)
    fn = lib().orc_layernorm_c
    fn.argtypes = [f32p, f32p, f32p, f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                   ctypes.c_int, ctypes.c_float]
    fn(xp, gp, bp, y.ctypes.data_as(f32p), B, C, T, float(np.float32(eps)), int(do_tanh),
       float(np.float32(post_scale)))
    return y


def attention(Q, K, V, heads):
    Q_, qp = _f(Q)
    K_, kp = _f(K)
    V_, vp = _f(V)
    B, C, Tq = Q_.shape
    Tk = K_.shape[2]
    ctx = np.zeros_like(Q_)
    if Tq:
        lib().orc_attention(qp, kp, vp, ctx.ctypes.data_as(f32p), B, heads, C // heads, Tq, Tk)
    return ctx


# ------------------------------------------------------------------------------- DAC-24k architecture
ENC_RATES = (2, 4, 5, 8)
DEC_RATES = (8, 5, 4, 2)
ENC_DIM = 64
DEC_DIM = 1536
LATENT = 1024
N_CODEBOOKS = 32
CODEBOOK_SIZE = 1024
CODEBOOK_DIM = 8


def _wn(sd, prefix):
    """Fold weight_g / weight_v under `prefix` (old-style weight_norm names)."""
    return weight_norm(sd[prefix + ".weight_v"], sd[prefix + ".weight_g"]), np.asarray(sd[prefix + ".bias"], np.float32)


def _residual_unit(sd, p, x, dil):
    """ResidualUnit: x + conv1(snake(conv7(snake(x))))   [upstream dac/model/dac.py ResidualUnit]"""
    w7, b7 = _wn(sd, p + ".block.1")
    w1, b1 = _wn(sd, p + ".block.3")
    h = conv1d(x, w7, b7, dil=dil, pad=3 * dil, alpha_in=sd[p + ".block.0.alpha"])
    return conv1d(h, w1, b1, alpha_in=sd[p + ".block.2.alpha"], residual=x)


def dac_encoder(sd, x, rates=ENC_RATES, prefix="", collect=None):
    """Encoder.forward: conv k7 -> 4x EncoderBlock -> snake -> conv k3.  x[B,1,T] -> [B,1024,T/320]."""
    P = prefix
    w, b = _wn(sd, P + "block.0")
    h = conv1d(x, w, b, pad=3)
    if collect is not None: collect.append(("enc.in", h))
    for i, s in enumerate(rates):
        p = f"{P}block.{i + 1}"
        for j, dil in enumerate((1, 3, 9)):
            h = _residual_unit(sd, f"{p}.block.{j}", h, dil)
        w, b = _wn(sd, p + ".block.4")
        h = conv1d(h, w, b, stride=s, pad=math.ceil(s / 2), alpha_in=sd[p + ".block.3.alpha"])
        if collect is not None: collect.append((f"enc.b{i}", h))
    n = len(rates) + 1
    w, b = _wn(sd, f"{P}block.{n + 1}")
    h = conv1d(h, w, b, pad=1, alpha_in=sd[f"{P}block.{n}.alpha"])
    if collect is not None: collect.append(("enc.out", h))
    return h


def dac_decoder(sd, z, rates=DEC_RATES, prefix="", collect=None, output_padding=False):
    """Decoder.forward: conv k7 -> 4x DecoderBlock(snake, convT, 3 RU) -> snake -> conv k7 -> tanh.
    output_padding=True: the DecoderBlock variant with ``output_padding = stride % 2`` on its ConvTranspose1d (believed to be
    upstream's repository head; the 1.0.0 release -- the default here -- has none): 75 tokens -> 24 000 samples, not 23 992."""
    P = prefix
    w, b = _wn(sd, P + "model.0")
    h = conv1d(z, w, b, pad=3)
    if collect is not None: collect.append(("dec.in", h))
    for i, s in enumerate(rates):
        p = f"{P}model.{i + 1}"
        w, b = _wn(sd, p + ".block.1")          # ConvTranspose1d weight [Cin,Cout,k], WN over dim 0
        h = conv_transpose1d(h, w, b, stride=s, pad=math.ceil(s / 2), alpha_in=sd[p + ".block.0.alpha"],
                             output_padding=(s % 2) if output_padding else 0)
        for j, dil in enumerate((1, 3, 9)):
            h = _residual_unit(sd, f"{p}.block.{j + 2}", h, dil)
        if collect is not None: collect.append((f"dec.b{i}", h))
    n = len(rates) + 1
    w, b = _wn(sd, f"{P}model.{n + 1}")
    y = conv1d(h, w, b, pad=3, alpha_in=sd[f"{P}model.{n}.alpha"], tanh=True)
    if collect is not None: collect.append(("dec.out", y))
    return y


def dac_quantizer(sd, z, n_quantizers=None, prefix=""):
    """ResidualVectorQuantize.forward (eval): returns (z_q, codes[B,nq,T] int64, latents[B,nq*8,T], 0., 0.)."""
    z_ = np.ascontiguousarray(z, np.float32)
    B, C, T = z_.shape
    nq_all = 0
    while f"{prefix}quantizers.{nq_all}.codebook.weight" in sd:
        nq_all += 1
    nq = nq_all if n_quantizers is None else min(int(n_quantizers), nq_all)
    in_w, in_b, cbs, out_w, out_b = [], [], [], [], []
    for i in range(nq):
        p = f"{prefix}quantizers.{i}"
        w, b = _wn(sd, p + ".in_proj"); in_w.append(w.reshape(w.shape[0], -1)); in_b.append(b)
        w, b = _wn(sd, p + ".out_proj"); out_w.append(w.reshape(w.shape[0], -1)); out_b.append(b)
        cbs.append(np.asarray(sd[p + ".codebook.weight"], np.float32))
    in_w = np.ascontiguousarray(np.stack(in_w)); in_b = np.ascontiguousarray(np.stack(in_b))
    out_w = np.ascontiguousarray(np.stack(out_w)); out_b = np.ascontiguousarray(np.stack(out_b))
    cb = np.ascontiguousarray(np.stack(cbs))
    K, Dc = cb.shape[1], cb.shape[2]
    zq = np.zeros_like(z_)
    codes = np.zeros((B, nq, T), np.int32)
    lat = np.zeros((B, nq * Dc, T), np.float32)
    lib().orc_dac_rvq(z_.ctypes.data_as(f32p), in_w.ctypes.data_as(f32p), in_b.ctypes.data_as(f32p),
                      cb.ctypes.data_as(f32p), out_w.ctypes.data_as(f32p), out_b.ctypes.data_as(f32p),
                      B, C, T, nq, K, Dc, zq.ctypes.data_as(f32p), codes.ctypes.data_as(i32p),
                      lat.ctypes.data_as(f32p))
    return zq, codes.astype(np.int64), lat, np.float32(0), np.float32(0)


# --------------------------------------------------------------------- reference-owned model (proposed)
AR_CHUNK_TOK = 16   # Training/compare_dacvsproposal_5.py:65
CODE_DIM = 96       # ...:68


def pos_table(c, max_len=8192):
    """PosEnc1D buffer, built exactly as the reference does (torch CPU ops), Training/...5.py:214-220."""
    import torch
    pe = torch.zeros(max_len, c)
    pos = torch.arange(0, max_len).unsqueeze(1)
    div = torch.exp(torch.arange(0, c, 2) * (-math.log(10000.0) / c))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.numpy()


def _linear(x, w, b=None, residual=None):
    """nn.Linear applied on the channel axis of channel-major x[B,C,T] (== 1x1 conv)."""
    return conv1d(x, np.asarray(w, np.float32)[:, :, None], b, residual=residual)


def cross_predictor(sd, zt_prev, za, pe, heads=8, prefix="predict."):
    """CrossPredictor.forward (eval mode: dropout inert).  zt_prev[B,C,Tq], za[B,C,Tk] -> [B,C,Tq]."""
    P = prefix
    zt_prev = np.asarray(zt_prev, np.float32); za = np.asarray(za, np.float32)
    Tq, Tk = zt_prev.shape[2], za.shape[2]
    q = zt_prev + pe[:Tq].T[None]
    kv = za + pe[:Tk].T[None]
    q = layernorm_c(q, sd[P + "ln_q.weight"], sd[P + "ln_q.bias"])
    kv = layernorm_c(kv, sd[P + "ln_kv.weight"], sd[P + "ln_kv.bias"])
    Q = _linear(q, sd[P + "q_proj.weight"])
    K = _linear(kv, sd[P + "k_proj.weight"])
    V = _linear(kv, sd[P + "v_proj.weight"])
    ctx = attention(Q, K, V, heads)
    y1 = _linear(ctx, sd[P + "out.weight"], residual=q)                      # y + q
    h = layernorm_c(y1, sd[P + "ffn.0.weight"], sd[P + "ffn.0.bias"])
    h = gelu(_linear(h, sd[P + "ffn.1.weight"], sd[P + "ffn.1.bias"]))
    return _linear(h, sd[P + "ffn.3.weight"], sd[P + "ffn.3.bias"], residual=y1)   # ffn(y+q) + (y+q)


def proposed_encode_latents(sd, a, t, books_use=None, pe=None, return_aux=False, tactile_only=False):
    """ProposedEval.encode_latents / AllPredAR.forward_step (forward part).

    sd holds the checkpoint names of the reference model: A_ENC.*, A_QUANT.*, T_ENC.*, T_DEC.*,
    predict.*, tokennorm.ln.*, scale, proj_down.*, proj_up.*, vq.books.{i}."""
    za = qa = None
    if not tactile_only:      # tactile_only: SURVEY.md section 8d config 2 (z_pred == 0, no audio branch)
        za = dac_encoder(sd, a, prefix="A_ENC.")
        qa = dac_quantizer(sd, za, prefix="A_QUANT.")[0]
    zt = dac_encoder(sd, t, prefix="T_ENC.")
    B, C, Tlat = zt.shape
    if pe is None:
        pe = np.asarray(sd["predict.pos.pe"], np.float32) if "predict.pos.pe" in sd else pos_table(C)
    books = []
    while f"vq.books.{len(books)}" in sd:
        books.append(np.asarray(sd[f"vq.books.{len(books)}"], np.float32))
    scale = np.float32(min(max(float(np.float32(sd["scale"])), 5e-3), 0.5))
    z_run = np.zeros_like(zt)
    rD_all, idx_all = [], []
    for s in range(0, Tlat, AR_CHUNK_TOK):
        e = min(Tlat, s + AR_CHUNK_TOK)
        zt_prev = np.zeros((B, C, e - s), np.float32)
        if s == 0:
            zt_prev[..., 1:] = z_run[..., s:e - 1]
        else:
            zt_prev[...] = z_run[..., s - 1:e - 1]
        if tactile_only:
            z_pred = None
            r = zt[..., s:e]
        else:
            z_pred = cross_predictor(sd, zt_prev, qa[..., s:e], pe)
            r = zt[..., s:e] - z_pred
        rN = layernorm_c(r, sd["tokennorm.ln.weight"], sd["tokennorm.ln.bias"], do_tanh=True, post_scale=scale)
        rD = conv1d(rN, np.asarray(sd["proj_down.weight"], np.float32), sd["proj_down.bias"])
        qD, idx = rvq_ema_forward(rD, books, books_use)
        z_hat = conv1d(qD, np.asarray(sd["proj_up.weight"], np.float32), sd["proj_up.bias"], residual=z_pred)
        z_run[..., s:e] = z_hat
        rD_all.append(rD); idx_all.append(idx.reshape(idx.shape[0], B, e - s))
    if return_aux:
        return z_run, {"za": za, "qa": qa, "zt": zt, "r_tokens": np.concatenate(rD_all, -1),
                       "idx": np.concatenate(idx_all, -1)}
    return z_run


def proposed_forward_eval(sd, a, t, books_use=None, pe=None, tactile_only=False):
    z_run = proposed_encode_latents(sd, a, t, books_use, pe, tactile_only=tactile_only)
    return dac_decoder(sd, z_run, prefix="T_DEC.")


def psnr_batch(ref, est, eps=1e-12):
    """Evaluation/compare_dacvsproposal_5_eval.py:180-185 (peak 1.0), float64 accumulation."""
    ref = np.asarray(ref, np.float64); est = np.asarray(est, np.float64)
    mse = np.maximum(((ref - est) ** 2).mean(axis=(1, 2)), eps)
    return 10.0 * np.log10(1.0 / mse)


def psnr_global_peak_db(ref, est, peak, eps=1e-12):
    """Evaluation/dac_vcpwq_proposed6_latency.py:204-214."""
    ref = np.asarray(ref, np.float64).reshape(-1); est = np.asarray(est, np.float64).reshape(-1)
    mse = ((ref - est) ** 2).mean() + eps
    peak = max(float(peak), eps)
    return float(10.0 * np.log10(peak * peak / mse))


def align_by_xcorr(ref_1T, est_1T, max_shift=200):
    """Evaluation/dac_vcpwq_proposed6_latency.py:164-202 -> (ref_aligned[1,T'], est_aligned[1,T'], best_shift, corr)."""
    r_, rp = _f(np.reshape(ref_1T, -1))
    e_, ep = _f(np.reshape(est_1T, -1))
    assert r_.size == e_.size
    T = r_.size
    corr = np.zeros(2 * max_shift + 1, np.float32)
    s = lib().orc_align_xcorr(rp, ep, T, int(max_shift), corr.ctypes.data_as(f32p))
    if s < 0:
        r_a = r_[-s:]; e_a = e_[:r_a.size]
    elif s > 0:
        r_a = r_[:-s]; e_a = e_[s:s + r_a.size]
    else:
        r_a, e_a = r_, e_[:r_.size]
    return r_a[None], e_a[None], int(s), corr


# ------------------------------------------------------------------------------------- backward (row f1)
def mul_dsnake(g, x, alpha, residual=None):
    g_, gp = _f(g); x_, xp = _f(x); a_, ap = _f(np.reshape(alpha, -1))
    r_, rp = _opt(residual)
    out = np.empty_like(g_)
    B, C, T = g_.shape
    lib().orc_mul_dsnake(gp, xp, ap, rp, out.ctypes.data_as(f32p), B, C, T)
    return out


def mul_dtanh(g, y):
    g_, gp = _f(g); y_, yp = _f(y)
    out = np.empty_like(g_)
    fn = lib().orc_mul_dtanh
    fn.argtypes = [f32p, f32p, f32p, ctypes.c_size_t]
    fn(gp, yp, out.ctypes.data_as(f32p), g_.size)
    return out


def conv1d_dgrad(gy, w, dil=1, pad=0):
    """Input-gradient of y = conv1d(x, w[Cout,Cin,k], stride 1): conv1d(gy, w'), w'[ci,co,k] = w[co,ci,K-1-k];
    chain order = forward output channel, then tap (what orc_conv1d does with w')."""
    w = np.asarray(w, np.float32)
    wt = np.ascontiguousarray(w[:, :, ::-1].transpose(1, 0, 2))
    return conv1d(gy, wt, None, 1, dil, (w.shape[2] - 1) * dil - pad)


def conv_transpose1d_dgrad(gy, w, stride, pad):
    """Input-gradient of y = conv_transpose1d(x, w[Cin,Cout,k]): conv1d(gy, w viewed as [out=Cin, in=Cout, k], stride, pad)."""
    return conv1d(gy, np.ascontiguousarray(w, np.float32), None, stride, 1, pad)


def dac_decoder_saving(sd, z, rates=DEC_RATES, prefix=""):
    """Decoder forward that keeps every Snake input (same values as dac_decoder)."""
    P = prefix
    saved = {}
    w, b = _wn(sd, P + "model.0")
    h = conv1d(z, w, b, pad=3)
    for i, s in enumerate(rates):
        p = f"{P}model.{i + 1}"
        saved[f"b{i + 1}.x"] = h
        w, b = _wn(sd, p + ".block.1")
        h = conv_transpose1d(h, w, b, stride=s, pad=math.ceil(s / 2), alpha_in=sd[p + ".block.0.alpha"])
        for j, dil in enumerate((1, 3, 9)):
            q = f"{p}.block.{j + 2}"
            w7, b7 = _wn(sd, q + ".block.1"); w1, b1 = _wn(sd, q + ".block.3")
            t7 = conv1d(h, w7, b7, dil=dil, pad=3 * dil, alpha_in=sd[q + ".block.0.alpha"])
            saved[f"b{i + 1}.r{j + 2}.x"], saved[f"b{i + 1}.r{j + 2}.t7"] = h, t7
            h = conv1d(t7, w1, b1, alpha_in=sd[q + ".block.2.alpha"], residual=h)
    n = len(rates) + 1
    saved["hl"] = h
    w, b = _wn(sd, f"{P}model.{n + 1}")
    y = conv1d(h, w, b, pad=3, alpha_in=sd[f"{P}model.{n}.alpha"], tanh=True)
    saved["y"] = y
    return y, saved


def dac_decoder_backward_input(sd, saved, gy, rates=DEC_RATES, prefix=""):
    """dL/dz of the decoder (weights frozen): mirrors Decoder.backward_input of the product path."""
    P = prefix
    n = len(rates) + 1
    g = mul_dtanh(gy, saved["y"])
    w, _ = _wn(sd, f"{P}model.{n + 1}")
    g = mul_dsnake(conv1d_dgrad(g, w, 1, 3), saved["hl"], sd[f"{P}model.{n}.alpha"])
    for i in range(len(rates), 0, -1):
        p = f"{P}model.{i}"
        s = rates[i - 1]
        for j, dil in reversed(list(enumerate((1, 3, 9)))):
            q = f"{p}.block.{j + 2}"
            w7, _ = _wn(sd, q + ".block.1"); w1, _ = _wn(sd, q + ".block.3")
            x, t7 = saved[f"b{i}.r{j + 2}.x"], saved[f"b{i}.r{j + 2}.t7"]
            g1 = mul_dsnake(conv1d_dgrad(g, w1, 1, 0), t7, sd[q + ".block.2.alpha"])
            g = mul_dsnake(conv1d_dgrad(g1, w7, dil, 3 * dil), x, sd[q + ".block.0.alpha"], residual=g)
        w, _ = _wn(sd, p + ".block.1")
        g = mul_dsnake(conv_transpose1d_dgrad(g, w, s, math.ceil(s / 2)), saved[f"b{i}.x"], sd[p + ".block.0.alpha"])
    w, _ = _wn(sd, P + "model.0")
    return conv1d_dgrad(g, w, 1, 3)


# ------------------------------------------------------------------------------------- resampler (row f3)
def resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """torchaudio's default ``sinc_interp_hann`` kernel (torchaudio.functional.functional._get_sinc_resample_kernel),
    restated from the published algorithm -- torchaudio is absent from this image, so this row is PARITY UNPINNED
    against torchaudio itself and is validated by signal properties (tests/test_oracle_resample.py):
      g = gcd; orig, new = f/g; base = min(orig, new)*rolloff; width = ceil(lpw*orig/base)
      t[p][k] = (-p/new + (k - width)/orig) * base, clamped to [-lpw, lpw]
      kern[p][k] = sinc(pi t) * cos(pi t / (2 lpw))^2 * base/orig        (float64, then rounded to fp32)
    -> (kern[new, 2*width+orig] float32, width, orig, new)."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = int(math.ceil(lowpass_filter_width * orig / base))
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = (np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx) * base
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    with np.errstate(invalid="ignore", divide="ignore"):
        kern = np.where(t == 0, 1.0, np.sin(t) / t) * window * (base / orig)
    return kern.astype(np.float32), width, orig, new


def resample(x, orig_freq: int, new_freq: int, kern=None):
    """x[..., L] -> [..., ceil(new*L/orig)]  (identity when the rates are equal, as the reference short-circuits).
    ``kern``: use this filter bank instead of the oracle's own design (to compare the convolution alone)."""
    if int(orig_freq) == int(new_freq):
        return np.asarray(x, np.float32)
    kern0, width, orig, new = resample_kernel(orig_freq, new_freq)
    kern = kern0 if kern is None else np.ascontiguousarray(kern, np.float32)
    assert kern.shape == kern0.shape
    x_, xp = _f(x)
    lead, L = x_.shape[:-1], x_.shape[-1]
    B = int(np.prod(lead)) if lead else 1
    Lout = int(math.ceil(new * L / orig))
    y = np.empty((B, Lout), np.float32)
    k_, kp = _f(kern)
    lib().orc_resample(xp, kp, y.ctypes.data_as(f32p), B, L, Lout, orig, new, width, kern.shape[1])
    return y.reshape(lead + (Lout,))
