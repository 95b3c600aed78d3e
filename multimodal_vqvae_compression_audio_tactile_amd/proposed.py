"""The reference's own modules on the hot path, same names / constructor arguments / state-dict keys, running on
libmvq_hip.so:  PosEnc1D, TokenNorm, CrossPredictor, ResidualVQEMA, AllPredAR (forward), ProposedEval, ProposedWrapper.

Reference: Training/compare_dacvsproposal_5.py:214-326 (train-time classes) and
Evaluation/dac_vcpwq_proposed6_latency.py:339-487 (eval-time classes with ``n_books_use`` / ``encode_latents``).

MI355X-first differences that do not change results:
  * inside the AR loop every chunk tensor is kept TOKEN-FOLDED as [1, C, B*Tc] (column b*Tc+i), so the six
    predictor GEMMs, proj_down/up and the RVQ search see N = B*Tc contiguous columns instead of B tiny
    [C,16] problems; LayerNorm / attention take (batch, channel) strides for that layout;
  * PosEnc1D add is fused into the LayerNorm kernel, tanh and the clamp(scale) multiply into TokenNorm's,
    residual adds into GEMM epilogues;
  * the shift-by-one input ``zt_prev`` is built exactly as the reference does (only column 0 of a chunk with
    s > 0 is non-zero -- SURVEY.md section 3.1 "Observed data dependency").
Training (row f1 of SURVEY.md section 8): with autograd enabled, ``AllPredAR.forward_step`` builds the graph out of the
HIP-backed autograd Functions in train.py (same forward kernels, HIP backward kernels), so the reference's
``total.backward(); opt.step()`` works on these modules unchanged; ``net.train()`` enables the ctx dropout.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch
import torch.nn as nn

from . import ops, train
from ._lib import MvqError
from . import dac as _dac
from .dac import _Packed

CODE_DIM = 96        # Training/compare_dacvsproposal_5.py:68
AR_CHUNK_TOK = 16    # ...:65
EMA_DECAY = 0.99     # ...:69


class PosEnc1D(nn.Module):
    def __init__(self, c, max_len=8192):
        super().__init__()
        pe = torch.zeros(max_len, c)
        pos = torch.arange(0, max_len).unsqueeze(1)
        div = torch.exp(torch.arange(0, c, 2) * (-math.log(10000.0) / c))
        pe[:, 0::2] = torch.sin(pos * div)
        pe[:, 1::2] = torch.cos(pos * div)
        self.register_buffer("pe", pe)


class TokenNorm(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.ln = nn.LayerNorm(c)

    @torch.no_grad()
    def forward(self, z):
        return ops.layernorm_c(z, self.ln.weight.detach(), self.ln.bias.detach(), eps=self.ln.eps)


class _PackedLinear:
    """K-major packed image of an nn.Linear / 1x1 nn.Conv1d weight, rebuilt when the parameter changes."""

    def __init__(self, mod):
        self.mod = mod
        self.cache = _Packed()

    def wp(self):
        w = self.mod.weight
        return self.cache.get((w,), lambda: ops.pack_conv1d(w.detach().reshape(w.shape[0], w.shape[1], 1)))

    def wp_dgrad(self):
        """The input-gradient image of the same weight (train.Linear.backward), once per weight version instead of once per AR chunk."""
        w = self.mod.weight
        if not hasattr(self, "cache_d"):
            self.cache_d = _Packed()
        return self.cache_d.get((w,), lambda: ops.pack_conv1d_dgrad(w.detach().reshape(w.shape[0], w.shape[1], 1)))

    def __call__(self, x, residual=None, gelu=False):
        w = self.mod.weight
        b = self.mod.bias.detach() if getattr(self.mod, "bias", None) is not None else None
        if gelu and (w.shape[1] % 32 or w.shape[0] < 64):                  # GELU epilogue exists on the MFMA tiles only
            return ops.gelu(ops.conv1d(x, self.wp(), w.shape[0], 1, bias=b, residual=residual))
        return ops.conv1d(x, self.wp(), w.shape[0], 1, bias=b, residual=residual, gelu=gelu)


class CrossPredictor(nn.Module):
    def __init__(self, c, heads=8, mlp_mul=2, dropout=0.1):
        super().__init__()
        assert c % heads == 0
        self.pos = PosEnc1D(c)
        self.h, self.dh = heads, c // heads
        self.ln_q, self.ln_kv = nn.LayerNorm(c), nn.LayerNorm(c)
        self.q_proj, self.k_proj, self.v_proj = nn.Linear(c, c, False), nn.Linear(c, c, False), nn.Linear(c, c, False)
        self.out = nn.Linear(c, c, False)
        self.drop = nn.Dropout(dropout)
        self.ffn = nn.Sequential(nn.LayerNorm(c), nn.Linear(c, mlp_mul * c), nn.GELU(), nn.Linear(mlp_mul * c, c))
        self._lin = {n: _PackedLinear(m) for n, m in (("q", self.q_proj), ("k", self.k_proj), ("v", self.v_proj),
                                                      ("o", self.out), ("f1", self.ffn[1]), ("f3", self.ffn[3]))}

    @torch.no_grad()
    def keys_values(self, qa_folded, folded_batch, chunk):
        """K and V of EVERY chunk in one pass: the audio side of the predictor does not depend on the AR state, so the
        per-chunk LayerNorm + two GEMMs (15 launches per segment batch) collapse into 3 over all B*Ta tokens.  PosEnc1D
        restarts at each chunk (the reference applies it per chunk, Training/...5.py:305-309): row t of the table is
        pe[t mod chunk]."""
        Ta = qa_folded.shape[2] // folded_batch
        key = (Ta, chunk, str(qa_folded.device))
        if getattr(self, "_pe_tiled", (None,))[0] != key:
            reps = (Ta + chunk - 1) // chunk
            self._pe_tiled = (key, self.pos.pe[:chunk].repeat(reps, 1)[:Ta].contiguous())
        kv = ops.layernorm_c(qa_folded, self.ln_kv.weight.detach(), self.ln_kv.bias.detach(), pe=self._pe_tiled[1],
                             eps=self.ln_kv.eps, folded_batch=folded_batch)
        return self._lin["k"](kv), self._lin["v"](kv)

    @torch.no_grad()
    def run(self, zt_prev, za, folded_batch=None, kv_all=None, kv_slice=None):
        """zt_prev[B,C,Tq], za[B,C,Tk] (or both token-folded [1,C,B*T] with folded_batch=B) -> same layout.
        kv_all = keys_values(...) with kv_slice = (s, tk): attend to columns [s, s+tk) of the precomputed K / V."""
        fb = folded_batch
        pe = self.pos.pe
        q = ops.layernorm_c(zt_prev, self.ln_q.weight.detach(), self.ln_q.bias.detach(), pe=pe, eps=self.ln_q.eps,
                            folded_batch=fb)
        L = self._lin
        if kv_all is not None:
            ctx = ops.attention_kv_slice(L["q"](q), kv_all[0], kv_all[1], self.h, fb, kv_slice[0], kv_slice[1])
        else:
            kv = ops.layernorm_c(za, self.ln_kv.weight.detach(), self.ln_kv.bias.detach(), pe=pe, eps=self.ln_kv.eps,
                                 folded_batch=fb)
            ctx = ops.attention(L["q"](q), L["k"](kv), L["v"](kv), self.h, folded_batch=fb)
        y1 = L["o"](ctx, residual=q)                                        # out(ctx) + q
        hdn = ops.layernorm_c(y1, self.ffn[0].weight.detach(), self.ffn[0].bias.detach(), eps=self.ffn[0].eps,
                              folded_batch=fb)
        hdn = L["f1"](hdn, gelu=True)                                       # nn.GELU() in the GEMM epilogue
        return L["f3"](hdn, residual=y1)                                    # ffn(y) + y

    def run_train(self, zt_prev, za, folded_batch):
        """Same computation as run() on token-folded tensors, recorded for autograd (train.py Functions)."""
        fb, pe, L = folded_batch, self.pos.pe, self._lin
        lin = lambda n, mod, x, res=None: train.Linear.apply(x, mod.weight, getattr(mod, "bias", None), res, L[n])
        q = train.LayerNormC.apply(zt_prev, self.ln_q.weight, self.ln_q.bias, pe, self.ln_q.eps, fb)
        kv = train.LayerNormC.apply(za, self.ln_kv.weight, self.ln_kv.bias, pe, self.ln_kv.eps, fb)
        Q, K, V = lin("q", self.q_proj, q), lin("k", self.k_proj, kv), lin("v", self.v_proj, kv)
        ctx = train.Attention.apply(Q, K, V, self.h, fb)
        if self.training and self.drop.p > 0:
            ctx = train.Dropout.apply(ctx, float(self.drop.p))
        y1 = lin("o", self.out, ctx, q)
        hdn = train.LayerNormC.apply(y1, self.ffn[0].weight, self.ffn[0].bias, None, self.ffn[0].eps, fb)
        hdn = train.Gelu.apply(lin("f1", self.ffn[1], hdn))
        return lin("f3", self.ffn[3], hdn, y1)

    def forward(self, zt_prev, za):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            B, C, Tq = zt_prev.shape
            fold = lambda x: x.permute(1, 0, 2).reshape(1, C, -1).contiguous()
            y = self.run_train(fold(zt_prev), fold(za), B)
            return y.reshape(C, B, Tq).permute(1, 0, 2).contiguous()
        if self.training and self.drop.p > 0:
            raise MvqError("CrossPredictor: train-mode dropout needs autograd enabled; call .eval() for inference")
        return self.run(zt_prev, za)


class ResidualVQEMA(nn.Module):
    """Residual VQ with EMA codebooks.  ``forward(z[B,D,T], n_books_use=None)``; ``ema_step(z_tokens)``."""

    def __init__(self, dim: int, n_books: int, n_embed: int, decay: float = EMA_DECAY):
        super().__init__()
        self.books = nn.ParameterList([nn.Parameter(torch.randn(n_embed, dim) / math.sqrt(dim))
                                       for _ in range(n_books)])
        self.decay = float(decay)
        self.n_books, self.n_embed = int(n_books), int(n_embed)
        self._stack = _Packed()

    def stacked(self) -> torch.Tensor:
        """[n_books, K, D] view for the kernel.  Re-stacked on every call (1.5 MB at most): the reference updates the
        books through ``.data`` (ema_step), which no version counter sees, so nothing derived from them is cached."""
        return torch.stack([b.detach().float() for b in self.books]).contiguous()

    @torch.no_grad()
    def forward(self, z, n_books_use: Optional[int] = None, return_indices: bool = False):
        if len(self.books) == 0:
            return torch.zeros_like(z)
        return ops.rvq_ema_forward(z, self.stacked(), n_books_use, return_indices=return_indices)

    @torch.no_grad()
    def ema_step(self, z_tokens):
        books = self.stacked()
        ops.rvq_ema_step_(z_tokens, books, self.decay)
        for i, p in enumerate(self.books):
            p.data.copy_(books[i])


class _ProposedBase(nn.Module):
    def __init__(self, A_ENC, A_QUANT, T_ENC, T_DEC, c_lat, rvq_books, rvq_embed, decay=EMA_DECAY):
        super().__init__()
        self.A_ENC, self.A_QUANT, self.T_ENC, self.T_DEC = A_ENC, A_QUANT, T_ENC, T_DEC
        for m in [self.A_ENC, self.A_QUANT, self.T_ENC, self.T_DEC]:
            if m is not None:
                for p in m.parameters():
                    p.requires_grad_(False)
        self.predict = CrossPredictor(c=c_lat, heads=8, mlp_mul=2, dropout=0.1)
        self.tokennorm = TokenNorm(c_lat)
        self.scale = nn.Parameter(torch.tensor(0.08))
        self.proj_down = nn.Conv1d(c_lat, CODE_DIM, 1)
        self.proj_up = nn.Conv1d(CODE_DIM, c_lat, 1)
        self.vq = ResidualVQEMA(dim=CODE_DIM, n_books=rvq_books, n_embed=rvq_embed, decay=decay)
        self._pd, self._pu = _PackedLinear(self.proj_down), _PackedLinear(self.proj_up)
        self._scale_host = None

    def _scale_raw(self) -> float:
        """The scale parameter as a host float (one device read, cached per parameter version)."""
        key = (self.scale._version, self.scale.data_ptr())
        if self._scale_host is None or self._scale_host[0] != key:
            self._scale_host = (key, float(self.scale.detach().float().item()))
        return self._scale_host[1]

    def _scale_value(self) -> float:
        """clamp(scale, 5e-3, 0.5)   (Training/compare_dacvsproposal_5.py:314)."""
        return min(max(self._scale_raw(), 5e-3), 0.5)

    @torch.no_grad()
    def _ar_latents(self, qa, zt, books_use=None, want_tokens=False, tactile_only=False, want_indices=False):
        """The chunked AR loop (Training/...5.py:302-320 == Evaluation/...6_latency.py:461-477).
        ``want_indices``: also return the per-book code indices idx[n_books_use, B, Tlat] (int64)."""
        B, C, Tlat = zt.shape
        z_run = torch.zeros_like(zt)
        r_tokens = torch.empty(B, CODE_DIM, Tlat, device=zt.device, dtype=torch.float32) if want_tokens else None
        idx_all = [] if want_indices else None
        if B == 0 or Tlat == 0:                                               # empty batch / clip shorter than a token
            return (z_run, r_tokens, torch.zeros(0, B, Tlat, dtype=torch.int64, device=zt.device)) if want_indices \
                else (z_run, r_tokens)
        scale = self._scale_value()
        ln = self.tokennorm.ln
        books = self.vq.stacked() if len(self.vq.books) else None             # ONE stack per call, not one per chunk
        kv_all = None
        if not tactile_only:
            Ta = min(qa.shape[-1], Tlat)                                      # audio may be shorter (whole-file mode)
            if Ta > 0:                                                        # K, V of all chunks up front (3 launches)
                kv_all = self.predict.keys_values(ops.fold_time_slice(qa, 0, Ta), B, AR_CHUNK_TOK)
        mode = self._ar_one_call_mode(zt, books)
        if mode is not None:
            return self._ar_latents_fused(zt, z_run, r_tokens, kv_all, 0 if tactile_only else min(qa.shape[-1], Tlat), books, books_use,
                                          tactile_only, want_indices, staged=(mode == "staged"))
        zt_prev, zp_n = None, -1      # the shift-by-one input: all zero except column 0 of each item (s > 0), so one zeroed
        for s in range(0, Tlat, AR_CHUNK_TOK):                               # buffer per chunk width serves every chunk
            e = min(Tlat, s + AR_CHUNK_TOK)
            n = e - s
            zt_c = ops.fold_time_slice(zt, s, e)                             # [1,C,B*n]
            if tactile_only:
                z_pred = None
            else:
                if n != zp_n:
                    zt_prev, zp_n = torch.zeros(1, C, B * n, device=zt.device, dtype=torch.float32), n
                if s > 0:                                                     # column 0 <- z_run[..., s-1]
                    ops.fold_column_into_(zt_prev, 0, z_run, s - 1, B)
                ka = min(qa.shape[-1], e) - min(qa.shape[-1], s)
                if ka > 0:
                    z_pred = self.predict.run(zt_prev, None, folded_batch=B, kv_all=kv_all, kv_slice=(s, ka))
                else:
                    z_pred = self.predict.run(zt_prev, torch.zeros(1, C, 0, device=zt.device), folded_batch=B)
            rN = ops.layernorm_c(zt_c, ln.weight.detach(), ln.bias.detach(), eps=ln.eps, do_tanh=True, post_scale=scale,
                                 folded_batch=B, sub=z_pred)                  # tanh(TokenNorm(zt - z_pred)) * scale
            rD = self._pd(rN)                                                 # [1,96,B*n]
            if books is None:
                qD = torch.zeros_like(rD)
            elif want_indices:
                qD, idx = ops.rvq_ema_forward(rD, books, books_use, return_indices=True)
                idx_all.append(idx.reshape(idx.shape[0], B, n))
            else:
                qD = ops.rvq_ema_forward(rD, books, books_use)
            z_hat = self._pu(qD, residual=z_pred)
            ops.unfold_into_(z_run, s, z_hat, B)
            if want_tokens:
                ops.unfold_into_(r_tokens, s, rD, B)
        if want_indices:
            return z_run, r_tokens, torch.cat(idx_all, dim=2) if idx_all else torch.zeros(0, B, Tlat, dtype=torch.int64)
        return z_run, r_tokens

    # The loop as ONE persistent kernel (csrc/ar_fused.hip: eleven stages per chunk between grid-wide barriers), for up to this many
    # segments.  OFF by default (0): measured at the reference's operating points (tools/ar_fused_ab.py, gpurun_out/f8) it is bit-equal
    # to the launch-per-stage loop and SLOWER -- 1.22 against 0.99 ms for one segment, 1.74 against 1.09 ms for six.  The stages are
    # 5-45 us of dependent chain each; at that length the command processor already has the next launch queued behind the running
    # kernel, so a launch boundary costs less than the ~4 us a grid barrier (a device-scope atomic, an L2 write-back and an
    # invalidate across eight XCDs) does.  Kept as an opt-in (MVQ_AR_FUSED_MAX_BATCH) with its parity tests: it is the place where
    # per-stage clocks can be read (MVQ_AR_TIMING=1), which is how the LayerNorm / GEMM-epilogue round trips of round 5 were found.
    AR_FUSED_MAX_BATCH = int(_dac.HOST_ENV_SEEN.get("MVQ_AR_FUSED_MAX_BATCH", "0"))

    # The loop as ONE HOST CALL of the same stand-alone launches (mvq_ar_latents_staged_f32), for up to this many segments: at one
    # segment the Python loop's ~85 foreign calls and their allocations cost as much host time as the kernels take on the device
    # (eager encode 2.5 ms against 2.3 replayed as a graph).  Needs every GEMM in the latency form's range: 8 segments at most.
    AR_STAGED_MAX_BATCH = min(8, int(_dac.HOST_ENV_SEEN.get("MVQ_AR_STAGED_MAX_BATCH", "8")))

    def _ar_shapes_covered(self, zt, books):
        p = self.predict
        return (zt.is_cuda and zt.shape[0] > 0 and ops.get_arith() == "f32"
                and zt.shape[1] == 1024 and p.h == 8 and p.ffn[1].out_features == 2048 and CODE_DIM == 96 and p.ln_q.eps == p.ffn[0].eps
                and (books is None or (books.shape[1] <= 512 and books.shape[2] == CODE_DIM)))

    def _ar_one_call_mode(self, zt, books):
        """"fused" (opt-in persistent kernel), "staged" (one host call, stand-alone launches) or None (the Python loop)."""
        if self._ar_fused_wanted(zt, books):
            return "fused"
        if zt.shape[0] <= self.AR_STAGED_MAX_BATCH and self._ar_shapes_covered(zt, books):
            return "staged"
        return None

    def _ar_fused_wanted(self, zt, books):
        p = self.predict
        return (zt.is_cuda and 0 < zt.shape[0] <= self.AR_FUSED_MAX_BATCH and ops.get_arith() == "f32"
                and not torch.cuda.is_current_stream_capturing()            # a cooperative launch is not capturable
                and zt.shape[1] == 1024 and p.h == 8 and p.ffn[1].out_features == 2048 and CODE_DIM == 96 and p.ln_q.eps == p.ffn[0].eps
                and (books is None or (books.shape[1] <= 512 and books.shape[2] == CODE_DIM)))

    def _ar_latents_fused(self, zt, z_run, r_tokens, kv_all, t_audio, books, books_use, tactile_only, want_indices, staged=False):
        B, _, Tlat = zt.shape
        p, L, ln = self.predict, self.predict._lin, self.tokennorm.ln
        nb = 0 if books is None else (books.shape[0] if books_use is None else max(0, min(int(books_use), books.shape[0])))
        idx = torch.empty(nb, B, Tlat, device=zt.device, dtype=torch.int32) if want_indices else None
        det = lambda t: None if t is None else t.detach()
        ops.ar_latents_fused(
            zt.contiguous(), z_run, k_all=None if kv_all is None else kv_all[0], v_all=None if kv_all is None else kv_all[1],
            t_audio=t_audio if kv_all is not None else 0, pe=p.pos.pe, ln_q=(det(p.ln_q.weight), det(p.ln_q.bias)),
            wq=L["q"].wp(), wo=L["o"].wp(), ln_f=(det(p.ffn[0].weight), det(p.ffn[0].bias)), w1=L["f1"].wp(), b1=det(p.ffn[1].bias),
            w3=L["f3"].wp(), b3=det(p.ffn[3].bias), ln_eps=p.ln_q.eps, tok=(det(ln.weight), det(ln.bias)), tok_eps=ln.eps,
            scale=self._scale_value(), wd=self._pd.wp(), bd=det(self.proj_down.bias), wu=self._pu.wp(), bu=det(self.proj_up.bias),
            books=books, books_use=books_use, heads=p.h, c_ff=p.ffn[1].out_features, code_dim=CODE_DIM, r_tokens=r_tokens, idx_out=idx,
            tactile_only=tactile_only, chunk=AR_CHUNK_TOK, staged=staged)
        if want_indices:
            return z_run, r_tokens, idx.long()
        return z_run, r_tokens

    # The two encoder branches (qa = A_QUANT(A_ENC(a)), zt = T_ENC(t)) are independent; up to this many segments they run on two
    # HIP streams (round 1: 8 % at B = 1-8, 5 % at 32, 1 % at 48-64).  At 256 segments two streams still give 330.6 -> 329.7 ms
    # per step (the under-filled tail / latent-rate launches of one branch overlap the other's; gpurun_out/r3l, twice on one
    # box), but concurrent kernels stretch each other's durations, so the per-kernel HIP-event / rocprofv3 figures the bench
    # reports (roofline of the dominant kernel) would no longer describe a kernel running alone.  The default therefore keeps
    # one stream at throughput batch sizes; MVQ_TWO_STREAM_MAX_BATCH raises the cap.
    TWO_STREAM_MAX_BATCH = int(_dac.HOST_ENV_SEEN.get("MVQ_TWO_STREAM_MAX_BATCH", "64"))      # read once at import, reported by plan_overrides()

    def _encode_branches(self, a_1T, t_1T):
        """qa = A_QUANT(A_ENC(a)) and zt = T_ENC(t) are independent.  In the latency regime (few segments: every
        launch underfills the 256 CUs) the audio branch runs on a second HIP stream beside the tactile branch."""
        # The opt-in arithmetic modes (frozen, non-parity) keep ONE stream: round 5 saw the audio branch of the FIRST two-stream call
        # of a fresh model come back wrong in bf16x6 (B = 2, fixture G4) in 4 of 4 plain runs and in 0 of 12 runs with any
        # perturbation (a synchronisation, NaN-poisoned allocations, one more reference held, any single new kernel switched off);
        # the cause was not established, the exact path has never shown it.
        if a_1T.shape[0] > self.TWO_STREAM_MAX_BATCH or not a_1T.is_cuda or ops.get_arith() != "f32":
            za = self.A_ENC(a_1T)
            qa, *_ = self.A_QUANT(za)
            return qa, self.T_ENC(t_1T)
        cur = torch.cuda.current_stream()
        side = getattr(self, "_side_stream", None)
        if side is None or side.device != a_1T.device:
            side = torch.cuda.Stream(device=a_1T.device)
            self._side_stream = side
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            za = self.A_ENC(a_1T)
            qa, *_ = self.A_QUANT(za)
        zt = self.T_ENC(t_1T)
        cur.wait_stream(side)
        for x in (za, qa):
            x.record_stream(cur)
        return qa, zt

    @torch.no_grad()
    def encode_latents_with_indices(self, a_1T, t_1T, books_use=None):
        """encode_latents plus what a transmitter would send: -> (z_run, audio codes[B,32,Ta] of A_QUANT, RVQ idx[n_books_use,B,Tlat])."""
        za = self.A_ENC(a_1T)
        qa, codes, *_ = self.A_QUANT(za)
        z_run, _, idx = self._ar_latents(qa, self.T_ENC(t_1T), books_use, want_indices=True)
        return z_run, codes, idx

    def _ar_latents_train(self, qa, zt):
        """The same loop recorded for autograd: z_hat of chunk c feeds column 0 of chunk c+1's zt_prev WITH gradient
        (the reference writes z_hat into z_run in place and slices it back, Training/...5.py:303-319)."""
        B, C, Tlat = zt.shape
        dev = zt.device
        ln = self.tokennorm.ln
        scale_raw = self._scale_raw()
        books = self.vq.stacked()
        chunks, r_toks, prev = [], [], None
        for s in range(0, Tlat, AR_CHUNK_TOK):
            e = min(Tlat, s + AR_CHUNK_TOK)
            n = e - s
            zt_c = ops.fold_time_slice(zt, s, e)
            if prev is None:
                zt_prev = torch.zeros(1, C, B * n, device=dev, dtype=torch.float32)
            else:
                last = prev.reshape(C, B, -1)[:, :, -1:]                                 # z_run[..., s-1], with graph
                zt_prev = torch.cat([last, torch.zeros(C, B, n - 1, device=dev)], dim=2).reshape(1, C, B * n)
            ka = min(qa.shape[-1], e) - min(qa.shape[-1], s)
            qa_c = ops.fold_time_slice(qa, s, s + ka) if ka > 0 else torch.zeros(1, C, 0, device=dev)
            z_pred = self.predict.run_train(zt_prev, qa_c, B)
            r = ops.sub(zt_c, z_pred.detach())
            u = train.LayerNormC.apply(r, ln.weight, ln.bias, None, ln.eps, B)
            rN = train.ScaleTanh.apply(u, self.scale, scale_raw)
            rD = train.Linear.apply(rN, self.proj_down.weight, self.proj_down.bias, None, self._pd)
            qD = train.RvqSte.apply(rD, books, None)
            z_hat = train.Linear.apply(qD, self.proj_up.weight, self.proj_up.bias, z_pred, self._pu)
            chunks.append(z_hat.reshape(C, B, n))
            r_toks.append(rD.detach().reshape(CODE_DIM, B, n))
            prev = z_hat
        z_run = torch.cat(chunks, dim=2).permute(1, 0, 2).contiguous()
        r_tokens = torch.cat(r_toks, dim=2).permute(1, 0, 2).contiguous()
        return z_run, r_tokens


class ProposedEval(_ProposedBase):
    """Evaluation/dac_vcpwq_proposed6_latency.py:437-487."""

    @torch.no_grad()
    def encode_latents(self, a_1T, t_1T, books_use=None):
        qa, zt = self._encode_branches(a_1T, t_1T)
        return self._ar_latents(qa, zt, books_use)[0]

    @torch.no_grad()
    def forward_eval(self, a_1T, t_1T, books_use=None):
        return self.T_DEC(self.encode_latents(a_1T, t_1T, books_use=books_use))

    @torch.no_grad()
    def encode_latents_tactile_only(self, t_1T, books_use=None):
        """BASELINE.json configs[1]: the tactile-side chain with z_pred == 0:
        T_ENC -> tanh(TokenNorm)*scale -> proj_down -> RVQ -> proj_up."""
        zt = self.T_ENC(t_1T)
        z_run, _ = self._ar_latents(None, zt, books_use, tactile_only=True)
        return z_run

    @torch.no_grad()
    def forward_eval_tactile_only(self, t_1T, books_use=None):
        return self.T_DEC(self.encode_latents_tactile_only(t_1T, books_use))


RVQ_N_BOOKS_MAX = 10  # Evaluation/compare_dacvsproposal_3.5_eval.py:68
RVQ_EMBED = 128       # ...:69


class ProposedWrapper(_ProposedBase):
    """The eval wrapper of Evaluation/compare_dacvsproposal_3.5_eval.py:374-411: constructor
    ``(A_ENC, A_QUANT, T_ENC, T_DEC, c_lat)`` -- the RVQ shape comes from that script's module constants
    ``RVQ_N_BOOKS_MAX = 10`` x ``RVQ_EMBED = 128`` (...:68-69; built at ...:485) -- and ``forward_eval(a, t, books_use)`` with a
    REQUIRED ``books_use`` (swept over 1..3 at ...:504).  Its AR loop writes ``zt_prev[...] = z_run[..., s-1:e-1]`` for
    s > 0 and ``zt_prev[..., 1:] = z_run[..., s:e-1]`` for s == 0 (...:393-396): the entries of z_run read for positions
    1.. have not been written yet, so -- exactly as in the other scripts -- only column 0 of a chunk with s > 0 is non-zero
    and the launch plan is ``_ar_latents`` unchanged."""

    def __init__(self, A_ENC, A_QUANT, T_ENC, T_DEC, c_lat):
        super().__init__(A_ENC, A_QUANT, T_ENC, T_DEC, c_lat, RVQ_N_BOOKS_MAX, RVQ_EMBED)

    @torch.no_grad()
    def forward_eval(self, a_1T, t_1T, books_use: int):
        qa, zt = self._encode_branches(a_1T, t_1T)
        return self.T_DEC(self._ar_latents(qa, zt, int(books_use))[0])


class AllPredAR(_ProposedBase):
    """Training/compare_dacvsproposal_5.py:279-326.  Under ``torch.no_grad()`` (validation, ...:411-414) this is the
    fused inference path; with autograd enabled it records the HIP-backed graph of train.py for ``.backward()``."""

    def _forward_step(self, a_1T, tc_1T):
        Tw = tc_1T.shape[-1]
        with torch.no_grad():                                                  # frozen backbones: no graph
            qa, zt = self._encode_branches(a_1T, tc_1T)
        if torch.is_grad_enabled() and zt.numel() and any(p.requires_grad for p in self.parameters()):
            z_run, r_tokens = self._ar_latents_train(qa, zt)
            y_hat = self.T_DEC(z_run)                                          # _DecoderInputGrad: HIP backward w.r.t. z
        else:
            with torch.no_grad():
                z_run, r_tokens = self._ar_latents(qa, zt, None, want_tokens=True)
                y_hat = self.T_DEC(z_run)
        T = min(y_hat.shape[-1], tc_1T.shape[-1], Tw)
        fz = lambda x: torch.nan_to_num(x, nan=0.0, posinf=0.0, neginf=0.0)   # finite_or_zero (...5.py:99-100)
        return {"y_hat": fz(y_hat[..., :T]), "tgt": fz(tc_1T[..., :T]), "r_tokens": r_tokens}, qa, zt

    def forward_step(self, a_1T, tc_1T):
        return self._forward_step(a_1T, tc_1T)[0]


class AllPredAR3(AllPredAR):
    """The ``AllPredAR`` of Training/compare_dacvsproposal_3.py:278-340 (BASELINE.json configs[0]): same model, but the
    constructor takes no sweep arguments (the script's module constants RVQ_N_BOOKS = 10, RVQ_EMBED = 128, ...:61-63) and
    ``forward_step`` additionally returns ``z_teacher`` (= T_ENC(tc), indexed by the script's ``step()``, ...:389-394) and
    ``z_pred`` (one more ``predict()`` call on an all-zero ``zt_prev`` over the first chunk, ...:334-337 -- unused by the
    loss, but in train mode it draws one more dropout mask, so it is issued at the same point of the RNG stream)."""

    def __init__(self, A_ENC, A_QUANT, T_ENC, T_DEC, c_lat, rvq_books: int = 10, rvq_embed: int = 128, decay=EMA_DECAY):
        super().__init__(A_ENC, A_QUANT, T_ENC, T_DEC, c_lat, rvq_books, rvq_embed, decay)

    def forward_step(self, a_1T, tc_1T):
        out, qa, zt = self._forward_step(a_1T, tc_1T)
        Tlat = zt.shape[-1]
        n = min(AR_CHUNK_TOK, Tlat)
        z_pred = None
        if Tlat > 0:
            ka = min(qa.shape[-1], n)
            z_pred = self.predict(torch.zeros_like(zt[..., :n]), qa[..., :ka].contiguous())
        return {"y_hat": out["y_hat"], "tgt": out["tgt"], "z_pred": z_pred, "z_teacher": zt,
                "r_tokens": out["r_tokens"] if Tlat > 0 else None}


def psnr_batch(ref_1T, est_1T, eps=1e-12):
    """PSNR(dB), peak = 1.0 (Evaluation/compare_dacvsproposal_5_eval.py:180-185)."""
    ref = ref_1T.to(torch.float32); est = est_1T.to(torch.float32)
    mse = (ref - est).pow(2).mean(dim=(1, 2)).clamp_min(eps)
    return [float(v) for v in (10.0 * torch.log10(1.0 / mse)).cpu()]


def psnr_global_peak_db(ref, est, peak, eps=1e-12):
    """Evaluation/dac_vcpwq_proposed6_latency.py:204-214."""
    ref = ref.reshape(-1).to(torch.float32); est = est.reshape(-1).to(torch.float32)
    mse = torch.mean((ref - est) ** 2) + eps
    peak = max(float(peak), eps)
    return float(10.0 * torch.log10((peak * peak) / mse).cpu())


def crop_match(a_1T, b_1T):
    """Evaluation/dac_vcpwq_proposed6_latency.py:158-160."""
    T = min(a_1T.shape[-1], b_1T.shape[-1])
    return a_1T[..., :T], b_1T[..., :T]


@torch.no_grad()
def align_by_xcorr(ref_1T, est_1T, max_shift=200):
    """Evaluation/dac_vcpwq_proposed6_latency.py:164-202: align est to ref by the integer shift that maximises the
    cross-correlation.  All 2*max_shift+1 correlations run in one launch; ONE device->host read (the shift)."""
    r = ref_1T.reshape(-1).to(torch.float32); e = est_1T.reshape(-1).to(torch.float32)
    _, best = ops.align_xcorr(r, e, max_shift)
    s = int(best.item())
    if s < 0:
        r_a = r[-s:]; e_a = e[: r_a.numel()]
    elif s > 0:
        r_a = r[:-s]; e_a = e[s: s + r_a.numel()]
    else:
        r_a = r; e_a = e[: r.numel()]
    return r_a.unsqueeze(0), e_a.unsqueeze(0), s


ALIGN_MAX_SHIFT_SAMPLES = 200      # Evaluation/compare_dacvsproposal_5_eval.py:69
EVAL_SR, ORIG_3K = 24000, 3000     # ...:52-53


@torch.no_grad()
def align_pair_24k(ref_24, est_24, max_shift=ALIGN_MAX_SHIFT_SAMPLES):
    """Evaluation/compare_dacvsproposal_5_eval.py:188-210 ([1,1,T] in, [1,1,T'] out)."""
    r_a, e_a, s = align_by_xcorr(ref_24.reshape(1, -1), est_24.reshape(1, -1), max_shift)
    return r_a.unsqueeze(0), e_a.unsqueeze(0), s


_DOWN_3K = {}          # device -> Resample(24 000, 3 000): the filter bank is designed once, not once per call


@torch.no_grad()
def psnr_3k_aligned_batch(ref_24, est_24, max_shift=ALIGN_MAX_SHIFT_SAMPLES):
    """Evaluation/compare_dacvsproposal_5_eval.py:212-223: per item, align at 24 kHz (+-200 samples), resample both to
    3 kHz, PSNR with peak 1.  Three launches for the whole batch and ONE device->host copy (the B PSNR values): the B
    alignments run in one launch pair (grid dimension = item), the slice bounds each shift implies (...:196-207) are computed
    on the device, the 2B ragged slices are resampled in one launch (mvq_resample_ragged_f32), and the squared error is
    reduced per row on the device.  (The reference syncs 401 + 1 times per item.)"""
    from .resample import Resample
    ref = ref_24.reshape(ref_24.shape[0], -1).to(torch.float32).contiguous()
    est = est_24.reshape(est_24.shape[0], -1).to(torch.float32).contiguous()
    B, T = ref.shape
    if B == 0:
        return []
    dev = ref.device
    s = ops.align_xcorr_batch(ref, est, max_shift)                       # int32 [B], stays on the device
    length = (T - s.abs()).to(torch.int32)                               # both aligned slices have T - |s| samples
    zero = torch.zeros_like(s)
    off = torch.cat([torch.maximum(-s, zero), torch.maximum(s, zero)]).to(torch.int32)     # ref starts at -s (s < 0), est at s (s > 0)
    down = _DOWN_3K.get(str(dev))
    if down is None:
        down = _DOWN_3K[str(dev)] = Resample(EVAL_SR, ORIG_3K).to(dev)
    pitch = (down.new * T + down.orig - 1) // down.orig
    y, lout = ops.resample_ragged(torch.cat([ref, est]), down.kernel, off, torch.cat([length, length]), down.orig, down.new,
                                  down.width, pitch)
    n3 = lout[:B].to(torch.float32).clamp_min(1.0)
    mse = ((y[:B] - y[B:]).pow(2).sum(dim=1) / n3).clamp_min(1e-12)    # rows are zero past their length on both sides
    return [float(v) for v in (10.0 * torch.log10(1.0 / mse)).cpu()]
