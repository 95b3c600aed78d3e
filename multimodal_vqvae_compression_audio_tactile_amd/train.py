"""Autograd for the reference's trainable head: torch.autograd.Function wrappers whose
forward AND backward run on libmvq_hip.so, so that the reference's own training step

    out = net.forward_step(a, tc); total = ...losses(out["y_hat"], out["tgt"]); total.backward(); opt.step()

(Training/compare_dacvsproposal_5.py:379-396) works unchanged on the drop-in modules: the trainable parameters are
exactly the reference's (CrossPredictor, TokenNorm, scale, proj_down, proj_up; vq.books are EMA-updated, never by
gradient, ...:367), the frozen DAC decoder passes the gradient through (dac.Decoder.backward_input), and the encoders
and audio quantiser run without a graph, as in the reference where their outputs carry no gradient path to a
trainable tensor.

All tensors here are in the token-folded layout [1, C, B*n] of proposed._ar_latents.  Parity bar: gradients equal
torch autograd on the torch restatement within fp32 tolerance (tests/test_gpu_train.py); the forward values are the
bit-exact inference kernels.
"""
from __future__ import annotations

import torch

from . import ops

Function = torch.autograd.Function


def _c(g):
    return g.contiguous() if g is not None else None


class Linear(Function):
    """y = W x + b (+ residual) over folded tokens.  W is [O, I] or [O, I, 1]."""

    @staticmethod
    def forward(ctx, x, w, b, residual, packed):
        y = packed(x.detach(), residual=residual.detach() if residual is not None else None)
        ctx.save_for_backward(x, w)
        ctx.has_b, ctx.has_res = b is not None, residual is not None
        ctx.packed = packed if (hasattr(packed, "wp_dgrad") and getattr(getattr(packed, "mod", None), "weight", None) is w) else None
        return y

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = _c(g)
        O, I = w.shape[0], w.shape[1]
        N = g.shape[-1]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            wp = ctx.packed.wp_dgrad() if ctx.packed is not None else ops.pack_conv1d_dgrad(w.detach().reshape(O, I, 1))
            gx = ops.conv1d_dgrad(g, wp, I, N, 1)
        if ctx.needs_input_grad[1]:
            gw = ops.linear_wgrad(g.reshape(O, N), x.detach().reshape(I, N)).reshape(w.shape)
        if ctx.has_b and ctx.needs_input_grad[2]:
            gb = ops.rowsum(g.reshape(O, N))
        gres = g if (ctx.has_res and ctx.needs_input_grad[3]) else None
        return gx, gw, gb, gres, None


class LayerNormC(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, pe, eps, fb):
        y = ops.layernorm_c(x.detach(), gamma.detach(), beta.detach(), pe=pe, eps=eps, folded_batch=fb)
        ctx.save_for_backward(x, gamma)
        ctx.pe, ctx.eps, ctx.fb = pe, eps, fb
        return y

    @staticmethod
    def backward(ctx, g):
        x, gamma = ctx.saved_tensors
        gx, dg, db = ops.layernorm_c_bwd(x.detach(), gamma.detach(), _c(g), pe=ctx.pe, eps=ctx.eps, folded_batch=ctx.fb,
                                         need_gx=ctx.needs_input_grad[0])
        return gx, dg, db, None, None, None


class Gelu(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.gelu(x.detach())

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return ops.gelu_bwd(x.detach(), _c(g))


class Attention(Function):
    @staticmethod
    def forward(ctx, q, k, v, heads, fb):
        ctx.save_for_backward(q, k, v)
        ctx.heads, ctx.fb = heads, fb
        return ops.attention(q.detach(), k.detach(), v.detach(), heads, folded_batch=fb)

    @staticmethod
    def backward(ctx, g):
        q, k, v = ctx.saved_tensors
        gq, gk, gv = ops.attention_bwd(q.detach(), k.detach(), v.detach(), _c(g), ctx.heads, folded_batch=ctx.fb)
        return gq, gk, gv, None, None


class Dropout(Function):
    """x * mask / (1-p); the keep-mask comes from torch's generator (as nn.Dropout's does), the multiply is ours."""

    @staticmethod
    def forward(ctx, x, p):
        mask = torch.empty_like(x).bernoulli_(1.0 - p)
        ctx.mask, ctx.k = mask, 1.0 / (1.0 - p)
        return ops.mul_scaled(x.detach(), mask, ctx.k)

    @staticmethod
    def backward(ctx, g):
        return ops.mul_scaled(_c(g), ctx.mask, ctx.k), None


class ScaleTanh(Function):
    """y = clamp(scale, 5e-3, 0.5) * tanh(u)   (Training/...5.py:313-315)."""

    LO, HI = 5e-3, 0.5

    @staticmethod
    def forward(ctx, u, scale, raw=None):
        if raw is None:                                   # one device read; callers in a loop pass the cached host value
            raw = float(scale.detach().float().item())
        s = min(max(raw, ScaleTanh.LO), ScaleTanh.HI)
        ctx.save_for_backward(u)
        ctx.s, ctx.inside = s, (ScaleTanh.LO <= raw <= ScaleTanh.HI)
        return ops.scale_tanh(u.detach(), s)

    @staticmethod
    def backward(ctx, g):
        (u,) = ctx.saved_tensors
        gu, ds = ops.scale_tanh_bwd(u.detach(), _c(g), ctx.s)
        if not ctx.inside:
            ds = torch.zeros_like(ds)
        return gu, ds, None


class RvqSte(Function):
    """ResidualVQEMA.forward with its straight-through estimator: every stage adds `residual` back with gradient 1 and
    every stage's residual is the input minus detached codes, so d out / d z = (number of books used) * I
    (Training/...5.py:262-265)."""

    @staticmethod
    def forward(ctx, z, books, n_use):
        nb = books.shape[0] if n_use is None else max(0, min(int(n_use), books.shape[0]))
        ctx.nb = nb
        return ops.rvq_ema_forward(z.detach(), books, n_use)

    @staticmethod
    def backward(ctx, g):
        return g * float(ctx.nb), None, None
