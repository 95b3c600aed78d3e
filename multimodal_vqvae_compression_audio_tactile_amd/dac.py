"""Drop-in for the ``dac.DAC`` (24 kHz) object surface the reference scripts touch, backed by libmvq_hip.so.

What the reference uses:
  ``dac.DAC.load(path).to(DEVICE).eval()`` then ``.encoder``, ``.quantizer``, ``.decoder`` pulled apart and
  passed to ``AllPredAR`` / ``ProposedEval`` (Training/compare_dacvsproposal_5.py:329-338,
  Evaluation/dac_vcpwq_proposed6_latency.py:527-535); ``encoder(x)``, ``quantizer(z)`` -> 5-tuple,
  ``decoder(z)``, ``mdl.encode(x, n_quantizers=n)``, ``mdl.decode(z)``
  (Evaluation/compare_dacvsproposal_5_eval.py:369-370).

The modules below are ``torch.nn.Module`` parameter containers with the upstream parameter names
(``block.{i}...weight_g|weight_v|bias|alpha``, ``model.{i}...``, ``quantizers.{i}.in_proj|out_proj|codebook``)
so that reference checkpoints (``A_ENC.* / A_QUANT.* / T_ENC.* / T_DEC.*``) load unchanged.  ``forward`` never
touches torch math: it walks a fused launch plan over the C ABI (weight-norm folded and packed once, Snake1d
fused into the neighbouring conv's prologue/epilogue, residual adds and tanh fused into epilogues).
Parameters are frozen by the reference (``requires_grad_(False)``, Training/...5.py:283-284).  With autograd enabled
and an input that requires a gradient (the training config, ...5.py:322,393) ``Decoder.forward`` runs a saving forward and
a HIP backward w.r.t. its input (``_DecoderInputGrad``); the saved forward is released layer by layer during that backward.
"""
from __future__ import annotations

import math
import os
from typing import List, Optional

import torch
import torch.nn as nn

from . import ops
from ._lib import MvqError

ENC_RATES = (2, 4, 5, 8)
DEC_RATES = (8, 5, 4, 2)

# A/B switches of the launch plan, read ONCE at import (never per call) and reported by bench.py as `plan_overrides`; any of them
# (or an opt-in arithmetic mode, ops.set_arith) routes Encoder / Decoder through the per-layer Python plan below instead of the
# whole-stack C entry points (mvq_encoder_fwd_f32 / mvq_decoder_fwd_f32 / mvq_decoder_bwd_input_f32), whose plan is the library's.
PLAN_ENV = ("MVQ_RU_PRESNAKED", "MVQ_VPACKED_LATENTS", "MVQ_PACKED_MIN_BATCH", "MVQ_PACKED_LATENTS", "MVQ_PY_PLAN")
PLAN_OVERRIDES = {k: os.environ[k] for k in PLAN_ENV if k in os.environ}
HOST_ENV_SEEN = {k: os.environ[k] for k in ("MVQ_TWO_STREAM_MAX_BATCH", "MVQ_AR_FUSED_MAX_BATCH", "MVQ_AR_STAGED_MAX_BATCH") if k in os.environ}      # proposed.py: stream plan of the branches
USE_STACKS = not PLAN_OVERRIDES            # tests flip this to compare the two plans


def plan_overrides():
    """Names of the launch-plan / arithmetic A/B switches seen in the environment at import (empty in a product run)."""
    return sorted(PLAN_OVERRIDES) + sorted(HOST_ENV_SEEN) + sorted(ops.ARITH_ENV_SEEN)


class _Packed:
    """Cache of derived device tensors keyed on the (identity, version, storage) of their source parameters.
    In-place edits through ``param.data`` bypass the version counter: call ``.invalidate()`` after such an edit (the
    backbones are frozen in the reference, Training/compare_dacvsproposal_5.py:283-284; load_state_dict / .to() are seen)."""

    def __init__(self):
        self._key = None
        self.value = None

    def invalidate(self):
        self._key = None

    def get(self, params, make):
        key = tuple((id(p), p._version, p.data_ptr(), str(p.device)) for p in params)
        if key != self._key:
            self.value = make()
            self._key = key
        return self.value


def _stack_of(mod, describe):
    """Cached mvq_stack of an Encoder / Decoder.  The parameter list (upstream state-dict order) is resolved once; a call only
    compares a cheap stamp -- version counters, first storage address, device -- so that load_state_dict / .to() / an in-place
    update rebuild the handle (edits through ``param.data`` bypass the counters: call ``mod._stack_state = None`` after one)."""
    st = getattr(mod, "_stack_state", None)
    if st is None or st.get("probe") is None:              # first use, or a deep copy / unpickled module (handles are not copied)
        probe = describe()
        named = dict(mod.named_parameters())
        st = mod._stack_state = {"probe": probe, "params": [named[n] for n in probe.param_names()], "stamp": None, "stack": None}
    ps = st["params"]
    stamp = (sum(p._version for p in ps), ps[0].data_ptr(), ps[-1].data_ptr(), ps[0].device)
    if stamp != st["stamp"]:
        st["stack"] = st["probe"].bind(ps)
        st["stamp"] = stamp
    return st["stack"]


class Snake1d(nn.Module):
    """Parameter holder for the upstream Snake1d (alpha [1,C,1]); applied fused inside the adjacent conv."""

    def __init__(self, channels: int):
        super().__init__()
        self.alpha = nn.Parameter(torch.ones(1, channels, 1))

    def flat(self) -> torch.Tensor:
        return self.alpha.detach().reshape(-1).float().contiguous()


class _WNKeys:
    """Accept both spellings of a weight-normed layer in a checkpoint: the old hook-based ``weight_g`` / ``weight_v``
    (torch.nn.utils.weight_norm -- what upstream DAC releases and the reference's best.pth hold) and the parametrization
    spelling ``parametrizations.weight.original0`` (= g) / ``original1`` (= v) that newer torch writes when the same model
    is re-saved through torch.nn.utils.parametrizations.weight_norm.  state_dict() always emits the old names."""

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        for new, old in (("parametrizations.weight.original0", "weight_g"), ("parametrizations.weight.original1", "weight_v")):
            if prefix + new in state_dict and prefix + old not in state_dict:
                state_dict[prefix + old] = state_dict.pop(prefix + new)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class WNConv1d(_WNKeys, nn.Module):
    """weight_norm(nn.Conv1d) parameter holder (old-style names weight_g / weight_v / bias)."""

    def __init__(self, cin, cout, kernel_size, stride=1, dilation=1, padding=0):
        super().__init__()
        self.cin, self.cout, self.ks = cin, cout, kernel_size
        self.stride, self.dilation, self.padding = stride, dilation, padding
        v = torch.randn(cout, cin, kernel_size) / math.sqrt(cin * kernel_size)
        self.weight_g = nn.Parameter(v.reshape(cout, -1).norm(dim=1).reshape(cout, 1, 1))
        self.weight_v = nn.Parameter(v)
        self.bias = nn.Parameter(torch.zeros(cout))
        self._packed = _Packed()

    def packed(self) -> torch.Tensor:
        return self._packed.get((self.weight_g, self.weight_v),
                                lambda: ops.pack_conv1d(ops.weight_norm(self.weight_v.detach(), self.weight_g.detach())))

    def folded_weight(self) -> torch.Tensor:
        return ops.weight_norm(self.weight_v.detach(), self.weight_g.detach())

    def packed_bf16x3(self) -> torch.Tensor:
        """Three-piece bf16 image of the folded 7-tap weights: the opt-in, non-parity "bf16x6" arithmetic mode only (ops.set_arith)."""
        if not hasattr(self, "_packed_bf"):
            self._packed_bf = _Packed()
        return self._packed_bf.get((self.weight_g, self.weight_v), lambda: ops.pack_conv1d_k7_bf16x3(self.folded_weight()))

    def packed_f16x2(self):
        """(two-piece fp16 image, |w| maximum) of the folded 7-tap weights: the opt-in, non-parity "f16x3" mode only."""
        if not hasattr(self, "_packed_f16"):
            self._packed_f16 = _Packed()
        return self._packed_f16.get((self.weight_g, self.weight_v), lambda: ops.pack_conv1d_k7_f16x2(self.folded_weight()))

    def packed_dgrad(self) -> torch.Tensor:
        if not hasattr(self, "_packed_dg"):
            self._packed_dg = _Packed()
        return self._packed_dg.get((self.weight_g, self.weight_v), lambda: ops.pack_conv1d_dgrad(self.folded_weight()))

    def mode_eligible(self) -> bool:
        """An opt-in arithmetic mode is set and this is a layer it claims: a square 7-tap stride-1 conv of an eligible width."""
        return self.ks == 7 and self.stride == 1 and self.cin == self.cout and ops.bf16x6_eligible(self.cin)

    def packed_mode(self):
        return self.packed_f16x2() if ops.get_arith() == "f16x3" else self.packed_bf16x3()

    def packed_mode_dgrad(self):
        """The input-gradient image of the current mode (training config: Decoder.backward_input)."""
        mode = ops.get_arith()
        if not hasattr(self, "_packed_mode_dg"):
            self._packed_mode_dg = {}
        c = self._packed_mode_dg.setdefault(mode, _Packed())
        make = (lambda: ops.pack_conv1d_k7_f16x2(self.folded_weight(), dgrad=True)) if mode == "f16x3" else \
               (lambda: ops.pack_conv1d_k7_bf16x3(self.folded_weight(), dgrad=True))
        return c.get((self.weight_g, self.weight_v), make)

    def dgrad(self, gy, tin, dsnake_src=None, dsnake_alpha=None, residual=None):
        """Gradient w.r.t. this conv's input (length tin), times the derivative of the Snake in front of it."""
        if self.mode_eligible() and tin == gy.shape[-1]:         # opt-in modes: the same matrix-core kernel on the flipped image
            return ops.conv1d_k7_mode(gy, self.packed_mode_dgrad(), self.cin, self.dilation, dsn_src=dsnake_src,
                                      dsn_alpha=dsnake_alpha, residual=residual)
        return ops.conv1d_dgrad(gy, self.packed_dgrad(), self.cin, tin, self.ks, 1, self.dilation, self.padding,
                                dsnake_src=dsnake_src, dsnake_alpha=dsnake_alpha, residual=residual)

    def run(self, x, alpha_in=None, residual=None, alpha_out=None, tanh=False, alpha_dual=None, tvalid=0):
        if (alpha_dual is not None and alpha_in is None and residual is None and alpha_out is None and not tanh
                and self.mode_eligible()):
            # opt-in modes, training config (Decoder.forward_saving): pre-snaked input, dual output (pre-activation + its Snake)
            return ops.conv1d_k7_mode(x, self.packed_mode(), self.cout, self.dilation, bias=self.bias.detach(), alpha_out=alpha_dual,
                                      tvalid=tvalid, dual=True)
        return ops.conv1d(x, self.packed(), self.cout, self.ks, bias=self.bias.detach(), stride=self.stride,
                          dil=self.dilation, pad=self.padding, alpha_in=alpha_in, residual=residual,
                          alpha_out=alpha_out, tanh=tanh, alpha_dual=alpha_dual, tvalid=tvalid)

    def run_vpacked(self, x, seg_per_row, tin_valid, tout_rows, alpha_out=None):
        """The same conv on virtually packed rows (ops.conv1d_vpacked): x[B, cin, rows] with tin_valid data columns."""
        return ops.conv1d_vpacked(x, self.packed(), self.cout, self.ks, seg_per_row, tin_valid, tout_rows, bias=self.bias.detach(),
                                  stride=self.stride, dil=self.dilation, pad=self.padding, alpha_out=alpha_out)

    def forward(self, x):
        return self.run(x)


class WNConvTranspose1d(_WNKeys, nn.Module):
    """weight_norm(nn.ConvTranspose1d) parameter holder; weight_v is [Cin, Cout, k], norm over dim 0."""

    def __init__(self, cin, cout, kernel_size, stride, padding, output_padding=0):
        super().__init__()
        if kernel_size != 2 * stride:
            raise MvqError("WNConvTranspose1d: the path covers kernel_size == 2*stride (every DAC DecoderBlock)")
        self.cin, self.cout, self.ks, self.stride, self.padding = cin, cout, kernel_size, stride, padding
        self.output_padding = int(output_padding)
        v = torch.randn(cin, cout, kernel_size) / math.sqrt(cin * 2)
        self.weight_g = nn.Parameter(v.reshape(cin, -1).norm(dim=1).reshape(cin, 1, 1))
        self.weight_v = nn.Parameter(v)
        self.bias = nn.Parameter(torch.zeros(cout))
        self._packed = _Packed()

    def packed(self) -> torch.Tensor:
        return self._packed.get((self.weight_g, self.weight_v),
                                lambda: ops.pack_conv_transpose1d(
                                    ops.weight_norm(self.weight_v.detach(), self.weight_g.detach()), self.stride))

    def packed_dgrad(self) -> torch.Tensor:
        if not hasattr(self, "_packed_dg"):
            self._packed_dg = _Packed()
        return self._packed_dg.get((self.weight_g, self.weight_v), lambda: ops.pack_conv_transpose1d_dgrad(
            ops.weight_norm(self.weight_v.detach(), self.weight_g.detach())))

    def dgrad(self, gy, tin, dsnake_src=None, dsnake_alpha=None):
        return ops.conv1d_dgrad(gy, self.packed_dgrad(), self.cin, tin, self.ks, self.stride, 1, self.padding,
                                dsnake_src=dsnake_src, dsnake_alpha=dsnake_alpha)

    def run(self, x, alpha_in=None, alpha_out=None, alpha_dual=None, tout_rows=0, tvalid=0):
        return ops.conv_transpose1d(x, self.packed(), self.cout, self.stride, self.padding, bias=self.bias.detach(),
                                    alpha_in=alpha_in, alpha_out=alpha_out, alpha_dual=alpha_dual, tout_rows=tout_rows,
                                    tvalid=tvalid, output_padding=self.output_padding)

    def forward(self, x):
        return self.run(x)


class ResidualUnit(nn.Module):
    """x + conv1(snake(conv7_dilated(snake(x)))) through mvq_residual_unit_f32: one fused launch for C <= 128,
    otherwise two launches (Snake on load / on store in the first, skip add + the NEXT layer's Snake in the
    epilogue of the second)."""

    def __init__(self, dim: int, dilation: int):
        super().__init__()
        pad = ((7 - 1) * dilation) // 2
        self.block = nn.Sequential(Snake1d(dim), WNConv1d(dim, dim, 7, dilation=dilation, padding=pad),
                                   Snake1d(dim), WNConv1d(dim, dim, 1))

    # Units with C >= 96 take their input Snake from the producer's dual output: the wide ones (C >= 192, several 128-row
    # tiles per time tile) would otherwise re-evaluate it once per row tile while staging, and the fused ones (C = 96, 128)
    # then stage their 7-tap conv with LDS-DMA (no Snake on load, no halo re-evaluation): measured +3 % at C = 128, +0.6 % at
    # C = 96.  At C = 64 the extra Snake of the dual output and the second output tensor cost more than the staging saves
    # (89.7-96.2 vs 99.2-106.9 TFLOP/s, gpurun_out/r3e), so those units keep Snake-on-load.
    # MVQ_RU_PRESNAKED=0 / =all: never / always for the fused widths (A/B runs).
    PRESNAKED_MIN_C = {"0": 129, "all": 1}.get(PLAN_OVERRIDES.get("MVQ_RU_PRESNAKED", ""), 65)

    def wants_presnaked(self) -> bool:
        return self.block[1].cin >= self.PRESNAKED_MIN_C

    def run(self, x, alpha_next=None, x_snaked=None, alpha_dual=None, tvalid=0):
        c7, c1 = self.block[1], self.block[3]
        # opt-in "bf16x6" mode (ops.set_arith; NOT bit-identical to the oracle): the 7-tap conv of a wide unit on the bf16 matrix cores
        w7q = None
        if x_snaked is not None and ops.bf16x6_eligible(c7.cin):
            w7q = c7.packed_mode()
        return ops.residual_unit(x, c7.packed(), c7.bias.detach(), self.block[0].flat(), self.block[2].flat(),
                                 c1.packed(), c1.bias.detach(), c7.dilation, alpha_next=alpha_next, x_snaked=x_snaked,
                                 alpha_dual=alpha_dual, tvalid=tvalid, w7q=w7q)

    def forward(self, x):
        return self.run(x)


class EncoderBlock(nn.Module):
    def __init__(self, dim: int, stride: int):
        super().__init__()
        self.block = nn.Sequential(ResidualUnit(dim // 2, 1), ResidualUnit(dim // 2, 3), ResidualUnit(dim // 2, 9),
                                   Snake1d(dim // 2),
                                   WNConv1d(dim // 2, dim, 2 * stride, stride=stride, padding=math.ceil(stride / 2)))

    def run(self, x, alpha_next=None, x_snaked=None, alpha_dual=None, vpack=None):
        """vpack = (seg_per_row, follow_pad): run the strided conv on virtually packed rows and return (y_rows, valid length);
        falls back to the plain launch (y, None) when the shape does not qualify."""
        r0, r1, r2 = self.block[0], self.block[1], self.block[2]
        wide = r0.wants_presnaked()
        if wide:
            x, xs = r0.run(x, x_snaked=x_snaked, alpha_dual=r1.block[0].flat())
            x, xs = r1.run(x, x_snaked=xs, alpha_dual=r2.block[0].flat())
            x = r2.run(x, x_snaked=xs, alpha_next=self.block[3].flat())
        else:
            x = r0.run(x)
            x = r1.run(x)
            x = r2.run(x, alpha_next=self.block[3].flat())            # Snake before the strided conv, fused
        down = self.block[4]
        if vpack is not None:
            geo = ops.vpacked_geometry(x.shape[-1], x.shape[-1], down.ks, down.stride, down.dilation, down.padding, follow_pad=vpack[1])
            if geo is not None and alpha_dual is None and down.cin % 32 == 0 and down.cout % 128 == 0:
                tout, tout_rows, _ = geo
                try:
                    return down.run_vpacked(x, vpack[0], x.shape[-1], tout_rows, alpha_out=alpha_next), tout
                except MvqError as e:              # no LDS-DMA / regular-epilogue form for this shape (e.g. MVQ_NO_DMA=1): the plain launch
                    if getattr(e, "status", None) != -2:
                        raise
            return down.run(x, alpha_out=alpha_next, alpha_dual=alpha_dual), None
        return down.run(x, alpha_out=alpha_next, alpha_dual=alpha_dual)

    def first_alpha(self):
        """alpha of the first ResidualUnit's input Snake if that unit wants it pre-applied by the producer."""
        return self.block[0].block[0].flat() if self.block[0].wants_presnaked() else None

    def forward(self, x):
        return self.run(x)


class Encoder(nn.Module):
    """x[B,1,T] -> [B,d_latent,T/prod(strides)]  (36.8 GFLOP per 1-s segment at 24 kHz)."""

    def __init__(self, d_model: int = 64, strides=ENC_RATES, d_latent: int = 1024):
        super().__init__()
        self._desc = (int(d_model), tuple(int(s) for s in strides), int(d_latent))
        layers: List[nn.Module] = [WNConv1d(1, d_model, 7, padding=3)]
        for s in strides:
            d_model *= 2
            layers.append(EncoderBlock(d_model, s))
        layers += [Snake1d(d_model), WNConv1d(d_model, d_latent, 3, padding=1)]
        self.block = nn.Sequential(*layers)
        self.enc_dim = d_model

    def stack(self):
        """The mvq_stack of this encoder (include/mvq.h): weights folded and packed once per parameter version."""
        return _stack_of(self, lambda: ops.Stack.encoder(*self._desc))

    @torch.no_grad()
    def forward(self, x):
        """A_ENC(a) / T_ENC(t) (Training/compare_dacvsproposal_5.py:294,296): ONE C call, mvq_encoder_fwd_f32."""
        if USE_STACKS and ops.get_arith() == "f32" and x.is_cuda:
            return self.stack().encoder_fwd(x)
        return self.forward_plan(x)

    @torch.no_grad()
    def forward_plan(self, x):
        """The same launch plan walked from Python over the per-layer entry points (the opt-in arithmetic modes and A/B runs)."""
        n = len(self.block)
        a1 = self.block[1].first_alpha()                            # first unit wants a pre-snaked input: dual output of the 1 -> 64 conv
        out = self.block[0].run(x, alpha_dual=a1)
        h, hs = out if a1 is not None else (out, None)
        tail = self.block[n - 1]                                     # the k3 conv behind the trailing Snake
        vpack = self._vpack(x.shape[0], tail)
        tv = None
        for i in range(1, n - 2):
            last = i == n - 3
            nxt = None if last else self.block[i + 1].first_alpha()   # next block's first unit wants a pre-snaked input?
            if last and vpack is not None:
                h, tv = self.block[i].run(h, alpha_next=self.block[n - 2].flat(), x_snaked=hs, vpack=vpack)
                hs = None
                continue
            out = self.block[i].run(h, alpha_next=self.block[n - 2].flat() if last else None, x_snaked=hs, alpha_dual=nxt)
            h, hs = out if nxt is not None else (out, None)
        if tv is not None:
            # latent-rate layers on virtually packed rows: h[B, C, rows] carries tv valid columns + a zero tail; the k3 conv maps
            # rows -> rows (its padding is that zero tail), then the tail is cut off
            y = tail.run_vpacked(h, vpack[0], tv, h.shape[-1])
            return y[..., :tv].contiguous()
        return tail.run(h)

    # Throughput batches: the last strided conv (T 600 -> 75) and the k3 conv run on VIRTUALLY packed rows (include/mvq.h):
    # VPACK_SEG items side by side in one GEMM row -- 10 x 76 = 760 columns = six 128-column tiles, 98 % live (unpacked: one
    # 128 x 96 tile per item, 78 % live) -- with no repacked copy in memory.  MVQ_VPACKED_LATENTS=0 switches it off (A/B runs).
    VPACK_SEG = 10
    VPACK_MIN_BATCH = 32
    VPACKED = PLAN_OVERRIDES.get("MVQ_VPACKED_LATENTS", "1") != "0"

    def _vpack(self, batch, tail):
        if not self.VPACKED or batch < self.VPACK_MIN_BATCH:
            return None
        if tail.stride != 1 or tail.dilation != 1 or 2 * tail.padding != tail.ks - 1 or tail.cin % 32 or tail.cout % 128:
            return None
        return (self.VPACK_SEG, tail.padding)


class DecoderBlock(nn.Module):
    def __init__(self, input_dim: int, output_dim: int, stride: int, output_padding: bool = False):
        super().__init__()
        self.block = nn.Sequential(Snake1d(input_dim),
                                   WNConvTranspose1d(input_dim, output_dim, 2 * stride, stride, math.ceil(stride / 2),
                                                     output_padding=(stride % 2) if output_padding else 0),
                                   ResidualUnit(output_dim, 1), ResidualUnit(output_dim, 3), ResidualUnit(output_dim, 9))

    def run(self, x, alpha_next=None, pre_snaked=False, t_in=None):
        """`pre_snaked`: x already carries this block's leading Snake (fused into its producer's epilogue).
        `t_in`: true length of x's rows when they carry a zero tail (zero-padded rows, include/mvq.h); with it the call
        returns (y, true length of y) and y's rows are themselves padded to a multiple of 4 where the kernels allow."""
        up, r0, r1, r2 = self.block[1], self.block[2], self.block[3], self.block[4]
        a_in = None if pre_snaked else self.block[0].flat()
        rows, tv = 0, 0
        if t_in is not None:
            s_, p_, op_ = up.stride, up.padding, up.output_padding
            tn = (t_in - 1) * s_ - 2 * p_ + 2 * s_ + op_                 # true output length
            tnat = (x.shape[-1] - 1) * s_ - 2 * p_ + 2 * s_ + op_        # what the (possibly padded) input rows would give
            tp = (tn + 3) // 4 * 4
            can_pad = tp != tn and tp <= tnat - op_ + p_ and up.cin % 32 == 0 and up.cout % 32 == 0
            rows = tp if can_pad else tn
            tv = tn if can_pad else 0
            rows = 0 if (rows == tnat and tv == 0) else rows
        if r0.wants_presnaked():
            h, hs = up.run(x, alpha_in=a_in, alpha_dual=r0.block[0].flat(), tout_rows=rows, tvalid=tv)
            h, hs = r0.run(h, x_snaked=hs, alpha_dual=r1.block[0].flat(), tvalid=tv)
            h, hs = r1.run(h, x_snaked=hs, alpha_dual=r2.block[0].flat(), tvalid=tv)
            y = r2.run(h, x_snaked=hs, alpha_next=alpha_next, tvalid=tv)
        else:
            h = up.run(x, alpha_in=a_in, tout_rows=rows, tvalid=tv)
            h = r0.run(h, tvalid=tv)
            h = r1.run(h, tvalid=tv)
            y = r2.run(h, alpha_next=alpha_next, tvalid=tv)
        return y if t_in is None else (y, tn)

    def forward(self, x):
        return self.run(x)


class Decoder(nn.Module):
    """z[B,C,Tl] -> [B,1,~Tl*prod(rates)]  (83.4 GFLOP per segment)."""

    def __init__(self, input_channel: int = 1024, channels: int = 1536, rates=DEC_RATES, d_out: int = 1,
                 output_padding: bool = False):
        """output_padding=True: every DecoderBlock's ConvTranspose1d gets ``output_padding = stride % 2`` (the variant believed
        to be upstream's repository head: 75 tokens -> 24 000 samples).  Default = the 1.0.0 release (-> 23 992), which is what
        every fixture in tests/golden assumes."""
        super().__init__()
        layers: List[nn.Module] = [WNConv1d(input_channel, channels, 7, padding=3)]
        out = channels
        for i, s in enumerate(rates):
            inp, out = channels // 2 ** i, channels // 2 ** (i + 1)
            layers.append(DecoderBlock(inp, out, s, output_padding))
        layers += [Snake1d(out), WNConv1d(out, d_out, 7, padding=3), nn.Tanh()]
        self.model = nn.Sequential(*layers)
        self._desc = (int(input_channel), int(channels), tuple(int(r) for r in rates), int(d_out), bool(output_padding))

    def stack(self):
        """The mvq_stack of this decoder (forward + input-gradient images), made once per parameter version."""
        return _stack_of(self, lambda: ops.Stack.decoder(*self._desc))

    def _stacked(self, z) -> bool:
        return USE_STACKS and ops.get_arith() == "f32" and z.is_cuda and z.shape[0] > 0 and z.shape[-1] > 0

    # ---- training config: gradient w.r.t. the input, weights frozen ---------------
    @torch.no_grad()
    def forward_saving(self, z):
        """Same arithmetic as forward(), but every Snake input (and the final tanh output) is kept for the backward."""
        if self._stacked(z):                                       # ONE C call: mvq_decoder_fwd_saving_f32
            y, blob = self.stack().decoder_fwd_saving(z)
            return y, {"blob": blob, "batch": z.shape[0], "z_len": z.shape[-1]}
        m = self.model
        nblk = len(m) - 4
        saved = {"z_len": z.shape[-1]}
        # every producer emits (raw, Snake for the consumer): no conv evaluates Snake while staging
        h, hs = m[0].run(z, alpha_dual=m[1].block[0].flat())
        for i in range(1, nblk + 1):
            blk = m[i].block
            saved[f"b{i}.x"] = h
            h, hs = blk[1].run(hs, alpha_dual=blk[2].block[0].flat())
            for j in (2, 3, 4):
                ru = blk[j].block
                t7, t7s = ru[1].run(hs, alpha_dual=ru[2].flat())
                saved[f"b{i}.r{j}.x"], saved[f"b{i}.r{j}.t7"] = h, t7
                nxt = blk[j + 1].block[0].flat() if j < 4 else (m[i + 1].block[0].flat() if i < nblk else m[nblk + 1].flat())
                h, hs = ru[3].run(t7s, residual=h, alpha_dual=nxt)
        saved["hl"] = h
        y = m[nblk + 2].run(hs, tanh=True)
        saved["y"] = y
        return y, saved

    @torch.no_grad()
    def backward_input(self, saved, gy):
        """dL/dz from dL/dy: the same MFMA conv kernels on flipped / transposed weight images, Snake and tanh
        derivatives fused into their epilogues (83 GFLOP per segment, like the forward)."""
        if "blob" in saved:                                        # ONE C call: mvq_decoder_bwd_input_f32
            return self.stack().decoder_bwd_input(saved.pop("blob"), gy, saved["batch"], saved["z_len"])
        m = self.model
        nblk = len(m) - 4
        # every saved activation is released as soon as its layer's gradient has been queued (pop, not index): the backward's
        # footprint shrinks as it walks down the stack instead of holding the whole forward until the step ends
        y = saved.pop("y")
        g = ops.mul_dtanh(gy, y)
        del y
        hl = saved.pop("hl")
        g = m[nblk + 2].dgrad(g, hl.shape[-1], dsnake_src=hl, dsnake_alpha=m[nblk + 1].flat())
        del hl
        for i in range(nblk, 0, -1):
            blk = m[i].block
            for j in (4, 3, 2):
                ru = blk[j].block
                x, t7 = saved.pop(f"b{i}.r{j}.x"), saved.pop(f"b{i}.r{j}.t7")
                g1 = ru[3].dgrad(g, t7.shape[-1], dsnake_src=t7, dsnake_alpha=ru[2].flat())
                g = ru[1].dgrad(g1, x.shape[-1], dsnake_src=x, dsnake_alpha=ru[0].flat(), residual=g)
                del x, t7, g1
            xin = saved.pop(f"b{i}.x")
            g = blk[1].dgrad(g, xin.shape[-1], dsnake_src=xin, dsnake_alpha=blk[0].flat())
            del xin
        return m[0].dgrad(g, saved["z_len"])

    def forward(self, z):
        """Inference: the fused fast path.  With autograd enabled and z.requires_grad (the reference's training step,
        Training/compare_dacvsproposal_5.py:322,393): saving forward + HIP backward w.r.t. z; weights stay frozen."""
        if torch.is_grad_enabled() and z.requires_grad:
            return _DecoderInputGrad.apply(z, self)
        return self._forward_fast(z)

    @torch.no_grad()
    def _forward_fast(self, z):
        """T_DEC(z) (Training/compare_dacvsproposal_5.py:322): ONE C call, mvq_decoder_fwd_f32."""
        if self._stacked(z):
            return self.stack().decoder_fwd(z)
        if z.shape[-1] == 0:                                       # a clip shorter than a token: nothing to decode
            return z.new_zeros(z.shape[0], self._desc[3], 0)
        return self.forward_plan(z)

    @torch.no_grad()
    def forward_plan(self, z):
        """The same launch plan walked from Python over the per-layer entry points (the opt-in arithmetic modes and A/B runs)."""
        m = self.model
        nblk = len(m) - 4
        # Rows whose length is not a multiple of 4 (75 latent frames, the 2 999-sample block) are carried zero-padded to the
        # next multiple of 4 so that every layer runs its 16-byte load / store paths (include/mvq.h "Zero-padded rows").
        t = z.shape[-1]
        tv = 0
        if self._use_packed_latents(z):
            return self._forward_fast_packed(z)
        if t % 4 and t > 0 and m[0].cin % 32 == 0:
            zp = torch.zeros(z.shape[0], z.shape[1], (t + 3) // 4 * 4, device=z.device, dtype=torch.float32)
            zp[..., :t] = z
            z, tv = zp, t
        h = m[0].run(z, alpha_out=m[1].block[0].flat(), tvalid=tv)
        for i in range(1, nblk + 1):
            nxt = m[i + 1].block[0].flat() if i < nblk else m[nblk + 1].flat()
            h, t = m[i].run(h, alpha_next=nxt, pre_snaked=True, t_in=t)
        y = m[nblk + 2].run(h, tanh=True)
        return y if y.shape[-1] == t else y[..., :t].contiguous()


    # ---- packed latent-rate layers (include/mvq.h "PACKED latent-rate rows") ------------------------------
    # Throughput batches only: model.0 (k7 conv) and the first block's ConvTranspose1d run on rows of PACK_SEG segments at a period
    # of ceil((T + 3) / 4) * 4 columns (T = 75 -> 80: 8 x 80 = 640 = five full 128-column tiles, 94 % live columns instead of
    # 78 %); the transposed conv writes the ordinary unpacked [B, C/2, 8T] tensor, everything behind it is unchanged.
    # MVQ_PACKED_LATENTS=0 switches it off (A/B runs).
    PACK_SEG = 8
    PACK_MIN_BATCH = int(PLAN_OVERRIDES.get("MVQ_PACKED_MIN_BATCH", "32"))
    PACKED = PLAN_OVERRIDES.get("MVQ_PACKED_LATENTS", "1") != "0"

    def _use_packed_latents(self, z) -> bool:
        m = self.model
        up = m[1].block[1]
        t = z.shape[-1]
        return (self.PACKED and z.shape[0] >= self.PACK_MIN_BATCH and 0 < t <= 1024 and m[0].cin % 32 == 0 and m[0].cout % 128 == 0
                and up.output_padding == 0 and up.cin % 32 == 0 and (up.cout * up.stride) % 128 == 0
                and ((t - 1) * up.stride - 2 * up.padding + 2 * up.stride) % 4 == 0)

    @torch.no_grad()
    def _forward_fast_packed(self, z):
        m = self.model
        nblk = len(m) - 4
        B, _, t = z.shape
        per = (t + 3 + 3) // 4 * 4                      # >= 3 zero columns between segments: the k7 conv's padding
        zp = ops.pack_segments(z, self.PACK_SEG, per)
        c0, blk = m[0], m[1]
        up, r0, r1, r2 = blk.block[1], blk.block[2], blk.block[3], blk.block[4]
        hp = ops.conv1d_packed_rows(zp, c0.packed(), c0.cout, c0.ks, self.PACK_SEG, per, t, bias=c0.bias.detach(),
                                    dil=c0.dilation, pad=c0.padding, alpha_out=blk.block[0].flat())
        pre = r0.wants_presnaked()
        out = ops.conv_transpose1d_packed_rows(hp, up.packed(), up.cout, up.stride, up.padding, self.PACK_SEG, per, t, B,
                                               bias=up.bias.detach(), alpha_dual=r0.block[0].flat() if pre else None)
        nxt = m[2].block[0].flat() if nblk > 1 else m[nblk + 1].flat()
        if pre:
            h, hs = out
            h, hs = r0.run(h, x_snaked=hs, alpha_dual=r1.block[0].flat())
            h, hs = r1.run(h, x_snaked=hs, alpha_dual=r2.block[0].flat())
            h = r2.run(h, x_snaked=hs, alpha_next=nxt)
        else:
            h = r0.run(out)
            h = r1.run(h)
            h = r2.run(h, alpha_next=nxt)
        t = h.shape[-1]
        for i in range(2, nblk + 1):
            nxt = m[i + 1].block[0].flat() if i < nblk else m[nblk + 1].flat()
            h, t = m[i].run(h, alpha_next=nxt, pre_snaked=True, t_in=t)
        y = m[nblk + 2].run(h, tanh=True)
        return y if y.shape[-1] == t else y[..., :t].contiguous()


class _DecoderInputGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, dec):
        y, saved = dec.forward_saving(z.detach())
        # `saved` must not hold the tensor object this Function RETURNS: the returned y gets this node as its grad_fn, and
        # node -> ctx.saved -> y -> grad_fn -> node is a reference cycle only Python's cyclic collector can break -- the whole
        # saved forward (45 GB at 256 segments) then outlives the step until a full collection happens to run (measured:
        # +45 GB per step, 145 GB peak after three steps).  A detached alias shares the storage without the back pointer.
        if "blob" not in saved:
            saved["y"] = y.detach()
        ctx.dec, ctx.saved = dec, saved
        return y

    @staticmethod
    def backward(ctx, gy):
        saved, ctx.saved = ctx.saved, None                       # consumed once: nothing survives the backward
        if saved is None:
            raise RuntimeError("Decoder input-gradient: backward called twice (the saved forward is released by the first call)")
        return ctx.dec.backward_input(saved, gy.contiguous()), None


class VectorQuantize(nn.Module):
    def __init__(self, input_dim: int, codebook_size: int, codebook_dim: int):
        super().__init__()
        self.in_proj = WNConv1d(input_dim, codebook_dim, 1)
        self.out_proj = WNConv1d(codebook_dim, input_dim, 1)
        self.codebook = nn.Embedding(codebook_size, codebook_dim)


class ResidualVectorQuantize(nn.Module):
    """DAC residual VQ: all stages in one fused launch (mvq_dac_rvq_items_f32).  In ``train()`` mode with
    ``quantizer_dropout > 0`` upstream draws, for the first int(B*quantizer_dropout) items, a random number of stages in
    [1, n_codebooks] (torch.randint on the CPU generator, reproduced here call for call) and masks the later stages out of
    z_q; all stages still run.  Losses are not produced (the reference discards them, `qa, *_ =`)."""

    def __init__(self, input_dim: int = 1024, n_codebooks: int = 32, codebook_size: int = 1024,
                 codebook_dim: int = 8, quantizer_dropout: float = 0.0):
        super().__init__()
        self.n_codebooks, self.codebook_size, self.codebook_dim = n_codebooks, codebook_size, codebook_dim
        self.n_q, self.bins = n_codebooks, codebook_size       # probed by get_n_books_and_bins (…5_eval.py:233-246)
        self.quantizer_dropout = float(quantizer_dropout)
        self.quantizers = nn.ModuleList([VectorQuantize(input_dim, codebook_size, codebook_dim)
                                         for _ in range(n_codebooks)])
        self._stacked = _Packed()

    def _weights(self):
        qs = self.quantizers

        def make():
            in_w = torch.stack([q.in_proj.folded_weight().reshape(self.codebook_dim, -1) for q in qs]).contiguous()
            out_w = torch.stack([q.out_proj.folded_weight().reshape(-1, self.codebook_dim) for q in qs]).contiguous()
            in_b = torch.stack([q.in_proj.bias.detach() for q in qs]).contiguous()
            out_b = torch.stack([q.out_proj.bias.detach() for q in qs]).contiguous()
            cb = torch.stack([q.codebook.weight.detach() for q in qs]).contiguous()
            prep = ops.dac_rvq_prepare(cb) if cb.is_cuda else None          # normalised codebooks: once per weight version
            return in_w, in_b, cb, out_w, out_b, prep

        params = [p for q in qs for p in (q.in_proj.weight_g, q.in_proj.weight_v, q.in_proj.bias,
                                          q.out_proj.weight_g, q.out_proj.weight_v, q.out_proj.bias,
                                          q.codebook.weight)]
        return self._stacked.get(params, make)

    @torch.no_grad()
    def forward(self, z, n_quantizers: Optional[int] = None):
        nq = self.n_codebooks if n_quantizers is None else max(1, min(int(n_quantizers), self.n_codebooks))
        in_w, in_b, cb, out_w, out_b, prep = self._weights()
        nq_item = None
        if self.training:                                        # upstream ignores n_quantizers here and runs every stage
            nq = self.n_codebooks
            B = z.shape[0]
            lim = torch.ones((B,)) * self.n_codebooks + 1
            dropout = torch.randint(1, self.n_codebooks + 1, (B,))
            n_dropout = int(B * self.quantizer_dropout)
            lim[:n_dropout] = dropout[:n_dropout]
            if n_dropout > 0:
                nq_item = lim.to(torch.int32).to(z.device)
        zq, codes, latents = ops.dac_rvq(z, in_w, in_b, cb, out_w, out_b, nq, nq_item=nq_item, prepared=prep)
        zero = torch.zeros((), device=zq.device)
        return zq, codes, latents, zero, zero.clone()


class DAC(nn.Module):
    """encoder / quantizer / decoder with the 24 kHz hyper-parameters; ``encode`` / ``decode`` as upstream."""

    def __init__(self, encoder_dim=64, encoder_rates=ENC_RATES, latent_dim=None, decoder_dim=1536,
                 decoder_rates=DEC_RATES, n_codebooks=32, codebook_size=1024, codebook_dim=8, quantizer_dropout=0.0,
                 sample_rate=24000, decoder_output_padding=False):
        super().__init__()
        if latent_dim is None:
            latent_dim = encoder_dim * (2 ** len(encoder_rates))
        self.sample_rate = sample_rate
        self.hop_length = int(math.prod(encoder_rates))
        self.encoder = Encoder(encoder_dim, encoder_rates, latent_dim)
        self.quantizer = ResidualVectorQuantize(latent_dim, n_codebooks, codebook_size, codebook_dim, float(quantizer_dropout))
        self.decoder = Decoder(latent_dim, decoder_dim, decoder_rates, output_padding=bool(decoder_output_padding))

    @classmethod
    def load(cls, path, strict: bool = True, **kw):
        """Load a LOCAL checkpoint (there is no downloader: the reference's dac.utils.download is a network fetch).
        Accepted layouts: upstream's ``{"state_dict": ..., "metadata": {"kwargs": {...}}}`` (audiotools BaseModel.save --
        the constructor arguments stored there are honoured, explicit ``**kw`` override them, as upstream's
        BaseModel.load does; this is what sets ``quantizer_dropout``, which matters under ``net.train()``), a bare
        ``{"state_dict": ...}``, or a plain state dict.  Either weight-norm key spelling loads (see _WNKeys)."""
        import inspect
        obj = torch.load(str(path), map_location="cpu")
        sd = obj.get("state_dict", obj) if isinstance(obj, dict) else obj
        kwargs = {}
        meta = obj.get("metadata") if isinstance(obj, dict) else None
        if isinstance(meta, dict) and isinstance(meta.get("kwargs"), dict):
            accepted = set(inspect.signature(cls.__init__).parameters) - {"self"}
            kwargs = {k: v for k, v in meta["kwargs"].items() if k in accepted}
        kwargs.update(kw)
        model = cls(**kwargs)
        model.load_state_dict(sd, strict=strict)
        model.metadata = meta
        return model

    @torch.no_grad()
    def encode(self, audio_data, n_quantizers: Optional[int] = None):
        z = self.encoder(audio_data)
        return self.quantizer(z, n_quantizers)

    @torch.no_grad()
    def decode(self, z):
        return self.decoder(z)

    @torch.no_grad()
    def forward(self, audio_data, sample_rate=None, n_quantizers=None):
        length = audio_data.shape[-1]
        pad = (-length) % self.hop_length
        x = torch.nn.functional.pad(audio_data, (0, pad))
        z, codes, latents, cl, cbl = self.encode(x, n_quantizers)
        y = self.decode(z)
        return {"audio": y[..., :length], "z": z, "codes": codes, "latents": latents,
                "vq/commitment_loss": cl, "vq/codebook_loss": cbl}
