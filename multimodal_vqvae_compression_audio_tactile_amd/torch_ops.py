"""``torch.ops.mi355x_vqvae.*`` -- the operator-level face of the C ABI (last row).

The reference has no operator interface of its own (its boundary is the module surface mirrored in ``dac.py`` / ``proposed.py``);
this registers the hot-path entry points of ``include/mvq.h`` as ``torch.library`` custom operators so that a caller can reach them
as ``torch.ops.mi355x_vqvae.<name>`` and so that ``torch.compile`` / ``torch.export`` can trace THROUGH a call (every operator has a
shape-only fake implementation; the real one is the HIP kernel behind ``ops.py`` -- there is still no CPU path: a CPU tensor raises).

  conv1d_snake_f32            mvq_conv1d_f32             upstream dac WNConv1d (+ Snake1d / skip fused), ...5.py:294-296,322
  conv_transpose1d_snake_f32  mvq_conv_transpose1d_f32   upstream dac WNConvTranspose1d (DecoderBlock), ...5.py:322
  residual_unit_f32           mvq_residual_unit_f32      upstream dac ResidualUnit
  vq_rvq_search_f32           mvq_rvq_ema_forward_f32    ResidualVQEMA.forward, Training/compare_dacvsproposal_5.py:253-265
  vq_cosine_rvq_f32           mvq_dac_rvq_f32            upstream dac ResidualVectorQuantize.forward, ...5.py:295
  ema_update_f32              mvq_rvq_ema_step_f32       ResidualVQEMA.ema_step, ...5.py:266-277 (mutates `books`)
  encoder_fwd                 mvq_encoder_fwd_f32        A_ENC(a) / T_ENC(t), ...5.py:294,296 (whole stack: `stack` = ops.Stack(...).id)
  decoder_fwd                 mvq_decoder_fwd_f32        T_DEC(z), ...5.py:322
  decoder_bwd_input           mvq_decoder_fwd_saving_f32 + mvq_decoder_bwd_input_f32: dL/dz of T_DEC, ...5.py:393

Weights are the PACKED images of ``ops.pack_conv1d`` / ``ops.pack_conv_transpose1d`` (made once per weight load).  Import this module
to register (``import multimodal_vqvae_compression_audio_tactile_amd.torch_ops``); the package does not import it by itself.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from . import ops

NS = "mi355x_vqvae"


@torch.library.custom_op(f"{NS}::conv1d_snake_f32", mutates_args=())
def conv1d_snake_f32(x: Tensor, wp: Tensor, bias: Optional[Tensor], alpha_in: Optional[Tensor], residual: Optional[Tensor],
                     alpha_out: Optional[Tensor], cout: int, ks: int, stride: int, pad: int, dil: int) -> Tensor:
    return ops.conv1d(x, wp, cout, ks, bias=bias, stride=stride, dil=dil, pad=pad, alpha_in=alpha_in, residual=residual,
                      alpha_out=alpha_out)


@conv1d_snake_f32.register_fake
def _(x, wp, bias, alpha_in, residual, alpha_out, cout, ks, stride, pad, dil):
    return x.new_empty(x.shape[0], cout, ops.conv1d_out_len(x.shape[-1], ks, stride, dil, pad))


@torch.library.custom_op(f"{NS}::conv_transpose1d_snake_f32", mutates_args=())
def conv_transpose1d_snake_f32(x: Tensor, wp: Tensor, bias: Optional[Tensor], alpha_in: Optional[Tensor], alpha_out: Optional[Tensor],
                               cout: int, stride: int, pad: int) -> Tensor:
    return ops.conv_transpose1d(x, wp, cout, stride, pad, bias=bias, alpha_in=alpha_in, alpha_out=alpha_out)


@conv_transpose1d_snake_f32.register_fake
def _(x, wp, bias, alpha_in, alpha_out, cout, stride, pad):
    return x.new_empty(x.shape[0], cout, max((x.shape[-1] - 1) * stride - 2 * pad + 2 * stride, 0))


@torch.library.custom_op(f"{NS}::residual_unit_f32", mutates_args=())
def residual_unit_f32(x: Tensor, w7p: Tensor, b7: Tensor, alpha_a: Tensor, alpha_b: Tensor, w1p: Tensor, b1: Tensor, dil: int,
                      alpha_next: Optional[Tensor]) -> Tensor:
    return ops.residual_unit(x, w7p, b7, alpha_a, alpha_b, w1p, b1, dil, alpha_next=alpha_next)


@residual_unit_f32.register_fake
def _(x, w7p, b7, alpha_a, alpha_b, w1p, b1, dil, alpha_next):
    return torch.empty_like(x)


@torch.library.custom_op(f"{NS}::vq_rvq_search_f32", mutates_args=())
def vq_rvq_search_f32(z: Tensor, books: Tensor, n_use: int) -> Tuple[Tensor, Tensor]:
    q, idx = ops.rvq_ema_forward(z, books, n_books_use=n_use, return_indices=True)
    return q, idx


@vq_rvq_search_f32.register_fake
def _(z, books, n_use):
    nb = max(0, min(int(n_use), books.shape[0]))
    return torch.empty_like(z), z.new_empty(nb, z.shape[0] * z.shape[-1], dtype=torch.int64)


@torch.library.custom_op(f"{NS}::vq_cosine_rvq_f32", mutates_args=())
def vq_cosine_rvq_f32(z: Tensor, in_w: Tensor, in_b: Tensor, codebook: Tensor, out_w: Tensor, out_b: Tensor,
                      n_q: int) -> Tuple[Tensor, Tensor, Tensor]:
    zq, codes, lat = ops.dac_rvq(z, in_w, in_b, codebook, out_w, out_b, n_q)
    return zq, codes, lat


@vq_cosine_rvq_f32.register_fake
def _(z, in_w, in_b, codebook, out_w, out_b, n_q):
    B, _, T = z.shape
    return torch.empty_like(z), z.new_empty(B, n_q, T, dtype=torch.int64), z.new_empty(B, n_q * codebook.shape[-1], T)


@torch.library.custom_op(f"{NS}::ema_update_f32", mutates_args=("books",))
def ema_update_f32(books: Tensor, z_tokens: Tensor, decay: float) -> None:
    ops.rvq_ema_step_(z_tokens, books, decay)


@ema_update_f32.register_fake
def _(books, z_tokens, decay):
    return None


@torch.library.custom_op(f"{NS}::encoder_fwd", mutates_args=())
def encoder_fwd(x: Tensor, stack: int) -> Tensor:
    return ops.Stack.by_id(stack).encoder_fwd(x)


@encoder_fwd.register_fake
def _(x, stack):
    st = ops.Stack.by_id(stack)
    return x.new_empty(x.shape[0], st.desc.d_latent, max(st.out_len(x.shape[-1]), 0))


@torch.library.custom_op(f"{NS}::decoder_fwd", mutates_args=())
def decoder_fwd(z: Tensor, stack: int) -> Tensor:
    return ops.Stack.by_id(stack).decoder_fwd(z)


@decoder_fwd.register_fake
def _(z, stack):
    st = ops.Stack.by_id(stack)
    return z.new_empty(z.shape[0], st.desc.d_out, max(st.out_len(z.shape[-1]), 0))


@torch.library.custom_op(f"{NS}::decoder_bwd_input", mutates_args=())
def decoder_bwd_input(z: Tensor, gy: Tensor, stack: int) -> Tensor:
    """dL/dz of y = T_DEC(z) for a given dL/dy: saving forward + input-gradient backward (weights frozen)."""
    st = ops.Stack.by_id(stack)
    _, saved = st.decoder_fwd_saving(z)
    return st.decoder_bwd_input(saved, gy, z.shape[0], z.shape[-1])


@decoder_bwd_input.register_fake
def _(z, gy, stack):
    return torch.empty_like(z)


REGISTERED = ("encoder_fwd", "decoder_fwd", "decoder_bwd_input", "conv1d_snake_f32", "conv_transpose1d_snake_f32", "residual_unit_f32", "vq_rvq_search_f32", "vq_cosine_rvq_f32",
              "ema_update_f32")
