"""ctypes binding of libmvq_hip.so (the C ABI in include/mvq.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``csrc/Makefile``.  There is NO CPU
fallback: if the shared object is missing or a tensor is not on a HIP device, every op raises.
"""
from __future__ import annotations

import ctypes
import subprocess
from ctypes import c_char_p, c_float, c_int, c_size_t, c_void_p
from pathlib import Path

_PKG = Path(__file__).resolve().parent
import os as _os
# MVQ_LIB_PATH: measurement aid only (A/B runs of an alternative build of the SAME sources, tools/conv_microbench.py);
# it names another libmvq_hip build, never a fallback implementation.
SO_PATH = Path(_os.environ["MVQ_LIB_PATH"]).resolve() if _os.environ.get("MVQ_LIB_PATH") else _PKG / "libmvq_hip.so"
_lib = None

class EncoderDesc(ctypes.Structure):
    """mvq_encoder_desc (include/mvq.h)."""
    _fields_ = [("d_model", c_int), ("n_strides", c_int), ("strides", c_int * 8), ("d_latent", c_int)]


class DecoderDesc(ctypes.Structure):
    """mvq_decoder_desc (include/mvq.h)."""
    _fields_ = [("input_channel", c_int), ("channels", c_int), ("n_rates", c_int), ("rates", c_int * 8), ("d_out", c_int), ("output_padding", c_int)]


class ArArgs(ctypes.Structure):
    """mvq_ar_args (include/mvq.h)."""
    _fields_ = ([(n, c_int) for n in ("batch", "t_lat", "t_audio", "tactile_only", "books_use", "rvq_k", "c_lat", "c_ff", "code_dim", "heads", "chunk")]
                + [(n, c_float) for n in ("ln_eps", "tok_eps", "scale")]
                + [(n, c_void_p) for n in ("zt", "k_all", "v_all", "pe", "lnq_g", "lnq_b", "wq", "wo", "lnf_g", "lnf_b", "w1", "b1", "w3", "b3",
                                           "tok_g", "tok_b", "wd", "bd", "wu", "bu", "books", "z_run", "r_tokens", "idx_out")])


class ProfileEntry(ctypes.Structure):
    """mvq_profile_entry (include/mvq.h)."""
    _fields_ = [("kernel", ctypes.c_char * 96), ("seconds", ctypes.c_double), ("flops", ctypes.c_double), ("launches", c_int)]


EXPORTS = {
    # name: (restype, argtypes)
    "mvq_profile_begin": (c_int, []),
    "mvq_profile_end": (c_int, [ctypes.POINTER(ProfileEntry), c_int, ctypes.POINTER(c_int)]),
    "mvq_profile_end2": (c_int, [ctypes.POINTER(ProfileEntry), c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "mvq_profile_reserve": (c_int, [c_int]),
    "mvq_abi_version": (c_int, []),
    "mvq_build_flags": (ctypes.c_uint, []),
    "mvq_last_error": (c_char_p, []),
    "mvq_device_query": (c_int, [ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_char_p, c_int]),
    "mvq_weight_norm_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "mvq_conv1d_packed_floats": (c_size_t, [c_int, c_int, c_int]),
    "mvq_conv1d_pack_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mvq_conv_transpose1d_packed_floats": (c_size_t, [c_int, c_int, c_int]),
    "mvq_conv_transpose1d_pack_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mvq_conv_kernel_name": (c_int, [c_int] * 8 + [c_char_p, c_int]),
    "mvq_conv1d_f32": (c_int, [c_void_p] * 7 + [c_int] * 9 + [c_void_p]),
    "mvq_bf16x3_split_bytes": (c_size_t, [c_int] * 3),
    "mvq_bf16x3_split_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mvq_conv1d_k7_bf16x3_packed_bytes": (c_size_t, [c_int] * 2),
    "mvq_conv1d_k7_pack_bf16x3": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mvq_conv1d_k7_bf16x6_f32": (c_int, [c_void_p] * 9 + [c_int] * 6 + [c_void_p]),
    "mvq_f16x2_split_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mvq_conv1d_k7_f16x2_packed_bytes": (c_size_t, [c_int] * 2),
    "mvq_conv1d_k7_pack_f16x2": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mvq_conv1d_k7_f16x3_f32": (c_int, [c_void_p] * 11 + [c_int] * 6 + [c_void_p]),
    "mvq_residual_unit_scratch_floats": (c_size_t, [c_int] * 4),
    "mvq_residual_unit_kernel_name": (c_int, [c_int, c_int, c_char_p, c_int]),
    "mvq_residual_unit_f32": (c_int, [c_void_p] * 10 + [c_int] * 4 + [c_void_p]),
    "mvq_conv1d_dual_f32": (c_int, [c_void_p] * 9 + [c_int] * 9 + [c_void_p]),
    "mvq_conv_transpose1d_dual_f32": (c_int, [c_void_p] * 8 + [c_int] * 6 + [c_void_p]),
    "mvq_residual_unit_dual_f32": (c_int, [c_void_p] * 13 + [c_int] * 4 + [c_void_p]),
    "mvq_conv_transpose1d_f32": (c_int, [c_void_p] * 6 + [c_int] * 6 + [c_void_p]),
    "mvq_rvq_ema_forward_f32": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_void_p]),
    "mvq_rvq_ema_step_scratch_bytes": (c_size_t, [c_int] * 5),
    "mvq_rvq_ema_step_f32": (c_int, [c_void_p] * 3 + [c_int] * 5 + [c_float, c_void_p]),
    "mvq_dac_rvq_f32": (c_int, [c_void_p] * 9 + [c_int] * 6 + [c_void_p]),
    "mvq_dac_rvq_items_f32": (c_int, [c_void_p] * 10 + [c_int] * 6 + [c_void_p]),
    "mvq_dac_rvq_prepare_f32": (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_void_p]),
    "mvq_dac_rvq_prepared_f32": (c_int, [c_void_p] * 12 + [c_int] * 6 + [c_void_p]),
    "mvq_layernorm_c_f32": (c_int, [c_void_p] * 5 + [c_int] * 3 + [c_size_t] * 2 + [c_float, c_int, c_float, c_void_p]),
    "mvq_attention_f32": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_size_t] * 4 + [c_void_p]),
    "mvq_align_xcorr_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mvq_align_xcorr_batch_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mvq_resample_ragged_f32": (c_int, [c_void_p] * 6 + [c_int] * 7 + [c_void_p]),
    "mvq_conv1d_dgrad_packed_floats": (c_size_t, [c_int] * 3),
    "mvq_conv1d_pack_dgrad_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mvq_conv_transpose1d_pack_dgrad_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mvq_conv1d_dgrad_f32": (c_int, [c_void_p] * 6 + [c_int] * 9 + [c_void_p]),
    "mvq_mul_dtanh_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mvq_layernorm_c_bwd_f32": (c_int, [c_void_p] * 8 + [c_int] * 3 + [c_size_t] * 2 + [c_float, c_void_p]),
    "mvq_gelu_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mvq_scale_tanh_f32": (c_int, [c_void_p, c_float, c_void_p, c_size_t, c_void_p]),
    "mvq_scale_tanh_bwd_f32": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_size_t, c_void_p]),
    "mvq_attention_bwd_f32": (c_int, [c_void_p] * 7 + [c_int] * 5 + [c_size_t] * 4 + [c_void_p]),
    "mvq_mul_scaled_f32": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_size_t, c_void_p]),
    "mvq_transpose2d_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "mvq_rowsum_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "mvq_stft_frames_f32": (c_int, [c_void_p] * 3 + [c_int] * 5 + [c_size_t] * 2 + [c_void_p]),
    "mvq_spec_mag_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_size_t, c_float, c_void_p]),
    "mvq_spec_loss_partial_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_size_t, c_void_p]),
    "mvq_spec_grad_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p] + [c_int] * 4 + [c_size_t, c_float, c_void_p]),
    "mvq_overlap_add_f32": (c_int, [c_void_p] * 3 + [c_int] * 5 + [c_void_p]),
    "mvq_l1_loss_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_float, c_size_t, c_void_p]),
    "mvq_mel_max_f32": (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_size_t, c_void_p]),
    "mvq_mel_cos_f32": (c_int, [c_void_p] * 5 + [c_float] + [c_int] * 3 + [c_size_t, c_float, c_int, c_void_p]),
    "mvq_mel_max_grad_f32": (c_int, [c_void_p] * 4 + [c_int, c_int, c_float, c_void_p]),
    "mvq_resample_f32": (c_int, [c_void_p] * 3 + [c_int] * 7 + [c_void_p]),
    "mvq_sumsq_partial_f32": (c_int, [c_void_p, c_void_p, c_int, c_size_t, c_void_p]),
    "mvq_adamw_f32": (c_int, [c_void_p] * 5 + [c_size_t] + [c_float] * 5 + [c_int, c_void_p]),
    "mvq_conv1d_padded_f32": (c_int, [c_void_p] * 9 + [c_int] * 10 + [c_void_p]),
    "mvq_residual_unit_padded_f32": (c_int, [c_void_p] * 13 + [c_int] * 5 + [c_void_p]),
    "mvq_conv_transpose1d_padded_f32": (c_int, [c_void_p] * 8 + [c_int] * 8 + [c_void_p]),
    "mvq_conv1d_packed_rows_f32": (c_int, [c_void_p] * 8 + [c_int] * 10 + [c_void_p]),
    "mvq_conv1d_vpacked_f32": (c_int, [c_void_p] * 8 + [c_int] * 13 + [c_void_p]),
    "mvq_conv_transpose1d_packed_rows_f32": (c_int, [c_void_p] * 8 + [c_int] * 9 + [c_void_p]),
    "mvq_conv_transpose1d_op_f32": (c_int, [c_void_p] * 8 + [c_int] * 9 + [c_void_p]),
    "mvq_layernorm_c_sub_f32": (c_int, [c_void_p] * 6 + [c_int] * 3 + [c_size_t] * 2 + [c_float, c_int, c_float, c_void_p]),
    "mvq_gelu_f32": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "mvq_sub3d_f32": (c_int, [c_void_p, c_size_t, c_size_t] * 3 + [c_int] * 3 + [c_void_p]),
    "mvq_copy3d_f32": (c_int, [c_void_p, c_size_t, c_size_t] * 2 + [c_int] * 3 + [c_void_p]),
    # whole stacks (ABI 3)
    "mvq_encoder_create": (c_int, [ctypes.POINTER(c_void_p), ctypes.POINTER(EncoderDesc), ctypes.POINTER(c_void_p), c_void_p, c_size_t, c_void_p]),
    "mvq_decoder_create": (c_int, [ctypes.POINTER(c_void_p), ctypes.POINTER(DecoderDesc), ctypes.POINTER(c_void_p), c_void_p, c_size_t, c_void_p]),
    "mvq_stack_destroy": (None, [c_void_p]),
    "mvq_stack_param_count": (c_int, [c_void_p]),
    "mvq_stack_param_info": (c_int, [c_void_p, c_int, c_char_p, c_int, ctypes.POINTER(c_int)]),
    "mvq_stack_weights_bytes": (c_size_t, [c_void_p]),
    "mvq_stack_set_plan": (c_int, [c_void_p, c_int, c_int]),
    "mvq_encoder_out_len": (c_int, [c_void_p, c_int]),
    "mvq_encoder_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "mvq_encoder_fwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "mvq_decoder_out_len": (c_int, [c_void_p, c_int]),
    "mvq_decoder_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "mvq_decoder_fwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "mvq_decoder_saved_bytes": (c_size_t, [c_void_p, c_int, c_int]),
    "mvq_decoder_fwd_saving_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "mvq_decoder_bwd_input_f32": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "mvq_ar_workspace_bytes": (c_size_t, [c_int, c_int]),
    "mvq_ar_latents_f32": (c_int, [ctypes.POINTER(ArArgs), c_void_p, c_size_t, c_void_p]),
    "mvq_ar_latents_staged_f32": (c_int, [ctypes.POINTER(ArArgs), c_void_p, c_size_t, c_void_p]),
    "mvq_ar_check": (c_int, [c_void_p, c_void_p]),
}


class MvqError(RuntimeError):
    """Raised when a libmvq_hip entry point returns a negative status."""


def build(force: bool = False) -> Path:
    """Compile the HIP sources for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", str(_PKG / "csrc"), "-j8"]
    if force:
        args.append("-B")
    res = subprocess.run(args, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libmvq_hip.so failed:\n" + res.stdout[-4000:] + res.stderr[-4000:])
    return SO_PATH


def lib():
    """The loaded library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not SO_PATH.exists():
            raise MvqError(f"{SO_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the MI355X path)")
        handle = ctypes.CDLL(str(SO_PATH))
        for name, (res, args) in EXPORTS.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        # a timing build (pieces of kernels compiled out) or an A/B override in the environment must never pass for the product
        flags = int(handle.mvq_build_flags())
        if flags & 0xFFFF and _os.environ.get("MVQ_ALLOW_TIMING_BUILD") != "1":
            raise MvqError(f"{SO_PATH}: mvq_build_flags() = {flags:#x} (timing build or MVQ_NO_DMA / MVQ_ROWFAST_MAX_KB / "
                           "MVQ_NO_TOKEN_RVQ in the environment; include/mvq.h).  Only tools/conv_microbench.py-style A/B runs "
                           "may load it: set MVQ_ALLOW_TIMING_BUILD=1 to do so.")
        _lib = handle
    return _lib


def build_flags() -> int:
    """mvq_build_flags() of the loaded library (0 = product build, no A/B override in the environment)."""
    return int(lib().mvq_build_flags())


def check(status: int, what: str) -> None:
    if status != 0:
        msg = lib().mvq_last_error()
        err = MvqError(f"{what} failed ({status}): {msg.decode() if msg else ''}")
        err.status = status                       # MVQ_EINVAL -1 | MVQ_EUNSUPPORTED -2 | MVQ_EHIP -3 (include/mvq.h)
        raise err
