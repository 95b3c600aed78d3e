"""``Resample`` -- the ``torchaudio.transforms.Resample(orig_freq, new_freq)`` object the reference builds for every file
(``resample_to``: Training/compare_dacvsproposal_5.py:110-113; Evaluation/dac_vcpwq_proposed6_latency.py:151-156), default
arguments (sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99), on libmvq_hip.so.

The filter bank is designed in float64 on the host exactly as torchaudio's ``_get_sinc_resample_kernel`` does and rounded
to fp32; the convolution is one HBM-bound launch (``mvq_resample_f32``).  torchaudio itself is not a dependency.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import _lib, ops


def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """-> (kern[new, 2*width + orig] fp32, width, orig, new) with orig/new the rates divided by their gcd."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = int(math.ceil(lowpass_filter_width * orig / base))
    idx = torch.arange(-width, width + orig, dtype=torch.float64).unsqueeze(0) / orig
    t = (torch.arange(0, -new, -1, dtype=torch.float64).unsqueeze(1) / new + idx) * base
    t = t.clamp(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kern = torch.where(t == 0, torch.ones_like(t), torch.sin(t) / t) * window * (base / orig)
    return kern.to(torch.float32).contiguous(), width, orig, new


class Resample(nn.Module):
    def __init__(self, orig_freq: int = 16000, new_freq: int = 16000, resampling_method: str = "sinc_interp_hann",
                 lowpass_filter_width: int = 6, rolloff: float = 0.99):
        super().__init__()
        if resampling_method != "sinc_interp_hann":
            raise ops.MvqError("Resample: only the default sinc_interp_hann method is built (what the reference uses)")
        self.orig_freq, self.new_freq = int(orig_freq), int(new_freq)
        kern, self.width, self.orig, self.new = sinc_resample_kernel(orig_freq, new_freq, lowpass_filter_width, rolloff)
        self.register_buffer("kernel", kern, persistent=False)

    @torch.no_grad()
    def forward(self, waveform: torch.Tensor) -> torch.Tensor:
        if self.orig_freq == self.new_freq:
            return waveform
        x = ops._dev(waveform.to(torch.float32).contiguous(), "waveform")
        lead, L = x.shape[:-1], x.shape[-1]
        B = x.numel() // max(L, 1) if L else 0
        Lout = int(math.ceil(self.new * L / self.orig))
        y = torch.empty(lead + (Lout,), device=x.device, dtype=torch.float32)
        kern = self.kernel if self.kernel.device == x.device else self.kernel.to(x.device)
        ops.check(_lib.lib().mvq_resample_f32(x.data_ptr(), kern.data_ptr(), y.data_ptr(), B, L, Lout, self.orig, self.new,
                                              self.width, kern.shape[1], ops._stream()), "mvq_resample_f32")
        return y


def resample_to(wav: torch.Tensor, sr_in: int, sr_out: int) -> torch.Tensor:
    """The reference's helper (Training/compare_dacvsproposal_5.py:110-113)."""
    if sr_in == sr_out:
        return wav
    return Resample(sr_in, sr_out).to(wav.device)(wav)
