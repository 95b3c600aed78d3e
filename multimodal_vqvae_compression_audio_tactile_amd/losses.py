"""The reference's training losses on libmvq_hip.so, same class names and call shapes:
``safe_l1``, ``MultiResSTFTLoss``, ``MelCosineLoss`` (Training/compare_dacvsproposal_5.py:150-211) and the weighted
total the training step forms (...:74-76,387) as ``TrainingLoss``.

MI355X-first: the STFT is a windowed-frame matrix times a real DFT basis -- a k = 1 conv on the fp32 MFMA kernel
with N = 2*B*nframes columns (prediction and target side by side) -- and its gradient the transposed GEMM plus an
overlap-add; the 512/128 spectrogram is computed once and shared by the multi-resolution loss and the mel loss.  Loss
values and d loss / d y are produced in the same pass (the gradient is a closed form of quantities the forward already
has), and handed to autograd through one Function, so ``total.backward()`` continues into the decoder's HIP backward.
torchaudio's MelScale is not a dependency: the HTK filterbank is built here from its definition.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import ops

W_WAV_L1, W_STFT, W_MELCOS = 0.55, 0.25, 0.20        # Training/compare_dacvsproposal_5.py:74-76
MEL_NFFT, MEL_HOP, MEL_MELS = 512, 128, 64            # ...:79-81
TARGET_SR = 24000


def mel_filterbank(n_freqs=257, f_min=0.0, f_max=12000.0, n_mels=64, sample_rate=24000) -> torch.Tensor:
    """[n_freqs, n_mels] triangular HTK filterbank, norm=None (what MelScale(..., mel_scale="htk") applies)."""
    freqs = torch.linspace(0, sample_rate // 2, n_freqs, dtype=torch.float64)
    mel = lambda f: 2595.0 * math.log10(1.0 + f / 700.0)
    m_pts = torch.linspace(mel(f_min), mel(f_max), n_mels + 2, dtype=torch.float64)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    width = f_pts[1:] - f_pts[:-1]
    slope = f_pts.unsqueeze(0) - freqs.unsqueeze(1)
    fb = torch.minimum(-slope[:, :-2] / width[:-1], slope[:, 2:] / width[1:]).clamp_min(0.0)
    return fb.to(torch.float32)


class _SpecPlan:
    """Device constants of one STFT resolution: hann window, packed DFT basis (forward and transposed)."""

    _cache = {}

    def __init__(self, n_fft, device):
        self.n_fft, self.F = n_fft, n_fft // 2 + 1
        self.Fp = (self.F + 31) // 32 * 32          # the k = 1 MFMA path wants its reduction dimension in 32s
        k = torch.arange(self.F, dtype=torch.float64).unsqueeze(1)
        f = torch.arange(n_fft, dtype=torch.float64).unsqueeze(0)
        ang = 2.0 * math.pi * ((k * f) % n_fft) / n_fft
        W = torch.zeros(2 * self.Fp, n_fft, dtype=torch.float64)
        W[:self.F] = torch.cos(ang)
        W[self.Fp:self.Fp + self.F] = -torch.sin(ang)
        W = W.to(torch.float32).to(device).reshape(2 * self.Fp, n_fft, 1).contiguous()
        self.wp = ops.pack_conv1d(W)
        self.wp_t = ops.pack_conv1d_dgrad(W)
        self.window = torch.hann_window(n_fft, periodic=True, dtype=torch.float64).to(torch.float32).to(device)

    @classmethod
    def get(cls, n_fft, device):
        key = (n_fft, str(device))
        if key not in cls._cache:
            cls._cache[key] = cls(n_fft, device)
        return cls._cache[key]


class _MelPlan:
    _cache = {}

    def __init__(self, device, n_fft=MEL_NFFT, n_mels=MEL_MELS, sr=TARGET_SR):
        F = n_fft // 2 + 1
        Fp = (F + 31) // 32 * 32
        W = torch.zeros(n_mels, Fp)
        W[:, :F] = mel_filterbank(F, 0.0, sr * 0.5, n_mels, sr).t()
        W = W.to(device).reshape(n_mels, Fp, 1).contiguous()
        self.n_mels, self.Fp = n_mels, Fp
        self.wp, self.wp_t = ops.pack_conv1d(W), ops.pack_conv1d_dgrad(W)

    @classmethod
    def get(cls, device):
        key = str(device)
        if key not in cls._cache:
            cls._cache[key] = cls(device)
        return cls._cache[key]


class _Spectra:
    """|STFT| of prediction and target for one resolution (center=True, reflect padding, hann window = n_fft)."""

    def __init__(self, x, y, n_fft, hop, eps):
        plan = _SpecPlan.get(n_fft, x.device)
        B, T = x.shape
        self.plan, self.B, self.T, self.hop, self.eps = plan, B, T, hop, eps
        self.nfr = 1 + T // hop
        self.Nh = B * self.nfr
        frames = torch.empty(n_fft, 2 * self.Nh, device=x.device, dtype=torch.float32)
        ops.stft_frames(x, plan.window, frames, 0, n_fft, hop)
        ops.stft_frames(y, plan.window, frames, self.Nh, n_fft, hop)
        self.S = ops.conv1d(frames.reshape(1, n_fft, 2 * self.Nh), plan.wp, 2 * plan.Fp, 1)      # [1, 2Fp, 2Nh]
        self.mag = ops.spec_mag(self.S, plan.F, plan.Fp, eps)                                     # [Fp, 2Nh]

    def backward_into_(self, dy, coef_a, coef_b, extra):
        """dy[B,T] += d/dx of (per-item coef_a*(X-Y)^2/2-type term, coef_b*|X-Y|, extra . X)."""
        p = self.plan
        G = ops.spec_grad(self.S, self.mag, coef_a, coef_b, extra, p.F, p.Fp, self.B, self.nfr, self.eps)
        dF = ops.conv1d_dgrad(G.reshape(1, 2 * p.Fp, self.Nh), p.wp_t, p.n_fft, self.Nh, 1)
        return ops.overlap_add_(dy, dF.reshape(p.n_fft, self.Nh), p.window, p.n_fft, self.hop)


def _mrstft_terms(sp: _Spectra, eps):
    """Spectral convergence and magnitude-L1 of one resolution and the coefficients of their gradient."""
    F, B, nfr = sp.plan.F, sp.B, sp.nfr
    s = ops.spec_loss_sums(sp.mag, F, B, nfr)
    num = s[0].sqrt()
    den = s[1].sqrt().clamp_min(eps)
    sc = (num / den).mean()
    mag = s[2].sum() / float(B * F * nfr)
    coef_a = torch.where(num > 0, 1.0 / (float(B) * num * den), torch.zeros_like(num))    # d sc / dX = coef_a[b] * (X - Y)
    return sc, mag, coef_a.contiguous(), 1.0 / float(B * F * nfr)


def _mel_terms(sp: _Spectra, eps, want_grad, weight):
    """1 - mean cos of the log-mel frames; with want_grad the gradient w.r.t. the prediction's magnitudes [Fp, Nh]."""
    mp = _MelPlan.get(sp.S.device)
    B, nfr, Nh = sp.B, sp.nfr, sp.Nh
    M = ops.conv1d(sp.mag.reshape(1, mp.Fp, 2 * Nh), mp.wp, mp.n_mels, 1)                     # [1, n_mels, 2Nh]
    maxv, argm = ops.mel_max(M, mp.n_mels, B, nfr)
    cosv, dM, dden = ops.mel_cos(M, maxv, mp.n_mels, B, nfr, eps, coef=(-weight / float(Nh)) if want_grad else None)
    loss = 1.0 - ops.rowsum(cosv.reshape(1, Nh)).reshape(()) / float(Nh)
    extra = None
    if want_grad:
        ops.mel_max_grad_(dM, dden, maxv, argm, B, nfr, eps)
        extra = ops.conv1d_dgrad(dM.reshape(1, mp.n_mels, Nh), mp.wp_t, mp.Fp, Nh, 1).reshape(mp.Fp, Nh)
    return loss, extra


def _prep(x):
    if x.dim() == 3:
        x = x[:, 0, :]
    return ops._dev(x.detach().to(torch.float32).contiguous(), "waveform")


class _WithGrad(torch.autograd.Function):
    """value (0-d) with a precomputed gradient w.r.t. y: backward is g * dy."""

    @staticmethod
    def forward(ctx, y, value, dy):
        ctx.dy, ctx.shape = dy, y.shape
        return value.clone()

    @staticmethod
    def backward(ctx, g):
        return (ctx.dy * g).reshape(ctx.shape), None, None


def _evaluate(y, tgt, w_l1, w_stft, w_mel, ffts=(256, 512, 1024), hops=(64, 128, 256), eps=1e-7):
    """Weighted sum of the three losses, its parts, and (when y needs a gradient) d total / d y."""
    want = torch.is_grad_enabled() and y.requires_grad
    x, t = _prep(y), _prep(tgt)
    B, T = x.shape
    dy = torch.zeros(B, T, device=x.device, dtype=torch.float32) if want else None
    parts = {}
    if w_l1:
        parts["l1"] = ops.l1_loss_sum(x, t, dy, w_l1 / float(B * T)) / float(B * T)
    res = [(n, h) for n, h in zip(ffts, hops) if T >= max(8, n // 2)] if w_stft else []
    spectra = {}
    if w_stft:
        if not res:                                   # clip shorter than every resolution: 0.1 * l1 (Training/...5.py:171)
            parts["stft"] = 0.1 * ops.l1_loss_sum(x, t, dy, 0.1 * w_stft / float(B * T)) / float(B * T)
        sc_sum, mag_sum = 0.0, 0.0
        for n, h in res:
            sp = spectra[(n, h)] = _Spectra(x, t, n, h, eps)
            sc, mag, ca, cb = _mrstft_terms(sp, eps)
            sc_sum, mag_sum = sc_sum + sc, mag_sum + mag
            sp.coefs = (ca * (0.5 * w_stft / len(res)), cb * 0.5 * w_stft / len(res))
        if res:
            parts["stft"] = 0.5 * sc_sum / len(res) + 0.5 * mag_sum / len(res)
    extra = None
    if w_mel:
        key = (MEL_NFFT, MEL_HOP)
        sp = spectra.get(key)
        if sp is None:
            sp = _Spectra(x, t, MEL_NFFT, MEL_HOP, eps); sp.coefs = (None, 0.0)
            spectra[key] = sp
        parts["mel"], extra = _mel_terms(sp, eps, want, w_mel)
        sp.extra = extra
    if want:
        for sp in spectra.values():
            ca, cb = sp.coefs
            sp.backward_into_(dy, ca, cb, getattr(sp, "extra", None))
    total = sum(w * parts[k] for k, w in (("l1", w_l1), ("stft", w_stft), ("mel", w_mel)) if w)
    if want:
        total = _WithGrad.apply(y, total, dy)
    return total, parts


def safe_l1(x, y):
    """F.l1_loss(finite_or_zero(x), finite_or_zero(y))  (Training/...5.py:211)."""
    return _evaluate(x, y, 1.0, 0.0, 0.0)[0]


class MultiResSTFTLoss(nn.Module):
    def __init__(self, ffts=(256, 512, 1024), hops=(64, 128, 256), wins=(256, 512, 1024), eps=1e-7):
        super().__init__()
        if tuple(wins) != tuple(ffts):
            raise ops.MvqError("MultiResSTFTLoss: win_length must equal n_fft (the reference's configuration)")
        self.ffts, self.hops, self.wins, self.eps = tuple(ffts), tuple(hops), tuple(wins), eps

    def forward(self, x, y):
        return _evaluate(x, y, 0.0, 1.0, 0.0, self.ffts, self.hops, self.eps)[0]


class MelCosineLoss(nn.Module):
    def __init__(self, sr=TARGET_SR, n_fft=MEL_NFFT, hop=MEL_HOP, n_mels=MEL_MELS, eps=1e-7):
        super().__init__()
        if (sr, n_fft, hop, n_mels) != (TARGET_SR, MEL_NFFT, MEL_HOP, MEL_MELS):
            raise ops.MvqError("MelCosineLoss: only the reference's configuration (24 kHz, 512/128, 64 mels) is built")
        self.eps = eps

    def forward(self, x, y):
        return _evaluate(x, y, 0.0, 0.0, 1.0, eps=self.eps)[0]


class TrainingLoss(nn.Module):
    """total = 0.55*L1 + 0.25*MRSTFT + 0.20*MelCos in one pass (the 512/128 spectrogram is shared); ``parts`` holds the
    three terms of the last call as 0-d device tensors (the reference logs them, Training/...5.py:399)."""

    def __init__(self, w_l1=W_WAV_L1, w_stft=W_STFT, w_mel=W_MELCOS):
        super().__init__()
        self.w = (w_l1, w_stft, w_mel)
        self.parts = {}

    def forward(self, y, tgt):
        total, self.parts = _evaluate(y, tgt, *self.w)
        return total


@torch.no_grad()
def stsim_batch(ref_1T, est_1T):
    """Spectro-temporal similarity per item (Evaluation/compare_dacvsproposal_5_eval.py:142-177): 0.5*(1 + mean over
    frames of the cosine between the max-normalised 64-band HTK mel magnitudes, n_fft 512, hop 128).  Returns a list of
    floats like the reference (one device->host copy).  Inputs of equal length (the reference crops them first)."""
    r, e = _prep(ref_1T), _prep(est_1T)
    if r.shape != e.shape:
        raise ops.MvqError("stsim_batch: crop_match the signals first (equal lengths)")
    eps = 1e-8
    sp = _Spectra(r, e, MEL_NFFT, MEL_HOP, eps)
    mp = _MelPlan.get(r.device)
    M = ops.conv1d(sp.mag.reshape(1, mp.Fp, 2 * sp.Nh), mp.wp, mp.n_mels, 1)
    maxv, _ = ops.mel_max(M, mp.n_mels, sp.B, sp.nfr)
    cosv, _, _ = ops.mel_cos(M, maxv, mp.n_mels, sp.B, sp.nfr, eps, use_log=False)
    mean_cos = ops.rowsum(cosv.reshape(sp.B, sp.nfr)) / float(sp.nfr)
    return [float(v) for v in (0.5 * (mean_cos + 1.0)).cpu()]
