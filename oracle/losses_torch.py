"""ORACLE (test infrastructure): torch-CPU restatement of the reference's training losses
(Training/compare_dacvsproposal_5.py:150-211) -- safe_l1, MultiResSTFTLoss, MelCosineLoss -- and of the total
``0.55*L1 + 0.25*MRSTFT + 0.20*MelCos`` (...:74-76,387).

torchaudio is absent from this image, so ``MelScale`` (n_mels=64, sr=24000, n_stft=257, f_min=0, f_max=12000,
norm=None, mel_scale="htk") is restated from its published definition: triangular filters on the HTK mel scale
m = 2595*log10(1 + f/700) between n_mels+2 equally spaced mel points, evaluated at the n_stft linearly spaced bin
frequencies, no area normalisation; the filterbank is applied as  mel = fb^T @ |spec|.
Pinned by tests/golden/g7 (the reference's own loss classes run on seeded signals, with this MelScale standing in for
torchaudio's; the filterbank itself is therefore "parity unpinned", everything around it is pinned).
"""
import math

import torch
import torch.nn.functional as F

W_WAV_L1, W_STFT, W_MELCOS = 0.55, 0.25, 0.20       # Training/compare_dacvsproposal_5.py:74-76


def finite_or_zero(x):
    return torch.nan_to_num(x, nan=0.0, posinf=0.0, neginf=0.0)


def safe_l1(x, y):
    return F.l1_loss(finite_or_zero(x), finite_or_zero(y))


def mel_filterbank(n_freqs=257, f_min=0.0, f_max=12000.0, n_mels=64, sample_rate=24000):
    """[n_freqs, n_mels] HTK triangular filterbank, norm=None."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)


class MelScale(torch.nn.Module):
    def __init__(self, n_mels=64, sample_rate=24000, n_stft=257, f_min=0.0, f_max=None, norm=None, mel_scale="htk"):
        super().__init__()
        assert norm is None and mel_scale == "htk"
        self.register_buffer("fb", mel_filterbank(n_stft, f_min, f_max if f_max is not None else sample_rate / 2, n_mels, sample_rate))

    def forward(self, spec):
        return torch.matmul(spec.transpose(-1, -2), self.fb).transpose(-1, -2)


def stft_mag(x, n_fft, hop, win, eps):
    dt = torch.float64 if x.dtype == torch.float64 else torch.float32      # float64 in: the exact-value yardstick (fixture G7 "f64.*")
    window = torch.hann_window(win, dtype=dt)
    spec = torch.stft(x.to(dt), n_fft=n_fft, hop_length=hop, win_length=win, window=window, center=True,
                      pad_mode="reflect", return_complex=True)
    return spec.abs().clamp_min(eps)


def mrstft(x, y, ffts=(256, 512, 1024), hops=(64, 128, 256), wins=(256, 512, 1024), eps=1e-7):
    x = finite_or_zero(x); y = finite_or_zero(y)
    used, sc, mag = 0, 0.0, 0.0
    for n, h, w in zip(ffts, hops, wins):
        if x.shape[-1] < max(8, w // 2):
            continue
        X = stft_mag(x.squeeze(1), n, h, w, eps); Y = stft_mag(y.squeeze(1), n, h, w, eps)
        num = (X - Y).pow(2).sum(dim=(1, 2)).sqrt()
        den = Y.pow(2).sum(dim=(1, 2)).sqrt().clamp_min(eps)
        sc = sc + (num / den).mean()
        mag = mag + F.l1_loss(X, Y)
        used += 1
    if used == 0:
        return 0.1 * F.l1_loss(x, y)
    return 0.5 * sc / used + 0.5 * mag / used


def mel_log(x_1T, fb, n_fft=512, hop=128, eps=1e-7):
    mag = stft_mag(x_1T[:, 0, :], n_fft, hop, n_fft, eps)
    M = torch.matmul(mag.transpose(-1, -2), fb.to(mag.dtype)).transpose(-1, -2)
    den = M.amax(dim=(1, 2), keepdim=True).clamp_min(eps)
    return (M / den + eps).log()


def melcos(x, y, fb=None, eps=1e-7):
    fb = mel_filterbank() if fb is None else fb
    X, Y = mel_log(x, fb, eps=eps), mel_log(y, fb, eps=eps)          # equal lengths on this path: no interpolation
    num = (X * Y).sum(dim=1)
    den = (X.norm(dim=1) * Y.norm(dim=1)).clamp_min(eps)
    return 1.0 - (num / den).clamp(-1, 1).mean()


def total_loss(y, tgt):
    l1, st, me = safe_l1(y, tgt), mrstft(y, tgt), melcos(y, tgt)
    return W_WAV_L1 * l1 + W_STFT * st + W_MELCOS * me, (l1, st, me)


@torch.no_grad()
def stsim_batch(ref_1T, est_1T, fb=None):
    """Evaluation/compare_dacvsproposal_5_eval.py:142-177 (equal-length inputs: no interpolation branch)."""
    fb = mel_filterbank() if fb is None else fb

    def mel_mag(x):
        x = x[:, 0, :] if x.dim() == 3 else x
        mag = stft_mag(x, 512, 128, 512, 1e-8)
        M = torch.matmul(mag.transpose(-1, -2), fb).transpose(-1, -2)
        return M / M.amax(dim=(1, 2), keepdim=True).clamp_min(1e-8)

    R, E = mel_mag(ref_1T), mel_mag(est_1T)
    num = (R * E).sum(dim=1)
    den = (R.norm(dim=1) * E.norm(dim=1)).clamp_min(1e-8)
    return [float(v) for v in 0.5 * ((num / den).clamp(-1, 1).mean(dim=-1) + 1.0)]


class Resample(torch.nn.Module):
    """Stand-in for torchaudio.transforms.Resample (absent) inside the reference's own functions when fixtures are
    generated: the restated sinc-hann resampler of oracle/oracle.py behind the torchaudio call shape."""

    def __init__(self, orig_freq=16000, new_freq=16000, **_):
        super().__init__()
        self.orig_freq, self.new_freq = int(orig_freq), int(new_freq)

    def forward(self, x):
        from . import oracle as _orc
        return torch.from_numpy(_orc.resample(x.detach().cpu().numpy(), self.orig_freq, self.new_freq))
