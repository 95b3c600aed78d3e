"""Seeded synthetic weights and signals (there is no network for checkpoints or the corpus).

Everything is generated on the CPU with ``torch.Generator`` so that the HIP path, the CPU oracle and the
torch restatement all see bit-identical parameters.  Each tensor has its own generator seeded from
(seed, position in the ordered name list), so adding a tensor never changes the others.

Names follow the checkpoints the reference scripts write/read ("Checkpoint"):
``A_ENC.* / A_QUANT.* / T_ENC.* / T_DEC.*`` hold a DAC-24 kHz backbone with upstream parameter names
(``block.{i}...weight_g|weight_v|bias|alpha``, ``model.{i}...``, ``quantizers.{i}.in_proj|out_proj|codebook``),
``predict.* / tokennorm.ln.* / scale / proj_down.* / proj_up.* / vq.books.{i}`` are the reference's own
modules (Training/compare_dacvsproposal_5.py:279-290).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch

ENC_RATES = (2, 4, 5, 8)
DEC_RATES = (8, 5, 4, 2)
ENC_DIM = 64
DEC_DIM = 1536
LATENT = 1024
N_CODEBOOKS = 32
CODEBOOK_SIZE = 1024
CODEBOOK_DIM = 8
CODE_DIM = 96
TARGET_SR = 24000


class _Draw:
    def __init__(self, seed: int):
        self.seed = int(seed)
        self.i = 0

    def gen(self) -> torch.Generator:
        g = torch.Generator(device="cpu")
        g.manual_seed(self.seed * 1000003 + self.i)
        self.i += 1
        return g

    def normal(self, shape, std):
        return torch.randn(shape, generator=self.gen(), dtype=torch.float32) * float(std)

    def uniform(self, shape, lo, hi):
        return torch.rand(shape, generator=self.gen(), dtype=torch.float32) * (hi - lo) + lo


def _wn_conv(sd, d: _Draw, name, cout, cin, k, gain=1.0, transpose=False, taps_per_out=None):
    """weight-normalised conv: weight_v ~ N(0, gain^2/fan_in), weight_g = ||v|| per dim-0 row (=> w == v)."""
    if transpose:
        shape = (cin, cout, k)
        fan = cin * (taps_per_out or k)
    else:
        shape = (cout, cin, k)
        fan = cin * k
    v = d.normal(shape, gain / math.sqrt(fan))
    g = v.reshape(shape[0], -1).norm(dim=1).reshape(shape[0], 1, 1) * d.uniform((shape[0], 1, 1), 0.9, 1.1)
    sd[name + ".weight_g"] = g
    sd[name + ".weight_v"] = v
    sd[name + ".bias"] = d.normal((cout,), 0.02)


def _alpha(sd, d: _Draw, name, c):
    sd[name + ".alpha"] = d.uniform((1, c, 1), 0.6, 1.4)


def _res_unit(sd, d, p, c):
    _alpha(sd, d, p + ".block.0", c)
    _wn_conv(sd, d, p + ".block.1", c, c, 7, gain=0.7)
    _alpha(sd, d, p + ".block.2", c)
    _wn_conv(sd, d, p + ".block.3", c, c, 1, gain=0.35)


def encoder_state(seed: int, rates=ENC_RATES, d_model=ENC_DIM, d_latent=LATENT, prefix="") -> "OrderedDict[str, torch.Tensor]":
    sd, d = OrderedDict(), _Draw(seed)
    _wn_conv(sd, d, prefix + "block.0", d_model, 1, 7, gain=2.0)
    c = d_model
    for i, s in enumerate(rates):
        p = f"{prefix}block.{i + 1}"
        for j in range(3):
            _res_unit(sd, d, f"{p}.block.{j}", c)
        _alpha(sd, d, p + ".block.3", c)
        _wn_conv(sd, d, p + ".block.4", 2 * c, c, 2 * s, gain=0.7)
        c *= 2
    n = len(rates) + 1
    _alpha(sd, d, f"{prefix}block.{n}", c)
    _wn_conv(sd, d, f"{prefix}block.{n + 1}", d_latent, c, 3, gain=0.7)
    return sd


def decoder_state(seed: int, rates=DEC_RATES, channels=DEC_DIM, d_in=LATENT, prefix="") -> "OrderedDict[str, torch.Tensor]":
    sd, d = OrderedDict(), _Draw(seed)
    _wn_conv(sd, d, prefix + "model.0", channels, d_in, 7)
    c = channels
    for i, s in enumerate(rates):
        p = f"{prefix}model.{i + 1}"
        _alpha(sd, d, p + ".block.0", c)
        _wn_conv(sd, d, p + ".block.1", c // 2, c, 2 * s, gain=0.7, transpose=True, taps_per_out=2)
        c //= 2
        for j in range(3):
            _res_unit(sd, d, f"{p}.block.{j + 2}", c)
    n = len(rates) + 1
    _alpha(sd, d, f"{prefix}model.{n}", c)
    _wn_conv(sd, d, f"{prefix}model.{n + 1}", 1, c, 7, gain=0.5)
    return sd


def quantizer_state(seed: int, n_codebooks=N_CODEBOOKS, K=CODEBOOK_SIZE, dc=CODEBOOK_DIM, c=LATENT, prefix=""):
    sd, d = OrderedDict(), _Draw(seed)
    for i in range(n_codebooks):
        p = f"{prefix}quantizers.{i}"
        _wn_conv(sd, d, p + ".in_proj", dc, c, 1, gain=4.0)
        _wn_conv(sd, d, p + ".out_proj", c, dc, 1, gain=0.6)
        sd[p + ".codebook.weight"] = d.normal((K, dc), 1.0)
    return sd


def pos_table(c, max_len=8192):
    """PosEnc1D buffer, built with the same torch CPU ops as Training/compare_dacvsproposal_5.py:214-220."""
    pe = torch.zeros(max_len, c)
    pos = torch.arange(0, max_len).unsqueeze(1)
    div = torch.exp(torch.arange(0, c, 2) * (-math.log(10000.0) / c))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def proposed_head_state(seed: int, c=LATENT, rvq_books=8, rvq_embed=512, code_dim=CODE_DIM, mlp_mul=2,
                        with_pe=True):
    """The reference's own trainable modules (predict, tokennorm, scale, proj_down/up, vq.books)."""
    sd, d = OrderedDict(), _Draw(seed)
    b = 1.0 / math.sqrt(c)
    if with_pe:
        sd["predict.pos.pe"] = pos_table(c)
    for n in ("ln_q", "ln_kv"):
        sd[f"predict.{n}.weight"] = d.uniform((c,), 0.8, 1.2)
        sd[f"predict.{n}.bias"] = d.normal((c,), 0.05)
    for n in ("q_proj", "k_proj", "v_proj", "out"):
        sd[f"predict.{n}.weight"] = d.uniform((c, c), -b, b) * 1.7
    sd["predict.ffn.0.weight"] = d.uniform((c,), 0.8, 1.2)
    sd["predict.ffn.0.bias"] = d.normal((c,), 0.05)
    sd["predict.ffn.1.weight"] = d.uniform((mlp_mul * c, c), -b, b) * 1.7
    sd["predict.ffn.1.bias"] = d.uniform((mlp_mul * c,), -b, b)
    b2 = 1.0 / math.sqrt(mlp_mul * c)
    sd["predict.ffn.3.weight"] = d.uniform((c, mlp_mul * c), -b2, b2) * 1.7
    sd["predict.ffn.3.bias"] = d.uniform((c,), -b2, b2)
    sd["tokennorm.ln.weight"] = d.uniform((c,), 0.8, 1.2)
    sd["tokennorm.ln.bias"] = d.normal((c,), 0.05)
    sd["scale"] = torch.tensor(0.08)
    sd["proj_down.weight"] = d.uniform((code_dim, c, 1), -b, b) * 6.0
    sd["proj_down.bias"] = d.uniform((code_dim,), -b, b)
    b3 = 1.0 / math.sqrt(code_dim)
    sd["proj_up.weight"] = d.uniform((c, code_dim, 1), -b3, b3) * 1.7
    sd["proj_up.bias"] = d.uniform((c,), -b3, b3)
    for i in range(rvq_books):
        # ResidualVQEMA.__init__: randn(n_embed, dim)/sqrt(dim)  (Training/compare_dacvsproposal_5.py:249);
        # scaled so that successive residual books see comparable magnitudes
        sd[f"vq.books.{i}"] = d.normal((rvq_embed, code_dim), (0.6 ** i) / math.sqrt(code_dim) * 0.25)
    return sd


def proposed_model_state(seed: int = 7, rvq_books=8, rvq_embed=512, with_pe=True):
    """Full checkpoint-shaped state dict of ProposedEval / AllPredAR with DAC-24k backbones."""
    sd = OrderedDict()
    sd.update(encoder_state(seed * 10 + 1, prefix="A_ENC."))
    sd.update(quantizer_state(seed * 10 + 2, prefix="A_QUANT."))
    sd.update(encoder_state(seed * 10 + 3, prefix="T_ENC."))
    sd.update(decoder_state(seed * 10 + 4, prefix="T_DEC."))
    sd.update(proposed_head_state(seed * 10 + 5, rvq_books=rvq_books, rvq_embed=rvq_embed, with_pe=with_pe))
    return sd


def dac_state(seed: int = 7, n_codebooks=N_CODEBOOKS):
    """State dict of one DAC-24k model: encoder.*, quantizer.*, decoder.* (upstream names)."""
    sd = OrderedDict()
    sd.update(encoder_state(seed * 10 + 1, prefix="encoder."))
    sd.update(quantizer_state(seed * 10 + 2, n_codebooks=n_codebooks, prefix="quantizer."))
    sd.update(decoder_state(seed * 10 + 4, prefix="decoder."))
    return sd


# ---------------------------------------------------------------------------------------- signals
def _lowpass_fft(x: torch.Tensor, sr: int, cutoff_hz: float) -> torch.Tensor:
    X = torch.fft.rfft(x.to(torch.float64), dim=-1)
    f = torch.fft.rfftfreq(x.shape[-1], d=1.0 / sr)
    X = X * (f <= cutoff_hz).to(X.dtype)
    return torch.fft.irfft(X, n=x.shape[-1], dim=-1).to(torch.float32)


def tactile_segments(batch: int, seed: int = 7, T: int = TARGET_SR, sr: int = TARGET_SR) -> torch.Tensor:
    """Vibrotactile-like segments [B,1,T] at 24 kHz: noise low-passed at 1.4 kHz (content of a 2.8-3 kHz
    recording after upsampling), slow random envelope, peak 0.9, clamped like sanitize_wave
    (Training/compare_dacvsproposal_5.py:95-97)."""
    g = torch.Generator(device="cpu"); g.manual_seed(seed * 7919 + 11)
    x = torch.randn(batch, 1, T, generator=g)
    x = _lowpass_fft(x, sr, 1400.0)
    env = _lowpass_fft(torch.randn(batch, 1, T, generator=g), sr, 6.0)
    env = 0.35 + (env - env.amin(-1, keepdim=True)) / (env.amax(-1, keepdim=True) - env.amin(-1, keepdim=True) + 1e-9)
    x = x * env
    x = 0.9 * x / x.abs().amax(-1, keepdim=True).clamp_min(1e-9)
    return x.clamp(-1.0, 1.0).contiguous()


def audio_segments(batch: int, seed: int = 7, T: int = TARGET_SR, sr: int = TARGET_SR) -> torch.Tensor:
    """Audio-like segments [B,1,T] at 24 kHz: pink-ish (1/f-shaped) noise, peak 0.9, clamped."""
    g = torch.Generator(device="cpu"); g.manual_seed(seed * 7919 + 23)
    x = torch.randn(batch, 1, T, generator=g)
    X = torch.fft.rfft(x.to(torch.float64), dim=-1)
    f = torch.fft.rfftfreq(T, d=1.0 / sr)
    shape = 1.0 / torch.sqrt(torch.clamp(f, min=20.0))
    x = torch.fft.irfft(X * shape.to(X.dtype), n=T, dim=-1).to(torch.float32)
    x = 0.9 * x / x.abs().amax(-1, keepdim=True).clamp_min(1e-9)
    return x.clamp(-1.0, 1.0).contiguous()
