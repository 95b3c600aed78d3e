"""torch-CPU restatement of the hot path (TEST INFRASTRUCTURE ONLY -- see oracle/README.md).

Purpose: (1) the tolerance anchor -- this is the arithmetic the reference really executes on its CPU path
(ATen / oneDNN fp32 ``F.conv1d``, ``F.conv_transpose1d``, Sleef ``sin``/``tanh``, ``softmax``, ``nn.GELU``),
so ``tests/test_oracle_vs_torch.py`` bounds the canonical-order C oracle against it; (2) the
``cpu_baseline`` ("port") leg of ``bench.py`` -- timed on the GPU node's host cores.

Backbone = the DAC-24 kHz architecture of the third-party ``descript-audio-codec`` package that the
reference loads with ``dac.DAC.load(dac.utils.download("24khz"))``
(Training/compare_dacvsproposal_5.py:329-338).  The package is absent offline; module structure and
parameter names are restated from SURVEY.md section 8a/8b [upstream dac/model/dac.py, dac/nn/layers.py,
dac/nn/quantize.py].  PARITY UNPINNED for the backbone.  The reference's own modules are restated from
Training/compare_dacvsproposal_5.py:214-326 and Evaluation/dac_vcpwq_proposed6_latency.py:339-487.
"""
from __future__ import annotations

import math
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    from torch.nn.utils import weight_norm as _weight_norm


def WNConv1d(*a, **k):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return _weight_norm(nn.Conv1d(*a, **k))


def WNConvTranspose1d(*a, **k):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return _weight_norm(nn.ConvTranspose1d(*a, **k))


def snake(x, alpha):
    return x + (alpha + 1e-9).reciprocal() * torch.sin(alpha * x).pow(2)


class Snake1d(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.alpha = nn.Parameter(torch.ones(1, channels, 1))

    def forward(self, x):
        return snake(x, self.alpha)


class ResidualUnit(nn.Module):
    def __init__(self, dim, dilation):
        super().__init__()
        pad = ((7 - 1) * dilation) // 2
        self.block = nn.Sequential(Snake1d(dim), WNConv1d(dim, dim, kernel_size=7, dilation=dilation, padding=pad),
                                   Snake1d(dim), WNConv1d(dim, dim, kernel_size=1))

    def forward(self, x):
        y = self.block(x)
        pad = (x.shape[-1] - y.shape[-1]) // 2
        if pad > 0:
            x = x[..., pad:-pad]
        return x + y


class EncoderBlock(nn.Module):
    def __init__(self, dim, stride):
        super().__init__()
        self.block = nn.Sequential(ResidualUnit(dim // 2, 1), ResidualUnit(dim // 2, 3), ResidualUnit(dim // 2, 9),
                                   Snake1d(dim // 2),
                                   WNConv1d(dim // 2, dim, kernel_size=2 * stride, stride=stride,
                                            padding=math.ceil(stride / 2)))

    def forward(self, x):
        return self.block(x)


class Encoder(nn.Module):
    def __init__(self, d_model=64, strides=(2, 4, 5, 8), d_latent=1024):
        super().__init__()
        layers = [WNConv1d(1, d_model, kernel_size=7, padding=3)]
        for s in strides:
            d_model *= 2
            layers.append(EncoderBlock(d_model, s))
        layers += [Snake1d(d_model), WNConv1d(d_model, d_latent, kernel_size=3, padding=1)]
        self.block = nn.Sequential(*layers)

    def forward(self, x):
        return self.block(x)


class DecoderBlock(nn.Module):
    def __init__(self, input_dim, output_dim, stride, output_padding=False):
        super().__init__()
        self.block = nn.Sequential(Snake1d(input_dim),
                                   WNConvTranspose1d(input_dim, output_dim, kernel_size=2 * stride, stride=stride,
                                                     padding=math.ceil(stride / 2),
                                                     output_padding=(stride % 2) if output_padding else 0),
                                   ResidualUnit(output_dim, 1), ResidualUnit(output_dim, 3), ResidualUnit(output_dim, 9))

    def forward(self, x):
        return self.block(x)


class Decoder(nn.Module):
    def __init__(self, input_channel=1024, channels=1536, rates=(8, 5, 4, 2), d_out=1, output_padding=False):
        super().__init__()
        layers = [WNConv1d(input_channel, channels, kernel_size=7, padding=3)]
        out = channels
        for i, s in enumerate(rates):
            inp, out = channels // 2 ** i, channels // 2 ** (i + 1)
            layers.append(DecoderBlock(inp, out, s, output_padding))
        layers += [Snake1d(out), WNConv1d(out, d_out, kernel_size=7, padding=3), nn.Tanh()]
        self.model = nn.Sequential(*layers)

    def forward(self, x):
        return self.model(x)


class VectorQuantize(nn.Module):
    margin_log = None          # set to a list by tests/golden/make_golden.py to record arg-max margins

    def __init__(self, input_dim, codebook_size, codebook_dim):
        super().__init__()
        self.in_proj = WNConv1d(input_dim, codebook_dim, kernel_size=1)
        self.out_proj = WNConv1d(codebook_dim, input_dim, kernel_size=1)
        self.codebook = nn.Embedding(codebook_size, codebook_dim)

    def decode_latents(self, latents):
        B, D, T = latents.shape
        enc = latents.permute(0, 2, 1).reshape(B * T, D)
        cb = self.codebook.weight
        enc = F.normalize(enc)
        cbn = F.normalize(cb)
        dist = enc.pow(2).sum(1, keepdim=True) - 2 * enc @ cbn.t() + cbn.pow(2).sum(1, keepdim=True).t()
        idx = (-dist).max(1)[1].reshape(B, T)
        if VectorQuantize.margin_log is not None:        # fixture generation: top-1 / top-2 gap of every arg-max
            top = (-dist).topk(2, dim=1)[0]
            VectorQuantize.margin_log.append(((top[:, 0] - top[:, 1]).reshape(B, T), dist.abs().amax(dim=1).reshape(B, T)))
        z_q = F.embedding(idx, cb).transpose(1, 2)
        return z_q, idx

    def forward(self, z):
        z_e = self.in_proj(z)
        z_q, idx = self.decode_latents(z_e)
        commit = F.mse_loss(z_e, z_q.detach(), reduction="none").mean([1, 2])
        cbl = F.mse_loss(z_q, z_e.detach(), reduction="none").mean([1, 2])
        z_q = z_e + (z_q - z_e).detach()
        return self.out_proj(z_q), commit, cbl, idx, z_e


class ResidualVectorQuantize(nn.Module):
    def __init__(self, input_dim=1024, n_codebooks=32, codebook_size=1024, codebook_dim=8):
        super().__init__()
        self.n_codebooks, self.codebook_size, self.codebook_dim = n_codebooks, codebook_size, codebook_dim
        self.quantizers = nn.ModuleList([VectorQuantize(input_dim, codebook_size, codebook_dim)
                                         for _ in range(n_codebooks)])

    def forward(self, z, n_quantizers=None):
        z_q, residual, commit, cbl = 0, z, 0, 0
        codes, latents = [], []
        if n_quantizers is None:
            n_quantizers = self.n_codebooks
        for i, q in enumerate(self.quantizers):
            if i >= n_quantizers:
                break
            z_q_i, c_i, b_i, idx_i, z_e_i = q(residual)
            z_q = z_q + z_q_i
            residual = residual - z_q_i
            commit = commit + c_i.mean()
            cbl = cbl + b_i.mean()
            codes.append(idx_i)
            latents.append(z_e_i)
        return z_q, torch.stack(codes, dim=1), torch.cat(latents, dim=1), commit, cbl


class DAC(nn.Module):
    def __init__(self, n_codebooks=32):
        super().__init__()
        self.encoder = Encoder()
        self.quantizer = ResidualVectorQuantize(n_codebooks=n_codebooks)
        self.decoder = Decoder()

    def encode(self, x, n_quantizers=None):
        return self.quantizer(self.encoder(x), n_quantizers)

    def decode(self, z):
        return self.decoder(z)


# ------------------------------------------------------------------------ reference-owned modules
class PosEnc1D(nn.Module):
    def __init__(self, c, max_len=8192):
        super().__init__()
        pe = torch.zeros(max_len, c)
        pos = torch.arange(0, max_len).unsqueeze(1)
        div = torch.exp(torch.arange(0, c, 2) * (-math.log(10000.0) / c))
        pe[:, 0::2] = torch.sin(pos * div)
        pe[:, 1::2] = torch.cos(pos * div)
        self.register_buffer("pe", pe)

    def forward(self, x):
        return x + self.pe[:x.size(-1), :].T.unsqueeze(0).to(x.dtype)


class TokenNorm(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.ln = nn.LayerNorm(c)

    def forward(self, z):
        return self.ln(z.permute(0, 2, 1)).permute(0, 2, 1)


class CrossPredictor(nn.Module):
    def __init__(self, c, heads=8, mlp_mul=2, dropout=0.1):
        super().__init__()
        self.pos = PosEnc1D(c)
        self.h, self.dh = heads, c // heads
        self.ln_q, self.ln_kv = nn.LayerNorm(c), nn.LayerNorm(c)
        self.q_proj, self.k_proj, self.v_proj = (nn.Linear(c, c, False) for _ in range(3))
        self.out = nn.Linear(c, c, False)
        self.drop = nn.Dropout(dropout)
        self.ffn = nn.Sequential(nn.LayerNorm(c), nn.Linear(c, mlp_mul * c), nn.GELU(), nn.Linear(mlp_mul * c, c))

    def _split(self, x):
        B, T, C = x.shape
        return x.view(B, T, self.h, self.dh).permute(0, 2, 1, 3)

    def forward(self, zt_prev, za):
        q = self.ln_q(self.pos(zt_prev).permute(0, 2, 1))
        kv = self.ln_kv(self.pos(za).permute(0, 2, 1))
        Q, K, V = self._split(self.q_proj(q)), self._split(self.k_proj(kv)), self._split(self.v_proj(kv))
        attn = (Q @ K.transpose(-2, -1)) / math.sqrt(self.dh)
        ctx = attn.softmax(dim=-1) @ V
        B, H, T, D = ctx.shape
        y = self.out(self.drop(ctx.permute(0, 2, 1, 3).contiguous().view(B, T, H * D)))
        y = y + q
        y = y + self.ffn(y)
        return y.permute(0, 2, 1)


class ResidualVQEMA(nn.Module):
    def __init__(self, dim, n_books, n_embed, decay=0.99):
        super().__init__()
        self.books = nn.ParameterList([nn.Parameter(torch.randn(n_embed, dim) / math.sqrt(dim)) for _ in range(n_books)])
        self.decay = float(decay)

    @staticmethod
    def _nearest_l2(x, emb):
        return (x @ emb.t() - 0.5 * (emb * emb).sum(dim=1).unsqueeze(0)).argmax(dim=1)

    def forward(self, z, n_books_use=None):
        n = len(self.books) if n_books_use is None else min(n_books_use, len(self.books))
        B, D, T = z.shape
        x = z.permute(0, 2, 1).reshape(B * T, D)
        residual, q_sum = x, torch.zeros_like(x)
        for cb in list(self.books)[:n]:
            emb = cb.detach()
            q = F.embedding(self._nearest_l2(residual, emb), emb)
            q_sum = q_sum + (q - residual).detach() + residual
            residual = residual - q
        return q_sum.view(B, T, D).permute(0, 2, 1).contiguous()


class ProposedEval(nn.Module):
    """Same attribute names as the reference so that checkpoint-shaped state dicts load directly."""

    def __init__(self, c_lat=1024, rvq_books=8, rvq_embed=512, n_codebooks=32, code_dim=96, chunk=16):
        super().__init__()
        self.A_ENC, self.T_ENC = Encoder(), Encoder()
        self.A_QUANT = ResidualVectorQuantize(n_codebooks=n_codebooks)
        self.T_DEC = Decoder()
        self.predict = CrossPredictor(c_lat)
        self.tokennorm = TokenNorm(c_lat)
        self.scale = nn.Parameter(torch.tensor(0.08))
        self.proj_down = nn.Conv1d(c_lat, code_dim, 1)
        self.proj_up = nn.Conv1d(code_dim, c_lat, 1)
        self.vq = ResidualVQEMA(code_dim, rvq_books, rvq_embed)
        self.chunk = chunk

    @torch.no_grad()
    def encode_latents(self, a, t, books_use=None, tactile_only=False):
        zt = self.T_ENC(t)
        B, C, Tlat = zt.shape
        qa = None
        if not tactile_only:
            qa, *_ = self.A_QUANT(self.A_ENC(a))
        z_run = torch.zeros_like(zt)
        for s in range(0, Tlat, self.chunk):
            e = min(Tlat, s + self.chunk)
            if tactile_only:
                z_pred = zt.new_zeros(B, C, e - s)
            else:
                zt_prev = zt.new_zeros(B, C, e - s)
                if s == 0:
                    zt_prev[..., 1:] = z_run[..., s:e - 1]
                else:
                    zt_prev[...] = z_run[..., s - 1:e - 1]
                z_pred = self.predict(zt_prev, qa[..., s:e])
            r = zt[..., s:e] - z_pred
            rN = torch.tanh(self.tokennorm(r))
            rD = self.proj_down(self.scale.clamp(5e-3, 0.5) * rN)
            qD = self.vq(rD, n_books_use=books_use)
            z_run[..., s:e] = self.proj_up(qD) + z_pred
        return z_run

    @torch.no_grad()
    def forward_eval(self, a, t, books_use=None, tactile_only=False):
        return self.T_DEC(self.encode_latents(a, t, books_use, tactile_only))

    def forward_step(self, a, tc):
        """The reference's training forward WITH autograd (Training/compare_dacvsproposal_5.py:293-326): z_run is
        written in place chunk by chunk, so chunk c+1's zt_prev column 0 carries gradient into chunk c's z_hat;
        z_pred is detached inside the residual; the RVQ is straight-through.  Gradient oracle for tests/test_gpu_train.py."""
        Tw = tc.shape[-1]
        with torch.no_grad():
            qa, *_ = self.A_QUANT(self.A_ENC(a))
            zt = self.T_ENC(tc)
        B, C, Tlat = zt.shape
        z_run = torch.zeros_like(zt)
        rD_all = []
        for s in range(0, Tlat, self.chunk):
            e = min(Tlat, s + self.chunk)
            zt_prev = zt.new_zeros(B, C, e - s)
            if s == 0:
                zt_prev[..., 1:] = z_run[..., s:e - 1]
            else:
                zt_prev[...] = z_run[..., s - 1:e - 1]
            z_pred = self.predict(zt_prev, qa[..., s:e])
            r = zt[..., s:e] - z_pred.detach()
            rN = torch.tanh(self.tokennorm(r))
            rD = self.proj_down(self.scale.clamp(5e-3, 0.5) * rN)
            qD = self.vq(rD)
            z_run[..., s:e] = z_pred + self.proj_up(qD)
            rD_all.append(rD.detach())
        y_hat = self.T_DEC(z_run)
        T = min(y_hat.shape[-1], tc.shape[-1], Tw)
        return {"y_hat": y_hat[..., :T], "tgt": tc[..., :T], "r_tokens": torch.cat(rD_all, dim=-1)}
