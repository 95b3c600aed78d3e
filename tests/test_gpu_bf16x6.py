"""The OPT-IN, NON-PARITY arithmetic mode "bf16x6" (include/mvq.h; csrc/conv_k7_bf16.hip): three-piece bf16 split of both
operands, six piece products per fp32 product on the bf16 matrix cores.  It is not bit-identical to the oracle by design, so
these tests check (1) the split itself bit for bit, (2) the conv against the EXACT fp32 path and against a float64 yardstick:
its error must stay at the fp32 chain's own rounding level."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _split_ref(x):
    a0 = x.bfloat16(); r1 = x - a0.float(); a1 = r1.bfloat16(); r2 = r1 - a1.float(); a2 = r2.bfloat16()
    return a0, a1, a2


def test_split_is_bit_exact_and_lossless_to_24_bits(dev):
    from multimodal_vqvae_compression_audio_tactile_amd import ops
    torch.manual_seed(3)
    B, C, T = 3, 32, 77
    x = (torch.randn(B, C, T, device=dev) * torch.logspace(-6, 3, T, device=dev)).contiguous()
    xs = ops.bf16x3_split(x).view(torch.bfloat16).reshape(B, C // 8, 3, T, 8)
    ref = _split_ref(x)
    for p in range(3):
        want = ref[p].reshape(B, C // 8, 8, T).permute(0, 1, 3, 2)          # [B][octet][T][8]
        assert torch.equal(xs[:, :, p].view(torch.int16), want.contiguous().view(torch.int16)), f"piece {p}"
    back = sum(r.float() for r in ref)
    assert ((back - x).abs() <= x.abs() * 2.0 ** -23).all()


@pytest.mark.parametrize("wide", [0, 1])
@pytest.mark.parametrize("C,T,dil,tvalid", [(128, 300, 1, 0), (256, 601, 3, 0), (384, 260, 9, 259), (256, 75, 9, 0)])
def test_conv_k7_bf16x6_is_fp32_accurate(C, T, dil, tvalid, wide, dev):
    from multimodal_vqvae_compression_audio_tactile_amd import ops
    torch.manual_seed(C + T + dil)
    B = 3
    x = torch.randn(B, C, T, device=dev)
    if tvalid:
        x[..., tvalid:] = 0.0
    w = torch.randn(C, C, 7, device=dev) / math.sqrt(7 * C)
    bias = torch.randn(C, device=dev)
    alpha = torch.rand(C, device=dev) + 0.5
    y = ops.conv1d_k7_bf16x6(ops.bf16x3_split(x), ops.pack_conv1d_k7_bf16x3(w), B, C, T, C, dil, bias=bias, alpha_out=alpha,
                             tvalid=tvalid, wide=wide)
    Tx = T
    xe = x
    if T % 4:                                                                 # the exact path wants 16-byte rows for its fast form; any T works
        pass
    exact = ops.conv1d(xe, ops.pack_conv1d(w), C, 7, bias=bias, dil=dil, pad=3 * dil, alpha_out=alpha, tvalid=tvalid)
    h = torch.nn.functional.conv1d(x.double(), w.double(), bias.double(), padding=3 * dil, dilation=dil)
    a = alpha.double()[None, :, None]
    truth = h + torch.sin(a * h) ** 2 / (a + 1e-9)
    if tvalid:
        truth[..., tvalid:] = 0.0
        assert (y[..., tvalid:] == 0).all()
    assert y.shape == (B, C, Tx) and torch.isfinite(y).all()
    e_new = (y.double() - truth).abs()
    e_old = (exact.double() - truth).abs()
    scale = truth.abs().max().item()
    # the exact chain's own error is a few 1e-7 of the scale; the split path must be in the same class (not 1e-5 like bf16x3 would be)
    assert e_new.max().item() <= max(4.0 * e_old.max().item(), 2e-6 * scale), (e_new.max().item(), e_old.max().item(), scale)
    assert e_new.pow(2).mean().sqrt().item() <= 2.0 * e_old.pow(2).mean().sqrt().item() + 1e-9


def test_conv_k7_bf16x6_rejects_bad_shapes(dev):
    from multimodal_vqvae_compression_audio_tactile_amd import ops
    from multimodal_vqvae_compression_audio_tactile_amd._lib import MvqError
    with pytest.raises(MvqError):
        ops.pack_conv1d_k7_bf16x3(torch.randn(192, 192, 7, device=dev))          # Cout % 128
    with pytest.raises(MvqError):
        ops.bf16x3_split(torch.randn(1, 12, 16, device=dev))                      # C % 8
    xs = ops.bf16x3_split(torch.randn(1, 128, 16, device=dev)); wq = ops.pack_conv1d_k7_bf16x3(torch.randn(128, 128, 7, device=dev))
    with pytest.raises(MvqError):
        ops.conv1d_k7_bf16x6(xs, wq, 1, 128, 16, 128, 2)                          # dilation
