"""CPU: the resampler restatement (row f3).  torchaudio is absent, so the design is validated by what a sinc resampler
must do (PARITY UNPINNED against torchaudio itself): output length, unit DC gain, exact reproduction of band-limited
sinusoids at the new rate, and the host-side filter design of the product (torch, float64) equal to the oracle's (numpy)."""
import math

import numpy as np
import pytest

RATES = [(44100, 24000), (3000, 24000), (2800, 24000), (24000, 3000), (48000, 24000)]


@pytest.mark.parametrize("fi,fo", RATES)
def test_kernel_design_matches_product(fi, fo, orc):
    from multimodal_vqvae_compression_audio_tactile_amd.resample import sinc_resample_kernel
    k_o, w_o, o_o, n_o = orc.resample_kernel(fi, fo)
    k_p, w_p, o_p, n_p = sinc_resample_kernel(fi, fo)
    assert (w_o, o_o, n_o) == (w_p, o_p, n_p) and k_o.shape == tuple(k_p.shape)
    assert np.allclose(k_o, k_p.numpy(), rtol=0, atol=1e-7)
    g = math.gcd(fi, fo)
    assert k_o.shape == (fo // g, 2 * w_o + fi // g)
    assert w_o == math.ceil(6 * (fi // g) / (min(fi, fo) // g * 0.99))


@pytest.mark.parametrize("fi,fo", RATES)
def test_length_dc_gain_and_sinusoid(fi, fo, orc):
    L = fi // 2                                                   # 0.5 s
    y = orc.resample(np.ones((1, L), np.float32), fi, fo)
    assert y.shape == (1, math.ceil(fo * L / fi))
    mid = y[0, y.shape[1] // 4: 3 * y.shape[1] // 4]
    assert np.allclose(mid, 1.0, atol=2e-3)                       # unit DC gain away from the (zero-padded) edges
    f0 = 0.2 * min(fi, fo) / 2                                    # well inside both bands
    x = np.sin(2 * np.pi * f0 * np.arange(L) / fi).astype(np.float32)[None]
    y = orc.resample(x, fi, fo)[0]
    want = np.sin(2 * np.pi * f0 * np.arange(y.size) / fo)
    sl = slice(y.size // 8, 7 * y.size // 8)
    assert np.max(np.abs(y[sl] - want[sl])) < 3e-3


def test_up_then_down_is_identity_for_band_limited_input(orc):
    r = np.random.default_rng(0)
    spec = np.zeros(1501, np.complex128); spec[1:400] = r.standard_normal(399) + 1j * r.standard_normal(399)
    x = np.fft.irfft(spec, 3000); x = (x / np.abs(x).max()).astype(np.float32)[None]      # 1 s at 3 kHz, content < 400 Hz
    up = orc.resample(x, 3000, 24000)
    back = orc.resample(up, 24000, 3000)
    assert up.shape == (1, 24000) and back.shape == (1, 3000)
    assert np.max(np.abs(back[0, 100:-100] - x[0, 100:-100])) < 5e-3


def test_equal_rates_short_circuit(orc):
    x = np.arange(7, dtype=np.float32)[None]
    assert np.array_equal(orc.resample(x, 24000, 24000), x)


def test_aligned_3k_psnr_chain_matches_reference_fixture(orc):
    """G9 (reference align_pair_24k -> resample_f32 -> psnr_batch, with the restated resampler standing in for torchaudio):
    the oracle's own chain reproduces it -- shifts exactly, PSNR to 1e-4 dB."""
    import os, sys
    sys.path.insert(0, os.path.dirname(__file__))
    import golden_inputs as gi
    G9 = np.load(os.path.join(os.path.dirname(__file__), "golden", "g9_aligned_psnr.npz"))
    ref, est, lags = gi.aligned_psnr_inputs()
    for i in range(ref.shape[0]):
        r_a, e_a, s, _ = orc.align_by_xcorr(ref[i].numpy(), est[i].numpy(), 200)
        assert s == int(G9["shifts"][i]) == lags[i]
        r3, e3 = orc.resample(r_a, 24000, 3000), orc.resample(e_a, 24000, 3000)
        mse = max(float(np.mean((r3.astype(np.float64) - e3.astype(np.float64)) ** 2)), 1e-12)
        assert abs(10.0 * math.log10(1.0 / mse) - float(G9["psnr"][i])) < 1e-4
