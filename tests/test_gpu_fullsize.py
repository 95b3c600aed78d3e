"""-m gpu: BASELINE-size checks.  One full 1-s segment pair goes through the oracle end to end (bit-exact); the batched
run at the bench size is then tied to it by size-independent properties: a segment's result does not depend on its
batch mates or its position in the batch, runs are bit-reproducible, and a conv is exactly linear under power-of-two
scaling.  Plus the edge cases of the boundary: empty batch, inputs shorter than a token, lengths that are not a multiple
of the hop, long clips, non-finite samples."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _np(sd):
    return {k: v.numpy() for k, v in sd.items()}


@pytest.fixture(scope="module")
def model(dev):
    from multimodal_vqvae_compression_audio_tactile_amd import build_proposed, synth
    sd = synth.proposed_model_state(7, rvq_books=8, rvq_embed=512)
    return sd, build_proposed(sd, rvq_books=8, rvq_embed=512, device=dev)


@pytest.fixture(scope="module")
def oracle_segment(model, orc):
    """One full 1-s segment pair through the C oracle (joint path: latents + waveform; tactile-only chain: waveform)."""
    from multimodal_vqvae_compression_audio_tactile_amd import synth
    sd, _ = model
    a, t = synth.audio_segments(1, seed=21), synth.tactile_segments(1, seed=21)
    want_z = orc.proposed_encode_latents(_np(sd), a.numpy(), t.numpy())
    want_y = orc.dac_decoder(_np(sd), want_z, prefix="T_DEC.")
    want_y_tact = orc.proposed_forward_eval(_np(sd), None, t.numpy(), tactile_only=True)
    return a, t, want_z, want_y, want_y_tact


def test_full_segment_against_oracle(model, oracle_segment, orc, dev):
    """configs[2] at its real size (24 000 + 24 000 samples -> 75 tokens -> 23 992 samples), B = 1, vs the oracle."""
    from multimodal_vqvae_compression_audio_tactile_amd import psnr_batch
    sd, net = model
    a, t, want_z, want_y, _ = oracle_segment
    z = net.encode_latents(a.to(dev), t.to(dev))
    y = net.T_DEC(z)
    assert z.shape == (1, 1024, 75) and y.shape == (1, 1, 23992)
    assert np.array_equal(z.cpu().numpy(), want_z)
    assert np.array_equal(y.cpu().numpy(), want_y)
    p = psnr_batch(t[..., :23992].to(dev), y)[0]
    assert abs(p - orc.psnr_batch(t.numpy()[..., :23992], want_y)[0]) <= 1e-5


def test_batch_independence_and_reproducibility_at_bench_size(model, dev):
    from multimodal_vqvae_compression_audio_tactile_amd import synth
    sd, net = model
    B = 48
    a, t = synth.audio_segments(B, seed=5).to(dev), synth.tactile_segments(B, seed=5).to(dev)
    y = net.forward_eval(a, t)
    assert y.shape == (B, 1, 23992) and torch.isfinite(y).all()
    assert torch.equal(y, net.forward_eval(a, t))                              # bit-reproducible
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).to(dev)
    assert torch.equal(net.forward_eval(a[perm], t[perm]), y[perm])            # position in the batch is irrelevant
    for i in (0, 17, B - 1):                                                   # batch mates are irrelevant (tile shapes differ!)
        assert torch.equal(net.forward_eval(a[i:i + 1], t[i:i + 1]), y[i:i + 1])
    z = net.encode_latents(a[:4], t[:4])
    assert torch.equal(net.T_DEC(z), y[:4])                                    # split encode / decode API == forward_eval


def test_headline_batch_256_is_tied_to_the_oracle(model, oracle_segment, dev):
    """The configuration bench.py's headline is quoted on (SURVEY 8d configs 2 / 3, B = 256: single-stream encoders, 23 full
    column tiles + tail split launches, packed latent rows with a full 32 x 8 layout): rows of the 256-segment batch that hold
    the oracle-checked segment must carry the ORACLE's waveform bit for bit, wherever they sit in the batch; other rows must
    equal their own B = 1 run; same for the tactile-only chain."""
    from multimodal_vqvae_compression_audio_tactile_amd import synth
    sd, net = model
    a1, t1, want_z, want_y, want_y_tact = oracle_segment
    B = 256
    a, t = synth.audio_segments(B, seed=5), synth.tactile_segments(B, seed=5)
    planted = (3, 129, B - 1)                         # first packed row, a middle one, the last column of the last row
    for i in planted:
        a[i], t[i] = a1[0], t1[0]
    a, t = a.to(dev), t.to(dev)
    y = net.forward_eval(a, t)
    assert y.shape == (B, 1, 23992) and torch.isfinite(y).all()
    want = torch.from_numpy(want_y).to(dev)
    for i in planted:
        assert torch.equal(y[i:i + 1], want), f"joint path, segment {i} of the 256-segment batch differs from the oracle"
    for i in (0, 77, 200):                            # batch mates are irrelevant (B = 1 runs other tiles / launch plans)
        assert torch.equal(net.forward_eval(a[i:i + 1], t[i:i + 1]), y[i:i + 1])
    z = net.encode_latents(a, t)
    zw = torch.from_numpy(want_z).to(dev)
    for i in planted:
        assert torch.equal(z[i:i + 1], zw)
    del y, z
    # tactile-only chain (configs[1]) at the same size
    yt = net.forward_eval_tactile_only(t)
    assert yt.shape == (B, 1, 23992)
    wt = torch.from_numpy(want_y_tact).to(dev)
    for i in planted:
        assert torch.equal(yt[i:i + 1], wt), f"tactile chain, segment {i} of the 256-segment batch differs from the oracle"
    for i in (0, 200):
        assert torch.equal(net.forward_eval_tactile_only(t[i:i + 1]), yt[i:i + 1])


def test_conv_power_of_two_linearity(dev):
    from multimodal_vqvae_compression_audio_tactile_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(8, 768, 600, generator=g).to(dev)
    w = (torch.randn(768, 768, 7, generator=g) / 73.0).to(dev)
    wp = ops.pack_conv1d(w)
    y = ops.conv1d(x, wp, 768, 7, dil=3, pad=9)
    assert torch.equal(ops.conv1d(4.0 * x, wp, 768, 7, dil=3, pad=9), 4.0 * y)
    assert torch.equal(ops.conv1d(-0.25 * x, wp, 768, 7, dil=3, pad=9), -0.25 * y)
    xa, xb = x.clone(), torch.zeros_like(x)
    xa[..., 300:] = 0; xb[..., 300:] = x[..., 300:]
    ya, yb = ops.conv1d(xa, wp, 768, 7, dil=3, pad=9), ops.conv1d(xb, wp, 768, 7, dil=3, pad=9)
    far = slice(0, 300 - 9)                                                    # outside the receptive field of the split
    assert torch.equal(ya[..., far], y[..., far]) and torch.equal(yb[..., 300 + 9:], y[..., 300 + 9:])


def test_edge_cases(model, orc, dev):
    import multimodal_vqvae_compression_audio_tactile_amd as mvq
    from multimodal_vqvae_compression_audio_tactile_amd import synth
    sd, net = model
    # empty batch
    y0 = net.forward_eval(torch.zeros(0, 1, 24000, device=dev), torch.zeros(0, 1, 24000, device=dev))
    assert y0.shape[0] == 0
    # length not a multiple of the hop (whole-file mode): Tl = floor-type conv arithmetic, must equal the oracle
    T = 320 * 20 + 137
    a, t = synth.audio_segments(1, seed=8, T=T), synth.tactile_segments(1, seed=8, T=T)
    want = orc.proposed_encode_latents(_np(sd), a.numpy(), t.numpy())
    got = net.encode_latents(a.to(dev), t.to(dev))
    assert got.shape == want.shape and np.array_equal(got.cpu().numpy(), want)
    # shorter than one token: the encoder yields zero tokens, the decoder an empty waveform
    z = net.T_ENC(torch.zeros(2, 1, 100, device=dev))
    assert z.shape[0] == 2 and z.shape[1] == 1024 and z.shape[2] == orc.dac_encoder(_np(sd), np.zeros((2, 1, 100), np.float32), prefix="T_ENC.").shape[2]
    # long clip (4 s) in one shot: finite, right length, prefix-consistent with the 1-s result up to the receptive field
    a4, t4 = synth.audio_segments(1, seed=9, T=96000).to(dev), synth.tactile_segments(1, seed=9, T=96000).to(dev)
    y4 = net.forward_eval(a4, t4)
    assert y4.shape == (1, 1, 96000 - 8) and torch.isfinite(y4).all()
    # non-finite input samples do not crash; callers sanitise with nan_to_num (Training/...5.py:324)
    bad = t4[..., :24000].clone(); bad[0, 0, 1000] = float("nan")
    yb = net.forward_eval(a4[..., :24000], bad)
    assert yb.shape == (1, 1, 23992)
    # DAC baseline mode at full size, every n_q the reference evaluates
    mdl = mvq.DAC(); mdl.load_state_dict(synth.dac_state(7), strict=True); mdl = mdl.to(dev).eval()
    x = synth.tactile_segments(2, seed=12).to(dev)
    prev = None
    for n_q in (1, 2, 3, 4, 8):
        z, codes, lat, _, _ = mdl.encode(x, n_quantizers=n_q)
        assert codes.shape == (2, n_q, 75) and int(codes.min()) >= 0 and int(codes.max()) < 1024
        if prev is not None:
            assert torch.equal(codes[:, :prev.shape[1]], prev)                 # residual stages are nested
        prev = codes
        assert mdl.decode(z).shape == (2, 1, 23992)


def test_dac_rate_sweep_full_size_against_oracle(orc, dev):
    """``DAC_NQ_LIST = [1, 4, 8, 16, 32]`` of Evaluation/compare_dacvsproposal_3.5_eval.py:75, the rate-scalable DAC-24 kHz
    baseline of its eval_dac24_ratescalable (``mdl.encode(t, n_quantizers=n_q)`` / ``mdl.decode(z)``, ...:444-446): one full
    24 000-sample segment per sweep point -- codes, latents, z and the decoded waveform bit-exact against the oracle; the
    encoder runs once in the oracle (it does not depend on n_q) and the decode is checked at the 32-book point."""
    import multimodal_vqvae_compression_audio_tactile_amd as mvq
    from multimodal_vqvae_compression_audio_tactile_amd import synth
    sd = synth.dac_state(7)
    sdn = _np(sd)
    mdl = mvq.DAC(); mdl.load_state_dict(sd, strict=True); mdl = mdl.to(dev).eval()
    x = synth.tactile_segments(1, seed=33)
    ze = orc.dac_encoder(sdn, x.numpy(), prefix="encoder.")
    assert ze.shape == (1, 1024, 75)
    for n_q in (1, 4, 8, 16, 32):
        want_zq, want_codes, want_lat, _, _ = orc.dac_quantizer(sdn, ze, n_q, prefix="quantizer.")
        z, codes, lat, _, _ = mdl.encode(x.to(dev), n_quantizers=n_q)
        assert codes.shape == (1, n_q, 75) and lat.shape == (1, 8 * n_q, 75)
        assert np.array_equal(codes.cpu().numpy(), want_codes)
        assert np.array_equal(lat.cpu().numpy(), want_lat)
        assert np.array_equal(z.cpu().numpy(), want_zq)
        if n_q == 32:                                             # (one oracle decode: 83 GFLOP of scalar fma chains on the host)
            y = mdl.decode(z)
            want_y = orc.dac_decoder(sdn, want_zq, prefix="decoder.")
            assert y.shape == (1, 1, 23992) and np.array_equal(y.cpu().numpy(), want_y)


def test_hipgraph_replay_is_bit_equal_to_eager(model, dev):
    """graphs.GraphedCall (one hipGraph per call, B = 1 latency regime): a replay -- also on NEW input values copied into the
    captured buffers -- returns exactly what the eager launch sequence returns, for encode_latents (two-stream fork inside
    the capture) and for the decoder."""
    from multimodal_vqvae_compression_audio_tactile_amd import synth
    from multimodal_vqvae_compression_audio_tactile_amd.graphs import GraphedCall
    sd, net = model
    a, t = synth.audio_segments(1, seed=33).to(dev), synth.tactile_segments(1, seed=33).to(dev)
    z = net.encode_latents(a, t)
    y = net.T_DEC(z)
    g_enc = GraphedCall(lambda aa, tt: net.encode_latents(aa, tt), a, t)
    g_dec = GraphedCall(lambda zz: net.T_DEC(zz), z)
    assert torch.equal(g_enc(a, t), z) and torch.equal(g_dec(z), y)
    a2, t2 = synth.audio_segments(1, seed=34).to(dev), synth.tactile_segments(1, seed=34).to(dev)
    z2 = net.encode_latents(a2, t2)
    assert not torch.equal(z2, z)
    assert torch.equal(g_enc(a2, t2), z2)
    assert torch.equal(g_dec(z2), net.T_DEC(z2))
    assert torch.equal(g_enc(a, t), z)                       # and back
