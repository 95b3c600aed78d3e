"""The persistent AR-loop kernel (csrc/ar_fused.hip, mvq_ar_latents_f32) against the launch-per-stage loop it replaces at small
batch sizes: same bits for z_run, the tokens the EMA update sees and the code indices, over the shapes the reference meets
(Training/compare_dacvsproposal_5.py:302-320 == Evaluation/dac_vcpwq_proposed6_latency.py:461-477), and end to end against the C
oracle.  The fused loop is an opt-in (proposed.py: AR_FUSED_MAX_BATCH, default 0): the tests switch it on."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nets(dev):
    from multimodal_vqvae_compression_audio_tactile_amd import build_proposed, synth
    out = {}
    for books, K in ((8, 512), (3, 128)):
        sd = synth.proposed_model_state(21 + books, rvq_books=books, rvq_embed=K)
        out[(books, K)] = build_proposed(sd, rvq_books=books, rvq_embed=K, device=dev)
    return out


def _latents(B, Tlat, Ta, seed, dev):
    g = torch.Generator().manual_seed(seed)
    zt = torch.randn(B, 1024, Tlat, generator=g) * 0.7
    qa = torch.randn(B, 1024, Ta, generator=g) * 0.7
    return qa.to(dev), zt.to(dev)


def _both(net, qa, zt, form="fused", **kw):
    """(one-call form, Python loop) of the same call.  form: "fused" = the persistent kernel (an opt-in), "staged" = one host call of
    stand-alone launches (the default up to 8 segments)."""
    keep = net.AR_FUSED_MAX_BATCH, net.AR_STAGED_MAX_BATCH
    try:
        net.AR_FUSED_MAX_BATCH, net.AR_STAGED_MAX_BATCH = (8, 0) if form == "fused" else (0, 8)
        assert net._ar_one_call_mode(zt, net.vq.stacked()) == form
        one = net._ar_latents(qa, zt, **kw)
        net.AR_FUSED_MAX_BATCH, net.AR_STAGED_MAX_BATCH = 0, 0
        assert net._ar_one_call_mode(zt, net.vq.stacked()) is None
        plain = net._ar_latents(qa, zt, **kw)
    finally:
        net.AR_FUSED_MAX_BATCH, net.AR_STAGED_MAX_BATCH = keep
    return one, plain


@pytest.mark.parametrize("form", ["fused", "staged"])
@pytest.mark.parametrize("B,Tlat,Ta", [(1, 75, 75), (6, 75, 75), (2, 35, 20), (3, 16, 16), (1, 5, 5), (2, 33, 40), (2, 20, 0), (8, 40, 40)])
def test_fused_loop_equals_launch_per_stage(B, Tlat, Ta, form, nets, dev):
    net = nets[(8, 512)]
    qa, zt = _latents(B, Tlat, Ta, 100 + B + Tlat, dev)
    fused, plain = _both(net, qa, zt, form, want_tokens=True, want_indices=True)
    for f, p, name in zip(fused, plain, ("z_run", "r_tokens", "indices")):
        assert f.shape == p.shape and f.dtype == p.dtype, name
        assert torch.equal(f, p), f"{name}: {int((f != p).sum())} of {f.numel()} elements differ"
    assert bool(torch.isfinite(fused[0]).all())


@pytest.mark.parametrize("form", ["fused", "staged"])
@pytest.mark.parametrize("books,K,use", [(8, 512, 3), (8, 512, 0), (3, 128, None), (3, 128, 2)])
def test_fused_loop_book_counts_and_sizes(books, K, use, form, nets, dev):
    net = nets[(books, K)]
    qa, zt = _latents(2, 40, 40, 7, dev)
    fused, plain = _both(net, qa, zt, form, books_use=use, want_indices=True)
    assert torch.equal(fused[0], plain[0])
    assert fused[1] is None and plain[1] is None
    assert torch.equal(fused[2], plain[2]) and fused[2].shape[0] == (books if use is None else use)


@pytest.mark.parametrize("form", ["fused", "staged"])
def test_fused_loop_tactile_only(form, nets, dev):
    net = nets[(8, 512)]
    _, zt = _latents(4, 75, 1, 9, dev)
    fused, plain = _both(net, None, zt, form, tactile_only=True, want_tokens=True)
    assert torch.equal(fused[0], plain[0]) and torch.equal(fused[1], plain[1])


def test_fused_loop_end_to_end_against_the_oracle(orc, dev, monkeypatch):
    from multimodal_vqvae_compression_audio_tactile_amd import build_proposed, synth
    sd = synth.proposed_model_state(13, rvq_books=2, rvq_embed=128)
    net = build_proposed(sd, rvq_books=2, rvq_embed=128, device=dev)
    monkeypatch.setattr(net, "AR_FUSED_MAX_BATCH", 8)
    a = synth.audio_segments(1, seed=4, T=320 * 20)            # 20 audio tokens under 35 tactile ones: chunk 2 sees Tk = 4, chunk 3 Tk = 0
    t = synth.tactile_segments(1, seed=4, T=320 * 35)
    want = orc.proposed_encode_latents({k: v.numpy() for k, v in sd.items()}, a.numpy(), t.numpy())
    got = net.encode_latents(a.to(dev), t.to(dev))
    assert np.array_equal(got.cpu().numpy(), want)


def test_fused_loop_selection(nets, dev, monkeypatch):
    from multimodal_vqvae_compression_audio_tactile_amd import ops
    net = nets[(8, 512)]
    books = net.vq.stacked()
    _, zt = _latents(1, 16, 1, 3, dev)
    assert not net._ar_fused_wanted(zt, books)                 # off by default: measured slower than the per-stage launches
    assert net._ar_one_call_mode(zt, books) == "staged"        # the default at small batches: one host call, the same launches
    monkeypatch.setattr(net, "AR_FUSED_MAX_BATCH", 8)
    assert net._ar_fused_wanted(zt, books)
    big = torch.empty(net.AR_FUSED_MAX_BATCH + 1, 1024, 16, device=dev)
    assert not net._ar_fused_wanted(big, books) and net._ar_one_call_mode(big, books) is None
    with ops.arith("bf16x6"):                                  # the opt-in modes keep their own kernels
        assert net._ar_one_call_mode(zt, books) is None
    g = torch.cuda.CUDAGraph()                                 # a cooperative launch cannot be captured: capture takes the staged form
    qa, zt = _latents(1, 32, 32, 5, dev)
    want = net._ar_latents(qa, zt)[0]
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        net._ar_latents(qa, zt)                                # warm the caches outside the capture
        with torch.cuda.graph(g, stream=s):
            got = net._ar_latents(qa, zt)[0]
    torch.cuda.current_stream().wait_stream(s)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(got, want)


def test_fused_loop_rejects_what_it_does_not_cover(dev):
    import ctypes
    from multimodal_vqvae_compression_audio_tactile_amd import _lib
    a = _lib.ArArgs()
    a.batch, a.t_lat, a.c_lat, a.c_ff, a.code_dim, a.heads, a.chunk, a.rvq_k = 1, 16, 512, 2048, 96, 8, 16, 512
    assert _lib.lib().mvq_ar_latents_f32(ctypes.byref(a), None, 0, None) == -2          # MVQ_EUNSUPPORTED: not the reference's widths
    a.c_lat = 1024
    assert _lib.lib().mvq_ar_latents_f32(ctypes.byref(a), None, 0, None) == -1          # MVQ_EINVAL: null tensors
    assert _lib.lib().mvq_ar_workspace_bytes(0, 16) == 0
    assert np.int64(_lib.lib().mvq_ar_workspace_bytes(6, 75)) > 0
