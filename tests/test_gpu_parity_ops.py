"""-m gpu: every C-ABI primitive of the HIP path against the CPU oracle, BIT-EXACT (fp32 results compared
with array_equal, indices exact).  Sizes are small enough for the oracle to finish in seconds."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rng(seed):
    return np.random.default_rng(seed)


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


CONV_CASES = [
    # (B, Cin, Tin, Cout, ks, stride, dil, pad, alpha_in, residual, alpha_out, tanh)
    (2, 64, 700, 64, 7, 1, 1, 3, True, False, True, False),      # BM=64 tile, RU conv7 d1
    (1, 128, 333, 128, 7, 1, 3, 9, True, False, True, False),    # BM=128, d3
    (2, 96, 530, 96, 7, 1, 9, 27, True, False, True, False),     # BM=96, d9
    (1, 192, 300, 192, 7, 1, 9, 27, True, False, False, False),
    (1, 256, 257, 256, 7, 1, 1, 3, False, False, False, False),
    (2, 64, 700, 64, 1, 1, 1, 0, False, True, False, False),     # RU conv1 + residual
    (1, 128, 90, 128, 1, 1, 1, 0, False, True, True, False),     # + fused snake (last RU of a block)
    (1, 96, 400, 96, 1, 1, 1, 0, False, True, True, False),
    (3, 1024, 16, 1024, 1, 1, 1, 0, False, False, False, False), # predictor linear on 16-token chunks
    (2, 1024, 11, 2048, 1, 1, 1, 0, False, False, False, False),
    (2, 1024, 16, 96, 1, 1, 1, 0, False, False, False, False),   # proj_down
    (2, 96, 16, 1024, 1, 1, 1, 0, False, True, False, False),    # proj_up + z_pred
    (1, 1024, 75, 1024, 3, 1, 1, 1, False, False, False, False), # encoder tail k3
    (1, 64, 1000, 128, 4, 2, 1, 1, False, False, False, False),  # strided s=2
    (1, 128, 800, 256, 8, 4, 1, 2, False, False, True, False),   # s=4
    (1, 256, 615, 512, 10, 5, 1, 3, False, False, False, False), # s=5
    (1, 512, 264, 1024, 16, 8, 1, 4, False, False, True, False), # s=8
    (2, 1, 900, 64, 7, 1, 1, 3, False, False, False, False),     # encoder input conv (direct)
    (2, 96, 900, 1, 7, 1, 1, 3, False, False, False, True),      # decoder output conv + tanh (direct)
    (1, 1024, 40, 8, 1, 1, 1, 0, False, False, False, False),    # tiny Cout (direct)
    (1, 40, 100, 24, 5, 2, 2, 3, True, False, True, False),      # odd shape -> direct fallback
    (2, 1024, 75, 1536, 7, 1, 1, 3, False, False, True, False),  # decoder input conv: narrow 128x96 tile
    (2, 512, 600, 1024, 16, 8, 1, 4, False, False, True, False), # last encoder down-sampling conv (Tout = 75)
    # latency regime (B = 1, short T): 64 x 64 tiles
    (1, 512, 600, 512, 7, 1, 3, 9, True, False, True, False),
    (1, 768, 600, 768, 1, 1, 1, 0, False, True, False, False),
    (1, 1024, 16, 2048, 1, 1, 1, 0, False, False, False, False),
    (1, 256, 3000, 512, 10, 5, 1, 3, False, False, True, False),
    (1, 128, 1200, 256, 8, 4, 1, 2, False, False, False, False),
    (1, 64, 800, 128, 4, 2, 1, 1, False, False, False, False),
    # throughput regime (grid >= 160 big tiles): the 128/96-row tiles, two-pass epilogue, XCD-aware row-fast order,
    # three blocks per CU -- odd lengths take the scalar staging / store paths
    (6, 256, 3503, 256, 7, 1, 3, 9, False, False, True, False),   # 128x128 k7, unaligned rows, Snake epilogue
    (5, 256, 3200, 256, 7, 1, 9, 27, True, True, False, False),   # aligned rows, Snake on load, residual
    (4, 384, 2999, 384, 1, 1, 1, 0, False, True, False, False),   # k1, 3 row tiles in row-fast order, unaligned (dec.b1)
    (4, 192, 2600, 200, 1, 1, 1, 0, False, True, True, False),    # partial last row tile (200 rows)
    (4, 192, 6000, 192, 7, 1, 9, 27, False, False, True, False),  # 96-row tile
    (4, 192, 6001, 192, 1, 1, 1, 0, False, True, False, False),   # 96-row k1, unaligned
    (4, 128, 12000, 256, 8, 4, 1, 2, False, False, True, False),  # strided s=4, 128-row tile
    (3, 64, 20000, 128, 4, 2, 1, 1, False, False, False, False),  # strided s=2
    (4, 256, 9000, 512, 10, 5, 1, 3, False, False, False, False), # strided s=5
    (8, 1024, 75, 1024, 3, 1, 1, 1, False, False, False, False),  # k3 at the latent rate, 128x96 tile
    # column split (conv_tail_width): T = 600 -> four 128-column tiles + one 96-column tail launch; T = 3000-ish -> 64-column tail
    (16, 256, 600, 256, 7, 1, 9, 27, False, False, True, False),
    (8, 512, 600, 512, 7, 1, 1, 3, True, True, False, False),
    (6, 768, 600, 768, 1, 1, 1, 0, False, True, True, False),
    (4, 256, 3000, 256, 7, 1, 3, 9, False, True, True, False),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[f"c{i}" for i in range(len(CONV_CASES))])
def test_conv1d_bit_exact(case, orc, dev):
    from multimodal_vqvae_compression_audio_tactile_amd import ops
    B, Cin, Tin, Cout, ks, stride, dil, pad, ai, res, ao, th = case
    r = _rng(hash(case) % (2 ** 31))
    x = r.standard_normal((B, Cin, Tin)).astype(np.float32)
    w = (r.standard_normal((Cout, Cin, ks)) / math.sqrt(Cin * ks)).astype(np.float32)
    b = (0.1 * r.standard_normal(Cout)).astype(np.float32)
    alpha_in = r.uniform(0.5, 1.5, Cin).astype(np.float32) if ai else None
    alpha_out = r.uniform(0.5, 1.5, Cout).astype(np.float32) if ao else None
    Tout = orc.conv1d_out_len(Tin, ks, stride, dil, pad)
    resid = r.standard_normal((B, Cout, Tout)).astype(np.float32) if res else None
    want = orc.conv1d(x, w, b, stride, dil, pad, alpha_in, resid, alpha_out, th)
    wp = ops.pack_conv1d(_t(w, dev))
    got = ops.conv1d(_t(x, dev), wp, Cout, ks, bias=_t(b, dev), stride=stride, dil=dil, pad=pad,
                     alpha_in=None if alpha_in is None else _t(alpha_in, dev),
                     residual=None if resid is None else _t(resid, dev),
                     alpha_out=None if alpha_out is None else _t(alpha_out, dev), tanh=th)
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    assert got.shape == want.shape
    assert np.array_equal(got, want), f"max abs diff {np.abs(got - want).max()}"


CONVTR_CASES = [
    # (B, Cin, Tin, Cout, stride, alpha_in, alpha_out)
    (1, 1536, 20, 768, 8, True, False),
    (2, 768, 77, 384, 5, True, False),
    (1, 384, 130, 192, 4, True, True),
    (2, 192, 301, 96, 2, True, False),
    (1, 48, 33, 20, 3, True, False),     # direct fallback
    (2, 1536, 75, 768, 8, False, False), # dec.b0 up-sampling at the latent rate (narrow tile), pre-snaked input
    (1, 256, 40, 128, 2, True, True),    # stride 2 with a 128-row tile
    (1, 768, 600, 384, 5, False, True),  # latency regime: 64-row tile, 5 phases (direct store)
    (1, 384, 299, 192, 4, True, False),  # latency regime: 64-row tile, pixel-shuffle through LDS
    # throughput regime: 128/96-row tiles, two-pass pixel-shuffle epilogue
    (8, 384, 701, 192, 4, False, True),
    (6, 768, 333, 384, 5, False, False),  # 5 phases: direct store
    (8, 192, 1500, 96, 2, True, False),   # 96-row tile
    (24, 1536, 75, 768, 8, False, True),  # latent rate, 128x96 tile
]


@pytest.mark.parametrize("case", CONVTR_CASES, ids=[f"t{i}" for i in range(len(CONVTR_CASES))])
def test_conv_transpose1d_bit_exact(case, orc, dev):
    from multimodal_vqvae_compression_audio_tactile_amd import ops
    B, Cin, Tin, Cout, s, ai, ao = case
    r = _rng(hash(case) % (2 ** 31))
    pad = math.ceil(s / 2)
    x = r.standard_normal((B, Cin, Tin)).astype(np.float32)
    w = (r.standard_normal((Cin, Cout, 2 * s)) / math.sqrt(Cin * 2)).astype(np.float32)
    b = (0.1 * r.standard_normal(Cout)).astype(np.float32)
    alpha_in = r.uniform(0.5, 1.5, Cin).astype(np.float32) if ai else None
    alpha_out = r.uniform(0.5, 1.5, Cout).astype(np.float32) if ao else None
    want = orc.conv_transpose1d(x, w, b, s, pad, alpha_in, alpha_out)
    wp = ops.pack_conv_transpose1d(_t(w, dev), s)
    got = ops.conv_transpose1d(_t(x, dev), wp, Cout, s, pad, bias=_t(b, dev),
                               alpha_in=None if alpha_in is None else _t(alpha_in, dev),
                               alpha_out=None if alpha_out is None else _t(alpha_out, dev))
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    assert got.shape == want.shape
    assert np.array_equal(got, want), f"max abs diff {np.abs(got - want).max()}"


def test_weight_norm_bit_exact(orc, dev):
    from multimodal_vqvae_compression_audio_tactile_amd import ops
    r = _rng(3)
    for shape in [(64, 1, 7), (768, 768, 7), (1536, 768, 16), (8, 1024, 1)]:
        v = r.standard_normal(shape).astype(np.float32)
        g = r.uniform(0.5, 2.0, (shape[0], 1, 1)).astype(np.float32)
        got = ops.weight_norm(_t(v, dev), _t(g, dev)).cpu().numpy()
        assert np.array_equal(got, orc.weight_norm(v, g))


@pytest.mark.parametrize("B,T,nb,K,use", [(6, 16, 3, 128, None), (2, 11, 10, 128, 7), (3, 16, 8, 512, None),
                                         (1, 75, 4, 256, 2), (5, 16, 1, 512, None), (2, 1, 2, 64, None),
                                         (20, 16, 3, 512, None), (9, 75, 2, 256, 1),   # 257..1023 tokens: 8 tokens per block (<= 256: one block per token)
                                         (40, 103, 2, 512, None),    # >= 1024 tokens: the MFMA form of the search
                                         (64, 16, 8, 512, None), (70, 16, 3, 128, 2), (67, 16, 10, 256, None), (14, 75, 2, 512, None)])
def test_rvq_ema_forward_bit_exact(B, T, nb, K, use, orc, dev):
    from multimodal_vqvae_compression_audio_tactile_amd import ops
    r = _rng(B * 1000 + K + nb)
    D = 96
    z = (0.3 * r.standard_normal((B, D, T))).astype(np.float32)
    books = np.stack([(0.5 ** i) * r.standard_normal((K, D)).astype(np.float32) / math.sqrt(D) for i in range(nb)])
    want_q, want_idx = orc.rvq_ema_forward(z, list(books), use)
    got_q, got_idx = ops.rvq_ema_forward(_t(z, dev), _t(books, dev), use, return_indices=True)
    assert np.array_equal(got_idx.cpu().numpy(), want_idx.astype(np.int64))
    assert np.array_equal(got_q.cpu().numpy(), want_q)


def test_rvq_ema_forward_ties_pick_lowest_index(orc, dev):
    from multimodal_vqvae_compression_audio_tactile_amd import ops
    r = _rng(11)
    D, K = 96, 512
    book = r.standard_normal((K, D)).astype(np.float32) / math.sqrt(D)
    book[300] = book[17]; book[499] = book[17]; book[40] = book[17]     # exact duplicates
    for T in (5, 300, 1100):                                             # token form / 8-token scalar form / MFMA form
        z = np.repeat(book[17][None, :, None], T, axis=2).astype(np.float32)   # query == that code
        q, idx = ops.rvq_ema_forward(_t(z, dev), _t(book[None], dev), return_indices=True)
        _, want = orc.rvq_ema_forward(z, [book])
        assert np.array_equal(idx.cpu().numpy(), want.astype(np.int64))
        assert (idx.cpu().numpy() == 17).all()


@pytest.mark.parametrize("B,T,nb,K", [(6, 75, 3, 128), (2, 75, 8, 512), (16, 75, 2, 512)])
def test_rvq_ema_step_bit_exact(B, T, nb, K, orc, dev):
    from multimodal_vqvae_compression_audio_tactile_amd import ops
    r = _rng(77 + K)
    D = 96
    z = (0.3 * r.standard_normal((B, D, T))).astype(np.float32)
    books = np.stack([r.standard_normal((K, D)).astype(np.float32) / math.sqrt(D) for _ in range(nb)])
    want, _ = orc.rvq_ema_step(z, list(books), 0.99)
    bt = _t(books.copy(), dev)
    ops.rvq_ema_step_(_t(z, dev), bt, 0.99)
    assert np.array_equal(bt.cpu().numpy(), want)
    assert not np.array_equal(want, books)


def test_rvq_ema_step_scales_bit_exact(orc, dev):
    """The EMA update at the size every rank sees in the 8-GPU training config (ema_step_all_ranks: 8 x 256 segments x 75 =
    153 600 gathered tokens, 8 books x K = 512): stable counting sort + per-code in-order sums == the sequential loop of
    Training/compare_dacvsproposal_5.py:266-277 (oracle), bit for bit.  Also: a code that owns MANY tokens (long chain), a
    1025-token case (segment boundary of the sort) and codes that own none (must not move)."""
    from multimodal_vqvae_compression_audio_tactile_amd import ops
    D = 96
    for (B, T, nb, K, spread) in [(2048, 75, 8, 512, 0.3), (41, 25, 2, 128, 0.02), (1, 1025, 1, 256, 0.3)]:
        r = _rng(B + K)
        z = (spread * r.standard_normal((B, D, T))).astype(np.float32)        # spread 0.02: most tokens fall onto few codes
        books = np.stack([r.standard_normal((K, D)).astype(np.float32) / math.sqrt(D) for _ in range(nb)])
        want, idx = orc.rvq_ema_step(z, list(books), 0.99)
        bt = _t(books.copy(), dev)
        ops.rvq_ema_step_(_t(z, dev), bt, 0.99)
        got = bt.cpu().numpy()
        assert np.array_equal(got, want), (B, T, nb, K)
        used = np.stack([np.bincount(idx[i], minlength=K) > 0 for i in range(nb)])
        assert np.array_equal((got != books).any(axis=2), used)                 # exactly the used codes moved
        if spread < 0.1:
            assert np.bincount(idx[0], minlength=K).max() > B * T // 8          # a long in-order chain was exercised


@pytest.mark.parametrize("B,T,nq", [(2, 75, 32), (1, 9, 8), (3, 16, 1)])
def test_dac_rvq_bit_exact(B, T, nq, orc, dev):
    from multimodal_vqvae_compression_audio_tactile_amd import ops, synth
    sd = {k: v.numpy() for k, v in synth.quantizer_state(5, n_codebooks=nq).items()}
    r = _rng(B + T)
    z = r.standard_normal((B, 1024, T)).astype(np.float32)
    want_zq, want_codes, want_lat, _, _ = orc.dac_quantizer(sd, z, nq)
    in_w = np.stack([orc.weight_norm(sd[f"quantizers.{i}.in_proj.weight_v"], sd[f"quantizers.{i}.in_proj.weight_g"]).reshape(8, 1024) for i in range(nq)])
    out_w = np.stack([orc.weight_norm(sd[f"quantizers.{i}.out_proj.weight_v"], sd[f"quantizers.{i}.out_proj.weight_g"]).reshape(1024, 8) for i in range(nq)])
    in_b = np.stack([sd[f"quantizers.{i}.in_proj.bias"] for i in range(nq)])
    out_b = np.stack([sd[f"quantizers.{i}.out_proj.bias"] for i in range(nq)])
    cb = np.stack([sd[f"quantizers.{i}.codebook.weight"] for i in range(nq)])
    zq, codes, lat = ops.dac_rvq(_t(z, dev), _t(in_w, dev), _t(in_b, dev), _t(cb, dev), _t(out_w, dev), _t(out_b, dev), nq)
    assert np.array_equal(codes.cpu().numpy(), want_codes)
    assert np.array_equal(lat.cpu().numpy(), want_lat)
    assert np.array_equal(zq.cpu().numpy(), want_zq)
    # with the prepared codebook (what the modules pass) a handful of tokens takes the latency form (128 threads per token)
    prep = ops.dac_rvq_prepare(_t(cb, dev))
    zq, codes, lat = ops.dac_rvq(_t(z, dev), _t(in_w, dev), _t(in_b, dev), _t(cb, dev), _t(out_w, dev), _t(out_b, dev), nq, prepared=prep)
    assert np.array_equal(codes.cpu().numpy(), want_codes)
    assert np.array_equal(lat.cpu().numpy(), want_lat)
    assert np.array_equal(zq.cpu().numpy(), want_zq)


def test_layernorm_attention_gelu_bit_exact(orc, dev):
    from multimodal_vqvae_compression_audio_tactile_amd import ops, synth
    r = _rng(5)
    C = 1024
    pe = synth.pos_table(C, 64).numpy()
    for (B, T) in [(3, 16), (2, 11), (1, 1)]:
        x = r.standard_normal((B, C, T)).astype(np.float32)
        g = r.uniform(0.5, 1.5, C).astype(np.float32); b = (0.1 * r.standard_normal(C)).astype(np.float32)
        want = orc.layernorm_c(x + pe[:T].T[None], g, b)
        got = ops.layernorm_c(_t(x, dev), _t(g, dev), _t(b, dev), pe=_t(pe, dev))
        assert np.array_equal(got.cpu().numpy(), want)
        want = orc.layernorm_c(x, g, b, do_tanh=True, post_scale=0.08)
        got = ops.layernorm_c(_t(x, dev), _t(g, dev), _t(b, dev), do_tanh=True, post_scale=0.08)
        assert np.array_equal(got.cpu().numpy(), want)
    for (B, Tq, Tk) in [(3, 16, 16), (2, 11, 11), (2, 16, 9), (1, 16, 0)]:
        q = r.standard_normal((B, C, Tq)).astype(np.float32)
        k = r.standard_normal((B, C, Tk)).astype(np.float32)
        v = r.standard_normal((B, C, Tk)).astype(np.float32)
        want = orc.attention(q, k, v, 8)
        got = ops.attention(_t(q, dev), _t(k, dev), _t(v, dev), 8)
        assert np.array_equal(got.cpu().numpy(), want)
    x = (3 * r.standard_normal(100000)).astype(np.float32)
    assert np.array_equal(ops.gelu(_t(x, dev)).cpu().numpy(), orc.gelu(x))


def test_ops_refuse_cpu_tensors():
    from multimodal_vqvae_compression_audio_tactile_amd import ops
    from multimodal_vqvae_compression_audio_tactile_amd._lib import MvqError
    with pytest.raises(MvqError):
        ops.gelu(torch.zeros(4))


RU_CASES = [(2, 64, 1000, 1, True), (1, 64, 517, 9, False), (2, 96, 700, 3, True), (1, 96, 333, 1, False),
            (2, 128, 600, 9, True), (1, 128, 257, 3, False), (1, 192, 300, 3, True), (1, 256, 75, 1, False),
            (3, 96, 11997, 9, True), (3, 128, 6001, 3, True), (4, 64, 9999, 1, False), (3, 96, 8000, 1, False)]


@pytest.mark.parametrize("case", RU_CASES, ids=[f"ru{i}" for i in range(len(RU_CASES))])
def test_residual_unit_bit_exact(case, orc, dev):
    """mvq_residual_unit_f32 (fused for C in {64,96,128}, two launches otherwise) == the two oracle convs."""
    from multimodal_vqvae_compression_audio_tactile_amd import ops
    B, C, T, dil, nxt = case
    r = _rng(C * 7 + T + dil)
    x = r.standard_normal((B, C, T)).astype(np.float32)
    w7 = (r.standard_normal((C, C, 7)) / math.sqrt(C * 7)).astype(np.float32)
    w1 = (r.standard_normal((C, C, 1)) / math.sqrt(C)).astype(np.float32)
    b7 = (0.1 * r.standard_normal(C)).astype(np.float32); b1 = (0.1 * r.standard_normal(C)).astype(np.float32)
    aa = r.uniform(0.5, 1.5, C).astype(np.float32); ab = r.uniform(0.5, 1.5, C).astype(np.float32)
    an = r.uniform(0.5, 1.5, C).astype(np.float32) if nxt else None
    h = orc.conv1d(x, w7, b7, dil=dil, pad=3 * dil, alpha_in=aa)
    want = orc.conv1d(h, w1, b1, alpha_in=ab, residual=x, alpha_out=an)
    got = ops.residual_unit(_t(x, dev), ops.pack_conv1d(_t(w7, dev)), _t(b7, dev), _t(aa, dev), _t(ab, dev),
                            ops.pack_conv1d(_t(w1, dev)), _t(b1, dev), dil,
                            alpha_next=None if an is None else _t(an, dev))
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy(), want), f"max abs diff {np.abs(got.cpu().numpy() - want).max()}"


def test_dual_output_and_presnaked_input(orc, dev):
    """y2 = snake(y_raw, alpha2) from conv / transposed-conv / residual-unit epilogues, and a residual unit fed with the
    pre-snaked input, all equal to the plain composition."""
    from multimodal_vqvae_compression_audio_tactile_amd import ops
    r = _rng(99)
    for (B, Cin, Tin, Cout, ks, stride, pad) in [(2, 128, 600, 256, 8, 4, 2), (1, 64, 333, 64, 1, 1, 0), (2, 1, 500, 64, 7, 1, 3)]:
        x = r.standard_normal((B, Cin, Tin)).astype(np.float32)
        w = (r.standard_normal((Cout, Cin, ks)) / math.sqrt(Cin * ks)).astype(np.float32)
        b = (0.1 * r.standard_normal(Cout)).astype(np.float32)
        a2 = r.uniform(0.5, 1.5, Cout).astype(np.float32)
        want = orc.conv1d(x, w, b, stride, 1, pad)
        y, y2 = ops.conv1d(_t(x, dev), ops.pack_conv1d(_t(w, dev)), Cout, ks, bias=_t(b, dev), stride=stride, pad=pad,
                           alpha_dual=_t(a2, dev))
        assert np.array_equal(y.cpu().numpy(), want)
        assert np.array_equal(y2.cpu().numpy(), orc.snake(want, a2))
    for (B, Cin, Tin, Cout, s) in [(1, 768, 77, 384, 5), (2, 384, 130, 192, 4), (1, 1536, 20, 768, 8)]:
        pad = math.ceil(s / 2)
        x = r.standard_normal((B, Cin, Tin)).astype(np.float32)
        w = (r.standard_normal((Cin, Cout, 2 * s)) / math.sqrt(Cin * 2)).astype(np.float32)
        b = (0.1 * r.standard_normal(Cout)).astype(np.float32)
        a2 = r.uniform(0.5, 1.5, Cout).astype(np.float32)
        want = orc.conv_transpose1d(x, w, b, s, pad)
        y, y2 = ops.conv_transpose1d(_t(x, dev), ops.pack_conv_transpose1d(_t(w, dev), s), Cout, s, pad, bias=_t(b, dev),
                                     alpha_dual=_t(a2, dev))
        assert np.array_equal(y.cpu().numpy(), want)
        assert np.array_equal(y2.cpu().numpy(), orc.snake(want, a2))
    for (B, C, T, dil) in [(1, 256, 300, 3), (2, 96, 400, 9)]:
        x = r.standard_normal((B, C, T)).astype(np.float32)
        w7 = (r.standard_normal((C, C, 7)) / math.sqrt(C * 7)).astype(np.float32)
        w1 = (r.standard_normal((C, C, 1)) / math.sqrt(C)).astype(np.float32)
        b7 = (0.1 * r.standard_normal(C)).astype(np.float32); b1 = (0.1 * r.standard_normal(C)).astype(np.float32)
        aa, ab, a2 = (r.uniform(0.5, 1.5, C).astype(np.float32) for _ in range(3))
        h = orc.conv1d(x, w7, b7, dil=dil, pad=3 * dil, alpha_in=aa)
        want = orc.conv1d(h, w1, b1, alpha_in=ab, residual=x)
        xs = ops.conv1d(_t(x, dev), ops.pack_conv1d(_t(np.eye(C, dtype=np.float32)[:, :, None], dev)), C, 1,
                        alpha_dual=_t(aa, dev))[1]                       # snake_a(x) via an identity conv's dual output
        assert np.array_equal(xs.cpu().numpy(), orc.snake(x, aa))
        y, y2 = ops.residual_unit(_t(x, dev), ops.pack_conv1d(_t(w7, dev)), _t(b7, dev), _t(aa, dev), _t(ab, dev),
                                  ops.pack_conv1d(_t(w1, dev)), _t(b1, dev), dil, x_snaked=xs, alpha_dual=_t(a2, dev))
        assert np.array_equal(y.cpu().numpy(), want)
        assert np.array_equal(y2.cpu().numpy(), orc.snake(want, a2))


def test_align_by_xcorr_bit_exact(orc, dev):
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    import golden_inputs as gi
    import multimodal_vqvae_compression_audio_tactile_amd as mvq
    from multimodal_vqvae_compression_audio_tactile_amd import ops
    g = np.load(Path(__file__).resolve().parent / "golden" / "g6_align.npz")
    for name, (T, shift, noise, seed) in gi.ALIGN_CASES.items():
        ref, est = gi.align_inputs(T, shift, noise, seed)
        want_r, want_e, want_s, want_corr = orc.align_by_xcorr(ref, est, 200)
        corr, best = ops.align_xcorr(_t(ref, dev), _t(est, dev), 200)
        assert int(best.item()) == want_s == int(g[f"{name}.shift"])
        assert np.array_equal(corr.cpu().numpy(), want_corr)
        r_a, e_a, s = mvq.align_by_xcorr(_t(ref, dev), _t(est, dev), 200)
        assert s == want_s and np.array_equal(r_a.cpu().numpy(), g[f"{name}.ref_a"]) and np.array_equal(e_a.cpu().numpy(), g[f"{name}.est_a"])


@pytest.mark.parametrize("B,C,T,ks,dil", [(5, 384, 2999, 7, 3), (5, 384, 2999, 1, 1), (2, 1024, 75, 7, 1), (6, 192, 1001, 7, 9)])
def test_zero_padded_rows_conv(B, C, T, ks, dil, orc, dev):
    """mvq_conv1d_padded_f32: rows rounded up to a multiple of 4 with a zero tail give bit-identical true columns and a zero
    tail again (so the next layer can consume them), with residual, Snake and dual output in the epilogue."""
    from multimodal_vqvae_compression_audio_tactile_amd import ops
    r = _rng(C + T + ks)
    Tp = (T + 3) // 4 * 4
    pad = (ks - 1) * dil // 2
    x = r.standard_normal((B, C, T)).astype(np.float32)
    w = (r.standard_normal((C, C, ks)) / math.sqrt(C * ks)).astype(np.float32)
    b = (0.1 * r.standard_normal(C)).astype(np.float32)
    res = r.standard_normal((B, C, T)).astype(np.float32)
    ao = r.uniform(0.5, 1.5, C).astype(np.float32); a2 = r.uniform(0.5, 1.5, C).astype(np.float32)
    want = orc.conv1d(x, w, b, 1, dil, pad, None, res, None, False)
    want_s = orc.snake(want, a2)
    xp = np.zeros((B, C, Tp), np.float32); xp[..., :T] = x
    rp = np.zeros((B, C, Tp), np.float32); rp[..., :T] = res
    y, y2 = ops.conv1d(_t(xp, dev), ops.pack_conv1d(_t(w, dev)), C, ks, bias=_t(b, dev), dil=dil, pad=pad,
                       residual=_t(rp, dev), alpha_dual=_t(a2, dev), tvalid=T)
    y, y2 = y.cpu().numpy(), y2.cpu().numpy()
    assert y.shape == (B, C, Tp)
    assert np.array_equal(y[..., :T], want) and np.array_equal(y2[..., :T], want_s)
    assert not y[..., T:].any() and not y2[..., T:].any()


@pytest.mark.parametrize("B,Cin,Tin,Cout,s", [(6, 768, 600, 384, 5), (24, 1536, 75, 768, 8), (8, 384, 2999, 192, 4)])
def test_zero_padded_rows_conv_transpose(B, Cin, Tin, Cout, s, orc, dev):
    """mvq_conv_transpose1d_padded_f32: (a) output rows padded up to a multiple of 4 with a zeroed tail, (b) input rows that
    carry a zero tail with only the true outputs produced."""
    from multimodal_vqvae_compression_audio_tactile_amd import ops
    r = _rng(Cin + Tin + s)
    pad = math.ceil(s / 2)
    x = r.standard_normal((B, Cin, Tin)).astype(np.float32)
    w = (r.standard_normal((Cin, Cout, 2 * s)) / math.sqrt(Cin * 2)).astype(np.float32)
    b = (0.1 * r.standard_normal(Cout)).astype(np.float32)
    want = orc.conv_transpose1d(x, w, b, s, pad, None, None)
    Tn = want.shape[-1]
    wp = ops.pack_conv_transpose1d(_t(w, dev), s)
    Tp = (Tn + 3) // 4 * 4
    if Tp != Tn and Tp <= Tn + pad:                                          # (a)
        y = ops.conv_transpose1d(_t(x, dev), wp, Cout, s, pad, bias=_t(b, dev), tout_rows=Tp, tvalid=Tn).cpu().numpy()
        assert y.shape[-1] == Tp and np.array_equal(y[..., :Tn], want) and not y[..., Tn:].any()
    Tip = (Tin + 3) // 4 * 4 if Tin % 4 else Tin + 4                          # (b) input with a zero tail
    xp = np.zeros((B, Cin, Tip), np.float32); xp[..., :Tin] = x
    y = ops.conv_transpose1d(_t(xp, dev), wp, Cout, s, pad, bias=_t(b, dev), tout_rows=Tn).cpu().numpy()
    assert y.shape[-1] == Tn and np.array_equal(y, want)


def test_profiler_entry_points(dev):
    """mvq_profile_begin / mvq_profile_end: one entry per kernel instantiation with its launch count and algorithmic FLOPs
    (the split launches of a T = 600 row report their own column shares), nothing recorded while off."""
    from multimodal_vqvae_compression_audio_tactile_amd import ops
    B, C, T = 32, 512, 600                                      # 32 x 4 row tiles: the tail launch can fill the chip, so the row is split
    x = torch.randn(B, C, T, device=dev)
    wp = ops.pack_conv1d(torch.randn(C, C, 7, device=dev) / math.sqrt(C * 7))
    ops.conv1d(x, wp, C, 7, dil=3, pad=9)                       # not profiled
    ops.profile_begin()
    for _ in range(3):
        ops.conv1d(x, wp, C, 7, dil=3, pad=9)
    prof = ops.profile_end()
    assert sum(v["launches"] for v in prof.values()) == 6, prof        # 3 x (four 128-column tiles + one 96-column tail launch)
    main = prof["conv1d_mfma_kernel<7, 1, 3, 4, 2, 2, 2, 2, 0>"]; tail = prof["conv1d_mfma_kernel<7, 1, 3, 4, 1, 3, 4, 1, 0>"]
    per_col = 2.0 * C * C * 7 * B
    assert main["launches"] == 3 and abs(main["flops"] - 3 * per_col * 512) < 1 and abs(tail["flops"] - 3 * per_col * 88) < 1
    assert main["seconds"] > 0 and tail["seconds"] > 0
    assert ops.profile_end() == {}                               # a second end without begin: empty, profiler off
    ops.conv1d(x, wp, C, 7, dil=3, pad=9)
    ops.profile_begin(); assert ops.profile_end() == {}
    # the reference's batch of 6: a tail launch of 6 x 4 blocks would walk the whole K chain on 24 of 256 CUs -> ONE launch
    ops.profile_begin()
    ops.conv1d(x[:6].contiguous(), wp, C, 7, dil=3, pad=9)
    small = ops.profile_end()
    assert len(small) == 1 and small[list(small)[0]]["launches"] == 1, small
    assert abs(small[list(small)[0]]["flops"] - 2.0 * C * C * 7 * 6 * 600) < 1


def test_dma_and_register_staging_agree(dev):
    """The LDS-DMA staged K loop (default for 16-byte rows) and the register-staged loop (MVQ_NO_DMA=1) are the same
    arithmetic: bit-equal outputs on a wide k7, a 1x1 with residual + dual Snake output, a strided and a transposed conv."""
    import os, subprocess, sys
    code = r'''
import hashlib, math, sys, torch
sys.path.insert(0, %r)
from multimodal_vqvae_compression_audio_tactile_amd import ops
g = torch.Generator().manual_seed(11)
dev = torch.device("cuda:0")
r = lambda *s: torch.randn(*s, generator=g).to(dev)
outs = []
x = r(6, 256, 600); w = r(256, 256, 7) / 42.0
outs.append(ops.conv1d(x, ops.pack_conv1d(w), 256, 7, bias=r(256), dil=9, pad=27, alpha_out=torch.rand(256, generator=g).to(dev) + 0.5))
w1 = r(256, 256, 1) / 16.0
outs += list(ops.conv1d(x, ops.pack_conv1d(w1), 256, 1, bias=r(256), residual=r(6, 256, 600), alpha_dual=torch.rand(256, generator=g).to(dev) + 0.5))
ws = r(512, 256, 10) / 50.0
outs.append(ops.conv1d(r(4, 256, 3000), ops.pack_conv1d(ws), 512, 10, stride=5, pad=3))
wt = r(768, 384, 10) / 40.0
outs.append(ops.conv_transpose1d(r(6, 768, 600), ops.pack_conv_transpose1d(wt, 5), 384, 5, 3, bias=r(384)))
torch.cuda.synchronize()
print(hashlib.sha256(b"".join(o.cpu().numpy().tobytes() for o in outs)).hexdigest())
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = []
    for no_dma in (False, True):
        env = {k: v for k, v in os.environ.items() if k != "MVQ_NO_DMA"}
        if no_dma:
            env["MVQ_NO_DMA"] = "1"
            env["MVQ_ALLOW_TIMING_BUILD"] = "1"          # an A/B knob in the environment: mvq_build_flags() != 0, loading is opt-in
        res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        digests.append(res.stdout.strip().splitlines()[-1])
    assert digests[0] == digests[1]


def test_conv1d_vpacked_equals_the_plain_conv(dev):
    """Virtually packed latent-rate rows (mvq_conv1d_vpacked_f32): the encoder's last strided conv (k 16, s 8, T 600 -> 75) and the
    k3 conv behind it, 10 items per GEMM row with remapped LDS-DMA sources / stores and no repacked copy, against the plain
    launches -- bit-equal on every valid column, zeros in the tail, ragged last row (batch not a multiple of 10) included."""
    from multimodal_vqvae_compression_audio_tactile_amd import ops
    g = torch.Generator().manual_seed(21)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    for B in (37, 10, 3):
        x = r(B, 512, 600)
        w = r(1024, 512, 16) / 90.0
        wp = ops.pack_conv1d(w)
        bias, al = r(1024), (torch.rand(1024, generator=g) + 0.5).to(dev)
        want = ops.conv1d(x, wp, 1024, 16, bias=bias, stride=8, pad=4, alpha_out=al)            # [B, 1024, 75]
        tout, rows, per_in = ops.vpacked_geometry(600, 600, 16, 8, 1, 4, follow_pad=1)
        assert (tout, rows, per_in) == (75, 76, 608)
        got = ops.conv1d_vpacked(x, wp, 1024, 16, 10, 600, rows, bias=bias, stride=8, pad=4, alpha_out=al)
        assert got.shape == (B, 1024, 76)
        assert torch.equal(got[..., :75], want) and not got[..., 75:].any()
        # the k3 conv on those rows (its padding = the zero tail of each item's row)
        w3 = r(1024, 1024, 3) / 55.0
        wp3 = ops.pack_conv1d(w3)
        b3 = r(1024)
        want3 = ops.conv1d(want, wp3, 1024, 3, bias=b3, pad=1)
        got3 = ops.conv1d_vpacked(got, wp3, 1024, 3, 10, 75, 76, bias=b3, pad=1)
        assert torch.equal(got3[..., :75], want3) and not got3[..., 75:].any()
    # shapes that do not qualify are refused, not mis-computed
    from multimodal_vqvae_compression_audio_tactile_amd._lib import MvqError
    with pytest.raises(MvqError):
        ops.conv1d_vpacked(r(4, 512, 602), wp, 1024, 16, 10, 602, 76, bias=bias, stride=8, pad=4)      # rows that are not 16-byte multiples
    with pytest.raises(MvqError):
        ops.conv1d_vpacked(r(4, 512, 600), wp, 1024, 16, 10, 600, 72, bias=bias, stride=8, pad=4)      # period too short for the outputs


def test_encoder_virtually_packed_tail_is_bit_equal(dev):
    """Encoder.forward with the latent-rate tail on virtually packed rows (batches >= 32) == the unpacked launch plan."""
    from multimodal_vqvae_compression_audio_tactile_amd import Encoder, synth
    enc = Encoder(); enc.load_state_dict(synth.encoder_state(5), strict=True); enc = enc.to(dev)
    x = synth.tactile_segments(37, seed=3).to(dev)
    assert Encoder.VPACKED
    y = enc(x)
    try:
        Encoder.VPACKED = False
        y0 = enc(x)
    finally:
        Encoder.VPACKED = True
    assert y.shape == y0.shape == (37, 1024, 75) and torch.equal(y, y0)
    # a length whose latent rows do not qualify falls back silently to the plain plan
    x2 = synth.tactile_segments(33, seed=4, T=24000 - 320 * 3 + 16).to(dev)
    assert enc(x2).shape[0] == 33


def test_dac_rvq_prepared_codebook_is_bit_identical(dev):
    """mvq_dac_rvq_prepare_f32 + mvq_dac_rvq_prepared_f32 (normalised codebooks computed once at load time) == the in-kernel
    normalisation: same codes, latents and z_q, bit for bit, also under per-item stage limits."""
    from multimodal_vqvae_compression_audio_tactile_amd import ops
    g = torch.Generator().manual_seed(9)
    B, C, T, nq, K, Dc = 5, 1024, 37, 6, 1024, 8
    z = torch.randn(B, C, T, generator=g).to(dev)
    in_w = (torch.randn(nq, Dc, C, generator=g) / 32).to(dev); in_b = (0.1 * torch.randn(nq, Dc, generator=g)).to(dev)
    cb = torch.randn(nq, K, Dc, generator=g).to(dev)
    cb[1, 7] = 0.0                                                               # a zero code: the 1e-12 clamp of F.normalize
    out_w = (torch.randn(nq, C, Dc, generator=g) / 3).to(dev); out_b = (0.1 * torch.randn(nq, C, generator=g)).to(dev)
    prep = ops.dac_rvq_prepare(cb)
    assert prep[0].shape == cb.shape and prep[1].shape == (nq, K) and float(prep[1][1, 7]) == 0.0
    lim = torch.tensor([6, 1, 3, 6, 2], dtype=torch.int32, device=dev)
    for nq_item in (None, lim):
        a = ops.dac_rvq(z, in_w, in_b, cb, out_w, out_b, nq, nq_item=nq_item)
        b = ops.dac_rvq(z, in_w, in_b, cb, out_w, out_b, nq, nq_item=nq_item, prepared=prep)
        for x, y in zip(a, b):
            assert torch.equal(x, y)


def test_torch_ops_call_the_same_kernels(dev):
    """torch.ops.mi355x_vqvae.* (last row) are the C-ABI entry points: bit-equal to the ops.* calls."""
    import math
    import multimodal_vqvae_compression_audio_tactile_amd.torch_ops  # noqa: F401  (registers)
    from multimodal_vqvae_compression_audio_tactile_amd import ops
    torch.manual_seed(11)
    o = torch.ops.mi355x_vqvae
    x = torch.randn(2, 64, 200, device=dev); w = torch.randn(128, 64, 4, device=dev) / math.sqrt(256)
    wp = ops.pack_conv1d(w); b = torch.randn(128, device=dev); al = torch.rand(64, device=dev) + 0.5
    assert torch.equal(o.conv1d_snake_f32(x, wp, b, al, None, None, 128, 4, 2, 1, 1),
                       ops.conv1d(x, wp, 128, 4, bias=b, stride=2, pad=1, alpha_in=al))
    wt = torch.randn(64, 32, 8, device=dev) / math.sqrt(128); wtp = ops.pack_conv_transpose1d(wt, 4)
    assert torch.equal(o.conv_transpose1d_snake_f32(x, wtp, None, al, None, 32, 4, 2), ops.conv_transpose1d(x, wtp, 32, 4, 2, alpha_in=al))
    z = torch.randn(5, 96, 16, device=dev); books = torch.randn(4, 128, 96, device=dev) / math.sqrt(96)
    q, idx = o.vq_rvq_search_f32(z, books, 3)
    q2, idx2 = ops.rvq_ema_forward(z, books, n_books_use=3, return_indices=True)
    assert torch.equal(q, q2) and torch.equal(idx, idx2)
    bk1, bk2 = books.clone(), books.clone()
    o.ema_update_f32(bk1, z, 0.99); ops.rvq_ema_step_(z, bk2, 0.99)
    assert torch.equal(bk1, bk2) and not torch.equal(bk1, books)
