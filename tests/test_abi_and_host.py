"""CPU: the C-ABI library loads and exports every symbol include/mvq.h declares; host-side logic that needs no GPU
(shape helpers, packed sizes, state-dict compatibility of the module mirror, loud failure without a device)."""
import ctypes
import re
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent


def _declared():
    txt = (ROOT / "include" / "mvq.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mvq_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from multimodal_vqvae_compression_audio_tactile_amd import _lib
    names = _declared()
    assert len(names) >= 20
    so = ctypes.CDLL(str(_lib.SO_PATH))
    missing = [n for n in names if not hasattr(so, n)]
    assert not missing, missing
    assert sorted(_lib.EXPORTS) == names, set(names) ^ set(_lib.EXPORTS)     # the ctypes table binds all of them
    assert _lib.lib().mvq_abi_version() == 3
    assert _lib.lib().mvq_build_flags() == 0 == _lib.build_flags()          # product build, no A/B knob in the environment


def test_host_only_entry_points():
    from multimodal_vqvae_compression_audio_tactile_amd import _lib, ops
    lib = _lib.lib()
    assert lib.mvq_conv1d_packed_floats(768, 768, 7) == 768 * 7 * 768
    assert lib.mvq_conv1d_packed_floats(96, 96, 1) == 96 * 96              # BM=96 tile: no padding
    assert lib.mvq_conv1d_packed_floats(1, 64, 7) == 7 * 64
    assert lib.mvq_conv1d_packed_floats(1024, 8, 1) == 1024 * 64           # tiny Cout padded to one 64-row tile
    assert lib.mvq_conv_transpose1d_packed_floats(1536, 768, 8) == 1536 * 2 * 768 * 8
    assert lib.mvq_residual_unit_scratch_floats(4, 96, 1000, 3) == 0       # fused
    assert lib.mvq_residual_unit_scratch_floats(4, 768, 600, 9) == 4 * 768 * 600
    import re
    def tile(name):          # <KS, S, D, CK, MT, NT, WM, WN, UPS> -> (KS, S, D, MT, NT, WM, WN, UPS): CK is a tuning knob
        v = [int(x) for x in re.findall(r"-?\d+", name.split("<")[1])]
        return tuple(v[:3] + v[4:])
    assert tile(ops.conv_kernel_name(768, 768, 7, 1, 9, tin=600)) == (7, 1, 9, 2, 2, 2, 2, 0)             # 128 x 128 tile
    assert ops.conv_kernel_name(768, 768, 7, 1, 9, tin=600, batch=1) == "conv1d_lat_kernel<7, 1, 9, 16>"   # latency regime: one wave per 16 x 16 tile (16x16x4 MFMA)
    assert tile(ops.conv_kernel_name(768, 768, 7, 1, 9, tin=1200, batch=1)) == (7, 1, 9, 1, 1, 2, 2, 0)   # ... too many tiles for that: 64 x 64
    assert ops.conv_kernel_name(1024, 1024, 1, tin=16, batch=6) == "conv1d_lat_kernel<1, 1, 1, 64>"        # a predictor GEMM over one AR chunk of six segments
    assert tile(ops.conv_kernel_name(768, 768, 1, tin=600, batch=1)) == (1, 1, 1, 1, 1, 2, 2, 0)          # many tiles, short chain: stays LDS-tiled
    assert tile(ops.conv_kernel_name(1024, 1536, 7, tin=75, batch=64)) == (7, 1, 1, 1, 3, 4, 1, 0)        # latent rate: 128 x 96
    assert tile(ops.conv_kernel_name(1536, 768, 16, 8, 1, True, tin=75)) == (2, 1, 1, 1, 3, 4, 1, 8)      # polyphase convT, 8 phases
    assert ops.conv_kernel_name(1, 64, 7) == "conv1d_cin1_kernel<7>" and ops.conv_kernel_name(40, 24, 5, 2, 2) == "conv1d_direct_kernel"
    assert ops.residual_unit_kernel_name(96, 3).startswith("residual_unit_kernel<3, ") and ops.residual_unit_kernel_name(96, 3).endswith(", 3, 1, 1, 4>")
    assert ops.conv1d_out_len(24000, 4, 2, 1, 1) == 12000 and ops.conv1d_out_len(600, 16, 8, 1, 4) == 75
    assert ops.conv1d_out_len(3, 7, 1, 9, 0) == 0
    assert lib.mvq_conv1d_f32(None, None, None, None, None, None, None, 1, -3, 5, 4, 7, 1, 1, 3, 0, None) == -1
    assert b"bad shape" in lib.mvq_last_error()


def test_module_mirror_loads_reference_shaped_state_dicts():
    import multimodal_vqvae_compression_audio_tactile_amd as mvq
    from multimodal_vqvae_compression_audio_tactile_amd import synth
    from oracle import dac24_torch as T
    sd = synth.proposed_model_state(3, rvq_books=3, rvq_embed=128)
    da, dt = mvq.DAC(), mvq.DAC()
    net = mvq.ProposedEval(da.encoder, da.quantizer, dt.encoder, dt.decoder, 1024, 3, 128)
    assert not any(p.requires_grad for m in (net.A_ENC, net.A_QUANT, net.T_ENC, net.T_DEC) for p in m.parameters())
    res = net.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    # identical key set / shapes as the torch restatement (== upstream parameter names)
    ref = T.ProposedEval(rvq_books=3, rvq_embed=128)
    a = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    b = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    assert a == b
    # get_n_books_and_bins probe of the reference (compare_dacvsproposal_5_eval.py:233-246)
    assert net.A_QUANT.n_q == 32 and net.A_QUANT.bins == 1024
    assert sum(p.numel() for p in da.encoder.parameters()) == 21521536
    dac_sd = synth.dac_state(5, n_codebooks=4)
    mvq.DAC(n_codebooks=4).load_state_dict(dac_sd, strict=True)


def test_proposed_wrapper_mirrors_the_3_5_eval_script():
    """Evaluation/compare_dacvsproposal_3.5_eval.py:374-411: ``ProposedWrapper(A_ENC, A_QUANT, T_ENC, T_DEC, c_lat)`` -- RVQ shape
    from that script's module constants (10 books x K = 128, ...:68-69), ``forward_eval(a, t, books_use)`` with a required
    ``books_use``; a checkpoint trained with 3 books (the script's comment at ...:68) loads strict=False as it does at ...:487."""
    import inspect
    import multimodal_vqvae_compression_audio_tactile_amd as mvq
    from multimodal_vqvae_compression_audio_tactile_amd import proposed, synth
    assert (proposed.RVQ_N_BOOKS_MAX, proposed.RVQ_EMBED, proposed.CODE_DIM, proposed.AR_CHUNK_TOK) == (10, 128, 96, 16)
    assert list(inspect.signature(mvq.ProposedWrapper.__init__).parameters) == ["self", "A_ENC", "A_QUANT", "T_ENC", "T_DEC", "c_lat"]
    fe = inspect.signature(mvq.ProposedWrapper.forward_eval).parameters
    assert list(fe) == ["self", "a_1T", "t_1T", "books_use"] and fe["books_use"].default is inspect.Parameter.empty
    da, dt = mvq.DAC(), mvq.DAC()
    net = mvq.ProposedWrapper(da.encoder, da.quantizer, dt.encoder, dt.decoder, c_lat=1024)
    assert len(net.vq.books) == 10 and tuple(net.vq.books[0].shape) == (128, 96)
    assert not any(p.requires_grad for m in (net.A_ENC, net.A_QUANT, net.T_ENC, net.T_DEC) for p in m.parameters())
    res = net.load_state_dict(synth.proposed_model_state(3, rvq_books=10, rvq_embed=128), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    res = net.load_state_dict(synth.proposed_model_state(3, rvq_books=3, rvq_embed=128), strict=False)     # "trained with 3"
    assert sorted(res.missing_keys) == [f"vq.books.{i}" for i in range(3, 10)] and not res.unexpected_keys
    ref = "/root/reference/Evaluation/compare_dacvsproposal_3.5_eval.py"
    import os
    if os.path.exists(ref):                                       # build container only: same key set as the reference's own class
        from oracle import dac24_torch as T, ref_import
        e35 = ref_import.eval35()
        rd, rt = T.DAC(), T.DAC()
        theirs = e35.ProposedWrapper(rd.encoder, rd.quantizer, rt.encoder, rt.decoder, c_lat=1024)
        a = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        b = {k: tuple(v.shape) for k, v in theirs.state_dict().items()}
        assert a == b


def test_no_cpu_fallback():
    import multimodal_vqvae_compression_audio_tactile_amd as mvq
    from multimodal_vqvae_compression_audio_tactile_amd import ops
    with pytest.raises(mvq.MvqError):
        ops.conv1d(torch.zeros(1, 64, 100), torch.zeros(64 * 7 * 64), 64, 7, pad=3)
    enc = mvq.Encoder()
    with pytest.raises(mvq.MvqError):
        enc(torch.zeros(1, 1, 3200))
    vq = mvq.ResidualVQEMA(96, 2, 128)
    with pytest.raises(mvq.MvqError):
        vq(torch.zeros(1, 96, 16))


def test_training_and_aux_rows_fail_loudly_without_a_device():
    """Rows f1-f4: losses, resampler, ST-SIM and the autograd Functions raise on CPU tensors (no fallback), and the new
    entry points validate their arguments before touching the device."""
    import multimodal_vqvae_compression_audio_tactile_amd as mvq
    from multimodal_vqvae_compression_audio_tactile_amd import _lib, train
    y, t = torch.zeros(1, 1, 2000), torch.zeros(1, 1, 2000)
    for fn in (mvq.safe_l1, mvq.MultiResSTFTLoss(), mvq.MelCosineLoss(), mvq.TrainingLoss(), mvq.stsim_batch):
        with pytest.raises(mvq.MvqError):
            fn(y, t)
    with pytest.raises(mvq.MvqError):
        mvq.Resample(3000, 24000)(torch.zeros(1, 300))
    with pytest.raises(mvq.MvqError):
        train.Gelu.apply(torch.zeros(1, 4, 8))
    with pytest.raises(mvq.MvqError):
        mvq.Resample(3000, 24000, resampling_method="sinc_interp_kaiser")
    with pytest.raises(mvq.MvqError):
        mvq.MultiResSTFTLoss(wins=(128, 256, 512))
    lib = _lib.lib()
    assert lib.mvq_resample_f32(None, None, None, 1, 100, 800, 1, 8, 7, 14, None) == -1          # ks != 2*width + orig
    assert lib.mvq_resample_f32(None, None, None, 1, 100, 801, 1, 8, 7, 15, None) == -1          # longer than ceil(new*L/orig)
    assert lib.mvq_stft_frames_f32(None, None, None, 1, 100, 512, 128, 1, 10, 0, None) == -1     # clip shorter than the padding
    assert lib.mvq_attention_bwd_f32(None, None, None, None, None, None, None, 1, 8, 128, 40, 16, 0, 0, 0, 0, None) == -1
    assert lib.mvq_layernorm_c_bwd_f32(None, None, None, None, None, None, None, None, 0, 1024, 16, 0, 0, 1e-5, None) == 0   # empty batch
    k, width, orig, new = mvq.resample.sinc_resample_kernel(44100, 24000)
    assert (orig, new, width) == (147, 80, 12) and tuple(k.shape) == (80, 171)
    assert abs(float(k.sum(dim=1).mean()) - 1.0) < 2e-3                                           # unit DC gain per phase


def test_reference_shaped_checkpoints_load(tmp_path):
    """What the reference loads: (1) ``dac.DAC.load(path)`` on an upstream-format file {"state_dict", "metadata.kwargs"}
    (Training/compare_dacvsproposal_5.py:329-331); (2) ``best.pth`` = {"model": net.state_dict(), "epoch", "hist", ...}
    (...5.py:423-435) whose keys are A_ENC.* / A_QUANT.* / T_ENC.* / T_DEC.* / predict.pos.pe / vq.books.*, loaded
    strict=False by the compression evals (Evaluation/compare_dacvsproposal_5_eval.py:432-433) and strict=True by the PLC
    eval (PLC/PLC1_eval.py:548).  Both weight-norm key spellings are accepted."""
    import multimodal_vqvae_compression_audio_tactile_amd as mvq
    from multimodal_vqvae_compression_audio_tactile_amd import synth
    # (1) upstream DAC file with constructor metadata
    dac_sd = synth.dac_state(5, n_codebooks=4)
    meta = {"kwargs": {"encoder_dim": 64, "encoder_rates": [2, 4, 5, 8], "decoder_dim": 1536, "decoder_rates": [8, 5, 4, 2],
                       "n_codebooks": 4, "codebook_size": 1024, "codebook_dim": 8, "quantizer_dropout": 1.0,
                       "sample_rate": 24000, "not_a_ctor_argument": 1}}
    torch.save({"state_dict": dac_sd, "metadata": meta}, tmp_path / "weights.pth")
    mdl = mvq.DAC.load(tmp_path / "weights.pth")
    assert mdl.quantizer.n_codebooks == 4 and mdl.quantizer.quantizer_dropout == 1.0 and mdl.sample_rate == 24000
    assert all(torch.equal(v, dac_sd[k]) for k, v in mdl.state_dict().items())
    assert mvq.DAC.load(tmp_path / "weights.pth", quantizer_dropout=0.0).quantizer.quantizer_dropout == 0.0     # kwargs override
    torch.save(dac_sd, tmp_path / "plain.pth")
    assert mvq.DAC.load(tmp_path / "plain.pth", n_codebooks=4).quantizer.n_codebooks == 4
    # (2) best.pth of the proposed model
    sd = synth.proposed_model_state(3, rvq_books=3, rvq_embed=128)
    assert {"A_ENC.block.0.weight_g", "A_QUANT.quantizers.0.codebook.weight", "T_ENC.block.0.weight_v",
            "T_DEC.model.1.block.1.weight_g", "predict.pos.pe", "vq.books.2", "scale"} <= set(sd)
    torch.save({"model": sd, "epoch": 12, "hist": {"train": [1.0]}, "rvq_books": 3, "rvq_embed": 128}, tmp_path / "best.pth")
    ckpt = torch.load(tmp_path / "best.pth", map_location="cpu")

    def fresh(cls=mvq.ProposedEval):
        da, dt = mvq.DAC(), mvq.DAC()
        return cls(da.encoder, da.quantizer, dt.encoder, dt.decoder, 1024, 3, 128)

    net = fresh()
    res = net.load_state_dict(ckpt["model"], strict=False)               # ...5_eval.py:433
    assert not res.missing_keys and not res.unexpected_keys
    fresh(mvq.AllPredAR).load_state_dict(ckpt["model"], strict=True)     # PLC1_eval.py:548 style
    assert all(torch.equal(v, sd[k]) for k, v in net.state_dict().items())
    # a checkpoint from a run with MORE books than the evaluated model loads non-strict and reports the extras
    sd5 = synth.proposed_model_state(3, rvq_books=5, rvq_embed=128)
    res = fresh().load_state_dict(sd5, strict=False)
    assert sorted(res.unexpected_keys) == ["vq.books.3", "vq.books.4"] and not res.missing_keys
    with pytest.raises(RuntimeError):
        fresh().load_state_dict(sd5, strict=True)
    # the parametrization spelling of weight norm
    new_style = {}
    for k, v in sd.items():
        k = k.replace(".weight_g", ".parametrizations.weight.original0").replace(".weight_v", ".parametrizations.weight.original1")
        new_style[k] = v
    assert any("parametrizations" in k for k in new_style)
    net2 = fresh()
    res = net2.load_state_dict(new_style, strict=True)
    assert all(torch.equal(v, sd[k]) for k, v in net2.state_dict().items())
    # compare_dacvsproposal_3.py's constructor (no sweep arguments): 10 books x 128
    da, dt = mvq.DAC(), mvq.DAC()
    n3 = mvq.AllPredAR3(da.encoder, da.quantizer, dt.encoder, dt.decoder, 1024)
    n3.load_state_dict(synth.proposed_model_state(3, rvq_books=10, rvq_embed=128), strict=True)


def test_vpacked_geometry_and_argument_checks():
    """Host side of the virtually packed rows (round 4): the geometry helper and the C entry point's argument validation
    (refused before any device access, so it runs without a GPU)."""
    from multimodal_vqvae_compression_audio_tactile_amd import _lib, ops
    # the encoder tail at 24 kHz: k 16, s 8, pad 4 over 600 columns -> 75 valid outputs in 76-float rows, input period 608
    assert ops.vpacked_geometry(600, 600, 16, 8, 1, 4, follow_pad=1) == (75, 76, 608)
    # the k3 conv behind it: rows -> rows
    assert ops.vpacked_geometry(76, 75, 3, 1, 1, 1, follow_pad=0) == (75, 76, 76)
    assert ops.vpacked_geometry(602, 602, 16, 8, 1, 4) is None                       # rows that are not 16-byte multiples
    # a longer clip: the period grows with the row, the gap always covers pad and the last output's overhang
    for tin in (600, 608, 1200, 2400):
        tout, rows, per_in = ops.vpacked_geometry(tin, tin, 16, 8, 1, 4, follow_pad=1)
        assert tout == ops.conv1d_out_len(tin, 16, 8, 1, 4) and rows % 4 == 0 and per_in == 8 * rows
        assert per_in - tin >= 4 and rows >= tout + 1
    lib = _lib.lib()
    bad = lambda *a: lib.mvq_conv1d_vpacked_f32(None, None, None, None, None, None, None, None, *a, None)
    #              batch cin tin_rows tin_valid cout ks stride dil pad act seg per_in tout_rows
    assert bad(4, 512, 600, 600, 1024, 16, 8, 1, 4, 0, 10, 600, 75) == -1 and b"per_in" in lib.mvq_last_error()      # tout_rows % 4
    assert bad(4, 512, 600, 600, 1024, 16, 8, 1, 4, 0, 10, 608, 72) == -1                                             # per_in != s * rows
    assert bad(4, 512, 600, 600, 1024, 16, 8, 1, 4, 0, 10, 576, 72) == -1                                             # rows longer than the period
    assert bad(4, 512, 76, 76, 1024, 3, 1, 1, 1, 0, 10, 76, 76) == -1                                                 # no zero gap for the padding
    assert bad(4, 512, 600, 600, 1024, 16, 8, 1, 4, 0, 10, 608, 76) == -1 and b"null tensor" in lib.mvq_last_error()  # geometry fine
    assert bad(0, 512, 600, 600, 1024, 16, 8, 1, 4, 0, 10, 608, 76) == 0                                              # empty batch


def test_arith_mode_switch_is_opt_in_and_validated():
    """The non-parity "bf16x6" mode (include/mvq.h) is off unless asked for, refuses unknown names, and only claims the wide units."""
    from multimodal_vqvae_compression_audio_tactile_amd import ops
    from multimodal_vqvae_compression_audio_tactile_amd._lib import MvqError
    assert ops.get_arith() == "f32"
    assert not any(ops.bf16x6_eligible(c) for c in (64, 96, 128, 192, 256, 768))
    with pytest.raises(MvqError):
        ops.set_arith("fp16")
    for mode in ("bf16x6", "f16x3"):
      ops.set_arith(mode)
      try:
        assert [c for c in (64, 96, 100, 128, 192, 256, 384, 512, 768) if ops.bf16x6_eligible(c)] == [96, 128, 192, 256, 384, 512, 768]
      finally:
        ops.set_arith("f32")
    assert ops.get_arith() == "f32"


def test_torch_ops_are_registered_with_shape_only_fakes():
    """SURVEY.md section 8b, last row: the hot-path entry points as torch.ops.mi355x_vqvae.* -- every operator is registered and
    its fake (shape-only) implementation answers on meta tensors with the shapes the HIP kernels produce (no GPU needed)."""
    import multimodal_vqvae_compression_audio_tactile_amd.torch_ops as T
    for name in T.REGISTERED:
        assert hasattr(torch.ops.mi355x_vqvae, name), name
    m = lambda *s, dtype=torch.float32: torch.empty(*s, device="meta", dtype=dtype)
    o = torch.ops.mi355x_vqvae
    assert o.conv1d_snake_f32(m(3, 64, 100), m(10), None, None, None, None, 128, 4, 2, 1, 1).shape == (3, 128, 50)
    assert o.conv1d_snake_f32(m(2, 256, 75), m(10), m(256), None, m(2, 256, 75), m(256), 256, 7, 1, 27, 9).shape == (2, 256, 75)
    assert o.conv_transpose1d_snake_f32(m(2, 1536, 75), m(10), m(768), m(1536), None, 768, 8, 4).shape == (2, 768, 600)
    assert o.residual_unit_f32(m(2, 64, 240), m(1), m(64), m(64), m(64), m(1), m(64), 3, None).shape == (2, 64, 240)
    q, idx = o.vq_rvq_search_f32(m(6, 96, 16), m(8, 512, 96), 3)
    assert q.shape == (6, 96, 16) and idx.shape == (3, 96) and idx.dtype == torch.int64
    zq, codes, lat = o.vq_cosine_rvq_f32(m(2, 1024, 75), m(32, 8, 1024), m(32, 8), m(32, 1024, 8), m(32, 1024, 8), m(32, 1024), 8)
    assert zq.shape == (2, 1024, 75) and codes.shape == (2, 8, 75) and codes.dtype == torch.int64 and lat.shape == (2, 64, 75)
    with pytest.raises(Exception):                       # the real implementations have no CPU path
        o.conv1d_snake_f32(torch.zeros(1, 16, 8), torch.zeros(4), None, None, None, None, 16, 1, 1, 0, 1)
    # whole stacks: a description-only handle (no weights, no device) answers the shape questions of the fakes
    from multimodal_vqvae_compression_audio_tactile_amd import ops
    enc = ops.Stack.encoder(64, (2, 4, 5, 8), 1024); dec = ops.Stack.decoder(1024, 1536, (8, 5, 4, 2))
    assert o.encoder_fwd(m(3, 1, 24000), enc.id).shape == (3, 1024, 75)
    assert o.decoder_fwd(m(3, 1024, 75), dec.id).shape == (3, 1, 23992)
    assert o.decoder_bwd_input(m(2, 1024, 75), m(2, 1, 23992), dec.id).shape == (2, 1024, 75)


def test_operators_trace_through_torch_compile():
    """The operator face is traceable: torch.compile(fullgraph=True) builds ONE graph through a torch.ops.mi355x_vqvae call (dynamo
    runs the shape-only fake; backend "eager" then executes it on meta tensors) -- no graph break, no fallback to Python."""
    import multimodal_vqvae_compression_audio_tactile_amd.torch_ops  # noqa: F401
    from multimodal_vqvae_compression_audio_tactile_amd import ops
    m = lambda *s: torch.empty(*s, device="meta")
    o = torch.ops.mi355x_vqvae
    f = torch.compile(lambda x, wp, b: o.conv1d_snake_f32(x, wp, b, None, None, None, 128, 4, 2, 1, 1) * 2.0, fullgraph=True, backend="eager")
    assert f(m(3, 64, 100), m(10), m(128)).shape == (3, 128, 50)
    enc = ops.Stack.encoder(64, (2, 4, 5, 8), 1024); dec = ops.Stack.decoder(1024, 1536, (8, 5, 4, 2))
    g = torch.compile(lambda x: o.decoder_fwd(o.encoder_fwd(x, enc.id), dec.id), fullgraph=True, backend="eager")
    assert g(m(2, 1, 24000)).shape == (2, 1, 23992)


def test_stack_handles_name_the_upstream_state_dict():
    """include/mvq.h "whole stacks": the parameter list a stack expects IS the upstream state-dict of the module it replaces (names,
    order of the module tree, shapes), so a checkpoint binds by name; lengths, workspace and blob queries need no device."""
    from multimodal_vqvae_compression_audio_tactile_amd import _lib, dac, ops
    for mod, st in ((dac.Encoder(), ops.Stack.encoder(64, dac.ENC_RATES, 1024)), (dac.Decoder(), ops.Stack.decoder(1024, 1536, dac.DEC_RATES))):
        sd = mod.state_dict()
        info = st.param_info()
        assert [n for n, _ in info] == list(sd.keys())
        for n, shp in info:
            assert int(torch.tensor(shp).prod()) == sd[n].numel(), n
    enc, dec = ops.Stack.encoder(64, dac.ENC_RATES, 1024), ops.Stack.decoder(1024, 1536, dac.DEC_RATES)
    assert [enc.out_len(t) for t in (24000, 320 * 18, 100, 0)] == [75, 18, 0, 0]                 # a clip shorter than a token: empty
    assert [dec.out_len(t) for t in (75, 18, 1, 0)] == [23992, 5752, 312, 0]
    assert ops.Stack.decoder(1024, 1536, dac.DEC_RATES, output_padding=True).out_len(75) == 24000
    L = _lib.lib()
    assert L.mvq_encoder_workspace_bytes(enc.handle, 1, 24000) < 64 << 20
    assert 4 << 30 < L.mvq_encoder_workspace_bytes(enc.handle, 256, 24000) < 8 << 30          # a few tensors of the widest stage, not one per layer
    assert L.mvq_decoder_saved_bytes(dec.handle, 256, 75) > 40 << 30                           # the training forward keeps every Snake input
    assert L.mvq_stack_weights_bytes(dec.handle) > 2 * 4 * 52_000_000                          # forward + input-gradient images
    with pytest.raises(_lib.MvqError):                                                         # no weights bound: the forward refuses
        _lib.check(L.mvq_encoder_fwd_f32(enc.handle, 1, 1, 1, 1024, 1, 24000, None), "mvq_encoder_fwd_f32")
    with pytest.raises(_lib.MvqError):
        ops.Stack.encoder(64, (), 1024)


def test_opt_in_mode_entry_points_validate_their_arguments():
    """The bf16x6 / f16x3 entry points (include/mvq.h) refuse bad shapes before any device access, accept empty batches, and size
    their images as documented."""
    from multimodal_vqvae_compression_audio_tactile_amd import _lib
    lib = _lib.lib()
    assert lib.mvq_bf16x3_split_bytes(3, 256, 100) == 3 * 256 * 100 * 6
    assert lib.mvq_conv1d_k7_bf16x3_packed_bytes(256, 256) == 256 * 256 * 7 * 6 and lib.mvq_conv1d_k7_f16x2_packed_bytes(192, 192) == 192 * 192 * 7 * 4
    assert lib.mvq_conv1d_k7_bf16x3_packed_bytes(160, 160) == 0 and lib.mvq_conv1d_k7_f16x2_packed_bytes(256, 250) == 0
    assert lib.mvq_bf16x3_split_f32(None, None, 2, 12, 16, None) == -1 and b"C % 8" in lib.mvq_last_error()
    assert lib.mvq_bf16x3_split_f32(None, None, 0, 16, 16, None) == 0                                     # empty batch
    assert lib.mvq_f16x2_split_f32(None, None, None, 2, 16, 16, None) == -1 and b"null" in lib.mvq_last_error()
    assert lib.mvq_conv1d_k7_pack_bf16x3(None, None, 160, 160, 0, None) == -1 and b"128 or 96" in lib.mvq_last_error()
    bf = lambda *a: lib.mvq_conv1d_k7_bf16x6_f32(None, None, None, None, None, None, None, None, None, *a, None)
    #            batch cin t cout dil tvalid
    assert bf(2, 256, 100, 256, 2, 0) == -1 and b"dilation" in lib.mvq_last_error()
    assert bf(2, 256, 100, 160, 1, 0) == -1 and bf(2, 250, 100, 256, 1, 0) == -1
    assert bf(2, 256, 100, 256, 3, 101) == -1 and b"tvalid" in lib.mvq_last_error()
    assert bf(2, 256, 100, 256, 3, 0) == -1 and b"null tensor" in lib.mvq_last_error()                    # shape fine
    assert bf(0, 256, 100, 256, 3, 0) == 0 and bf(2, 256, 0, 256, 3, 0) == 0                              # empty batch / empty rows
    f16 = lambda *a: lib.mvq_conv1d_k7_f16x3_f32(None, None, None, None, None, None, None, None, None, None, None, *a, None)
    assert f16(2, 192, 100, 192, 9, 0) == -1 and b"null tensor" in lib.mvq_last_error()
    assert f16(2, 192, 100, 192, 4, 0) == -1 and f16(0, 192, 100, 192, 9, 0) == 0
