"""-m gpu: the whole-stack C entry points (include/mvq.h "whole stacks": mvq_encoder_fwd_f32, mvq_decoder_fwd_f32,
mvq_decoder_fwd_saving_f32, mvq_decoder_bwd_input_f32) through ctypes -- bit-equal to the per-layer plan walked from Python (the
module path of rounds 1-4, kept as Encoder.forward_plan / Decoder.forward_plan) and to the C oracle, from one segment to the
headline batch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _models(dev, seed=7):
    import multimodal_vqvae_compression_audio_tactile_amd as mvq
    from multimodal_vqvae_compression_audio_tactile_amd import synth
    sd = synth.dac_state(seed)
    mdl = mvq.DAC(); mdl.load_state_dict(sd, strict=True)
    return mdl.to(dev).eval(), sd


@pytest.mark.parametrize("B", [1, 48, 256])
def test_stacks_equal_the_per_layer_plan(B, dev):
    """B = 1: latency tiles; 48: the packed / virtually packed latent-rate rows (>= 32); 256: the headline batch."""
    from multimodal_vqvae_compression_audio_tactile_amd import dac, synth
    mdl, _ = _models(dev)
    x = synth.tactile_segments(B, seed=3).to(dev)
    assert dac.USE_STACKS
    z = mdl.encoder(x)                                               # mvq_encoder_fwd_f32
    z_plan = mdl.encoder.forward_plan(x)
    assert z.shape == z_plan.shape == (B, 1024, 75) and torch.equal(z, z_plan)
    del z_plan
    y = mdl.decoder(z)                                               # mvq_decoder_fwd_f32
    y_plan = mdl.decoder.forward_plan(z)
    assert y.shape == y_plan.shape == (B, 1, 23992) and torch.equal(y, y_plan)


def test_stacks_equal_the_oracle(orc, dev):
    from multimodal_vqvae_compression_audio_tactile_amd import synth
    mdl, sd = _models(dev)
    sdn = {k: v.numpy() for k, v in sd.items()}
    T = 320 * 18
    x = synth.audio_segments(2, seed=5, T=T)
    z = mdl.encoder(x.to(dev))
    want_z = orc.dac_encoder({k[len("encoder."):]: v for k, v in sdn.items() if k.startswith("encoder.")}, x.numpy())
    assert np.array_equal(z.cpu().numpy(), want_z)
    y = mdl.decoder(z)
    want_y = orc.dac_decoder({k[len("decoder."):]: v for k, v in sdn.items() if k.startswith("decoder.")}, want_z)
    assert np.array_equal(y.cpu().numpy(), want_y)
    # ragged lengths: not a multiple of the hop, shorter than a token, empty batch
    for T2, B2 in ((320 * 20 + 137, 2), (100, 2), (24000, 0)):
        x2 = synth.tactile_segments(max(B2, 1), seed=6, T=T2)[:B2].to(dev)
        z2 = mdl.encoder(x2)
        assert torch.equal(z2, mdl.encoder.forward_plan(x2))
        y2 = mdl.decoder(z2)
        assert y2.shape == (B2, 1, mdl.decoder.stack().out_len(z2.shape[-1]))
        if z2.shape[-1]:
            assert torch.equal(y2, mdl.decoder.forward_plan(z2))


@pytest.mark.parametrize("B", [1, 6])
def test_decoder_backward_through_the_stack(B, dev):
    """mvq_decoder_fwd_saving_f32 + mvq_decoder_bwd_input_f32 == the per-layer saving forward / input-gradient of the Python plan
    (itself bit-exact against the oracle: test_decoder_input_gradient_bit_exact), through autograd as the training step uses it."""
    from multimodal_vqvae_compression_audio_tactile_amd import dac
    mdl, _ = _models(dev)
    dec = mdl.decoder
    for p in dec.parameters():
        p.requires_grad_(False)
    g = torch.Generator().manual_seed(B)
    z0 = torch.randn(B, 1024, 24, generator=g).to(dev)
    gy = torch.randn(B, 1, dec.stack().out_len(24), generator=g).to(dev)

    def run():
        z = z0.clone().requires_grad_(True)
        y = dec(z)
        y.backward(gy)
        return y.detach(), z.grad
    y1, g1 = run()
    dac.USE_STACKS = False
    try:
        y2, g2 = run()
    finally:
        dac.USE_STACKS = True
    assert torch.equal(y1, y2) and torch.equal(g1, g2)
    assert torch.equal(y1, dec(z0))                                  # the saving forward computes the inference values


def test_stack_calls_are_capturable_and_registered_as_operators(dev):
    import multimodal_vqvae_compression_audio_tactile_amd.torch_ops  # noqa: F401
    from multimodal_vqvae_compression_audio_tactile_amd import synth
    from multimodal_vqvae_compression_audio_tactile_amd.graphs import GraphedCall
    mdl, _ = _models(dev)
    x = synth.tactile_segments(2, seed=9, T=320 * 30).to(dev)
    z = mdl.encoder(x); y = mdl.decoder(z)
    o = torch.ops.mi355x_vqvae
    assert torch.equal(o.encoder_fwd(x, mdl.encoder.stack().id), z)
    assert torch.equal(o.decoder_fwd(z, mdl.decoder.stack().id), y)
    gy = torch.ones_like(y)
    zz = z.clone().requires_grad_(True)
    for p in mdl.decoder.parameters():
        p.requires_grad_(False)
    mdl.decoder(zz).backward(gy)
    assert torch.equal(o.decoder_bwd_input(z, gy, mdl.decoder.stack().id), zz.grad)
    g = GraphedCall(lambda xx: mdl.decoder(mdl.encoder(xx)), x)      # no allocation / synchronisation inside the C calls
    assert torch.equal(g(x), y)


def test_encoder_stack_falls_back_without_the_dma_form(dev):
    """at B >= 32 the virtually packed tail needs the LDS-DMA ring; where the library has no such form for a shape it
    answers MVQ_EUNSUPPORTED and both the stack and the Python plan take the plain launches (same results)."""
    from multimodal_vqvae_compression_audio_tactile_amd import dac, ops
    enc = dac.Encoder(d_model=16, strides=(2, 4), d_latent=64).to(dev).eval()      # widths 16 / 32: no 128-row LDS-DMA tile for the tail
    x = torch.randn(40, 1, 640, device=dev)
    z = enc(x)
    assert z.shape == (40, 64, 80) and torch.equal(z, enc.forward_plan(x))
