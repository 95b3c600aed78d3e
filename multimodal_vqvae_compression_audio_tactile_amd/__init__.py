"""MI355X-native encode -> vector-quantise -> decode path of Multimodal_VQVAE_compression_audio_tactile.

Drop-in surface:
  * ``DAC`` / ``Encoder`` / ``ResidualVectorQuantize`` / ``Decoder``  -- the dac.DAC(24 kHz) objects the reference
    pulls apart (``.encoder``, ``.quantizer``, ``.decoder``, ``.encode``, ``.decode``);
  * ``ResidualVQEMA`` / ``CrossPredictor`` / ``TokenNorm`` / ``PosEnc1D`` / ``AllPredAR`` / ``AllPredAR3`` (the compare_dacvsproposal_3.py
    variant) / ``ProposedEval`` / ``ProposedWrapper`` (compare_dacvsproposal_3.5_eval.py)
    -- the reference's own modules, same constructors and state-dict keys;
  * ``safe_l1`` / ``MultiResSTFTLoss`` / ``MelCosineLoss`` / ``TrainingLoss`` -- the training losses (losses.py) and
    ``train`` -- the HIP-backed autograd Functions behind ``AllPredAR.forward_step(...); total.backward()``;
  * ``Resample`` / ``resample_to`` -- torchaudio.transforms.Resample as the reference calls it; ``stsim_batch``;
  * ``optim`` -- ``AdamW`` / ``clip_grad_norm_`` with torch's call shapes on HIP kernels (the reference's own
    ``torch.optim.AdamW`` works on the modules as well);
  * ``ops`` -- tensor-level wrappers over the C ABI (include/mvq.h), ``synth`` -- seeded weights / signals.
All compute runs in libmvq_hip.so (hand-written HIP for gfx950); there is no CPU fallback.
"""
from . import ops, optim, synth, train  # noqa: F401
from .resample import Resample, resample_to  # noqa: F401
from .losses import MelCosineLoss, MultiResSTFTLoss, TrainingLoss, safe_l1, stsim_batch  # noqa: F401
from ._lib import MvqError, build, lib  # noqa: F401
from .dac import plan_overrides  # noqa: F401
from .dac import DAC, Decoder, Encoder, ResidualVectorQuantize, VectorQuantize, Snake1d, WNConv1d, WNConvTranspose1d  # noqa: F401
from .proposed import (AllPredAR, AllPredAR3, CrossPredictor, PosEnc1D, ProposedEval, ProposedWrapper, ResidualVQEMA, TokenNorm,  # noqa: F401
                       psnr_batch, psnr_global_peak_db, align_by_xcorr, crop_match, align_pair_24k,
                       psnr_3k_aligned_batch)


def build_proposed(state_dict=None, rvq_books=8, rvq_embed=512, n_codebooks=32, device="cuda", cls=None):
    """Assemble ProposedEval exactly as the reference does (build_backbones_for_eval + ProposedEval(...),
    Evaluation/dac_vcpwq_proposed6_latency.py:527-535,660-667) and optionally load a checkpoint-shaped state dict."""
    da, dt = DAC(n_codebooks=n_codebooks), DAC(n_codebooks=n_codebooks)
    net = (cls or ProposedEval)(da.encoder, da.quantizer, dt.encoder, dt.decoder, c_lat=1024,
                               rvq_books=rvq_books, rvq_embed=rvq_embed)
    if state_dict is not None:
        net.load_state_dict(state_dict, strict=True)
    return net.to(device).eval()
