"""CPU: the torch restatement of the training step (oracle/dac24_torch.ProposedEval.forward_step + oracle/losses_torch)
against fixture G7, which the reference's own AllPredAR / MultiResSTFTLoss / MelCosineLoss / safe_l1 produced
(tests/golden/make_golden.py).  This pins the gradient oracle the GPU tests use."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import golden_inputs as gi                      # noqa: E402
from oracle import dac24_torch as O             # noqa: E402
from oracle import losses_torch as LT           # noqa: E402

G7 = np.load(os.path.join(os.path.dirname(__file__), "golden", "g7_train_step.npz"))


@torch.enable_grad()
def test_restated_training_step_matches_reference_fixture():
    books, K, B, seed, T = gi.TRAIN_CASE
    sd = gi.model_state(seed, books, K)
    a, t = gi.train_inputs()
    net = O.ProposedEval(rvq_books=books, rvq_embed=K)
    net.load_state_dict({k: v for k, v in sd.items() if k != "predict.pos.pe"}, strict=False)
    net.eval()
    for m in (net.A_ENC, net.A_QUANT, net.T_ENC, net.T_DEC):
        for p in m.parameters():
            p.requires_grad_(False)
    out = net.forward_step(a, t)
    y = out["y_hat"]; y.retain_grad()
    total, (l1, st, me) = LT.total_loss(y, out["tgt"])
    total.backward()
    assert np.allclose(y.detach().numpy(), G7["y_hat"], atol=1e-6)
    assert np.array_equal(out["r_tokens"].numpy(), G7["r_tokens"])
    assert np.allclose([float(l1), float(st), float(me), float(total)], G7["losses"], rtol=1e-6)
    assert np.allclose(y.grad.numpy(), G7["dy"], rtol=1e-4, atol=1e-9)
    n = 0
    for name, p in net.named_parameters():
        if f"norm.{name}" not in G7.files:
            assert p.grad is None or name.startswith("vq.books"), name
            continue
        assert abs(float(p.grad.norm()) - float(G7[f"norm.{name}"])) <= 1e-4 * float(G7[f"norm.{name}"]), name
        sub = p.grad.reshape(-1)[::gi.GRAD_STRIDE].numpy()
        assert np.allclose(sub, G7[f"sub.{name}"], rtol=1e-3, atol=1e-4 * float(np.abs(G7[f"sub.{name}"]).max())), name
        n += 1
    assert n == 21


def test_mel_filterbank_properties():
    """The restated HTK filterbank (torchaudio absent: 'parity unpinned' for the matrix itself): shape, non-negativity,
    triangular partition (adjacent filters sum to 1 between their centres), centre frequencies on the HTK mel grid."""
    fb = LT.mel_filterbank()
    assert fb.shape == (257, 64) and float(fb.min()) >= 0.0 and float(fb.max()) <= 1.0
    s = fb.sum(dim=1)
    hz = lambda m: 700.0 * (10 ** (m / 2595.0) - 1.0)
    m_max = 2595.0 * np.log10(1 + 12000 / 700.0)
    centre = hz(np.linspace(0, m_max, 66)[1:-1])
    lo, hi = int(np.ceil(centre[0] / (12000 / 256))), int(np.floor(centre[-1] / (12000 / 256)))
    inner = s[lo:hi + 1]                   # between the first and last centre every bin is covered with weight 1
    assert torch.allclose(inner, torch.ones_like(inner), atol=1e-4)
    peak_bin = fb.argmax(dim=0).numpy() * (12000 / 256)
    assert np.all(np.abs(peak_bin - centre) <= 12000 / 256)


def test_restated_stsim_matches_reference_fixture():
    G8 = np.load(os.path.join(os.path.dirname(__file__), "golden", "g8_stsim.npz"))
    ref, est = gi.stsim_inputs()
    got = LT.stsim_batch(ref, est)
    assert np.allclose(got, G8["stsim"], rtol=1e-6)
    assert got[0] > 0.999999 and got[0] > got[1] > got[2] > got[3] > 0.5
