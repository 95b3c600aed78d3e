#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own classes (run in the build container only).

  python tests/golden/make_golden.py

Imports /root/reference through oracle/ref_import.py (stand-ins for dac / torchaudio / soundfile, os.makedirs
patched) and records inputs-by-seed + expected outputs of:
  G1 ResidualVQEMA.forward            (Training/compare_dacvsproposal_5.py:246-265 and the eval variant with
                                        n_books_use, Evaluation/dac_vcpwq_proposed6_latency.py:409-435)
  G2 ResidualVQEMA.ema_step           (Training/compare_dacvsproposal_5.py:266-277)
  G3 CrossPredictor.forward           (Training/compare_dacvsproposal_5.py:226-244), eval mode
  G4 ProposedEval.encode_latents / forward_eval (Evaluation/dac_vcpwq_proposed6_latency.py:451-487) driven with the
     torch restatement of the DAC backbones as A_ENC / A_QUANT / T_ENC / T_DEC (the real `dac` package is absent).
     Also stored: the 32-book audio ``codes`` of A_QUANT, the per-book RVQ ``idx`` the reference's ``_nearest_l2``
     returned in every chunk (recorded by wrapping that staticmethod while encode_latents runs) and, for every one of
     those arg-maxes, the top-1 / top-2 score gap (``margin``) -- a consumer may excuse an index mismatch only where
     the stored margin is below a stated round-off bound.
  G10 compare_dacvsproposal_3.py (BASELINE.json configs[0]): its own ``AllPredAR`` (Training/...3.py:278-340; 10 books x
     K = 128 from the script's module constants) on ONE full 24 000-sample pair, B = 1, eval mode: y_hat, z_pred, z_teacher,
     r_tokens, per-book idx + margins, and the three losses / total of the script's ``step()`` body (...3.py:383-398).
  G6 align_by_xcorr                   (Evaluation/dac_vcpwq_proposed6_latency.py:164-202)
  G7 one training step's loss and gradients: the reference's AllPredAR.forward_step (Training/...5.py:293-326) on the
     restated backbones, its MultiResSTFTLoss / MelCosineLoss / safe_l1 (...:150-211; torchaudio's MelScale replaced by
     oracle/losses_torch.MelScale because torchaudio is absent), total = .55/.25/.20 mix, autograd backward (eval mode:
     dropout off; fp32, no autocast).  Stored: the losses, y_hat, dL/dy_hat and for every trainable tensor its norm
     and the subsample flat[::997].
  G8 stsim_batch (Evaluation/compare_dacvsproposal_5_eval.py:142-177), MelScale stand-in as in G7
  G9 psnr_3k_aligned_batch (Evaluation/compare_dacvsproposal_5_eval.py:188-223): align at 24 kHz, resample to 3 kHz, PSNR;
     torchaudio's Resample replaced by the restated resampler (oracle), as MelScale is in G7/G8
  G11 compare_dacvsproposal_3.5_eval.py: its ``ProposedWrapper`` (Evaluation/...3.5_eval.py:374-411; constructor
     ``(A_ENC, A_QUANT, T_ENC, T_DEC, c_lat)``, RVQ 10 x 128 from the script's constants) swept over ``books_use`` 1..3 as its
     ``eval_proposed`` does (...:504): per sweep point y, the z_run handed to T_DEC, per-book idx + margins, psnr_batch.
     (``python tests/golden/make_golden.py g11`` regenerates this fixture alone.)
  G5 psnr_batch / psnr_global_peak_db (Evaluation/compare_dacvsproposal_5_eval.py:180-185, ...6_latency.py:204-214)
Only data is stored (arrays), never reference source.  Inputs are re-created from seeds by tests/golden_inputs.py.
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import golden_inputs as gi                      # noqa: E402
from oracle import dac24_torch as T             # noqa: E402
from oracle import ref_import                   # noqa: E402

OUT = Path(__file__).resolve().parent
torch.set_grad_enabled(False)
torch.manual_seed(0)


class RecordNearest:
    """While active, ``cls._nearest_l2`` (the reference's arg-max, ...6_latency.py:417-419 / ...3.py:250-252) also logs
    what it returned and the top-1 / top-2 gap of the scores it ranked.  calls: list of (idx[N], margin[N])."""

    def __init__(self, cls):
        self.cls, self.orig, self.calls = cls, cls.__dict__["_nearest_l2"], []

    def __enter__(self):
        orig = self.orig.__func__

        def rec(x, emb):
            idx = orig(x, emb)
            sc = x @ emb.t() - 0.5 * (emb * emb).sum(dim=1).unsqueeze(0)
            top = sc.topk(2, dim=1)
            assert torch.equal(top[1][:, 0], idx) or bool((top[0][:, 0] == top[0][:, 1]).any())
            self.calls.append((idx.clone(), (top[0][:, 0] - top[0][:, 1]).clone(), sc.abs().amax(dim=1)))
            return idx
        self.cls._nearest_l2 = staticmethod(rec)
        return self

    def __exit__(self, *a):
        self.cls._nearest_l2 = self.orig

    def stacked(self, n_books, B):
        """calls arrive chunk by chunk, book by book -> idx[n_books, B, Tl] int16 and, same shape float32, margin (top-1 minus
        top-2 score) and scale (largest |score| that arg-max ranked: the yardstick of its round-off)."""
        chunks = [self.calls[i:i + n_books] for i in range(0, len(self.calls), n_books)]
        cat = lambda j: torch.cat([torch.stack([c[j].reshape(B, -1) for c in ch]) for ch in chunks], -1).numpy()
        return cat(0).astype(np.int16), cat(1).astype(np.float32), cat(2).astype(np.float32)


def audio_codes_with_margins(dac_model, a):
    """A_QUANT(A_ENC(a)) on the torch restatement -> codes[B,32,Tl] int16, margin[B,32,Tl] (gap of -dist, top-1 minus top-2),
    scale[B,32,Tl] (largest |dist| ranked)."""
    T.VectorQuantize.margin_log = []
    try:
        codes = dac_model.quantizer(dac_model.encoder(a))[1]
        mar = torch.stack([m for m, _ in T.VectorQuantize.margin_log], dim=1)
        sca = torch.stack([s for _, s in T.VectorQuantize.margin_log], dim=1)
    finally:
        T.VectorQuantize.margin_log = None
    return codes.numpy().astype(np.int16), mar.numpy().astype(np.float32), sca.numpy().astype(np.float32)


def make_g11():
    """G11: the ProposedWrapper of compare_dacvsproposal_3.5_eval.py on the restated backbones, books_use in {1, 2, 3}."""
    e35, e5 = ref_import.eval35(), ref_import.eval5()
    seed, B, uses = gi.PW_CASE
    assert (e35.RVQ_N_BOOKS_MAX, e35.RVQ_EMBED, e35.CODE_DIM) == (10, 128, 96) and e35.DAC_NQ_LIST == [1, 4, 8, 16, 32]
    sdm = gi.model_state(seed, e35.RVQ_N_BOOKS_MAX, e35.RVQ_EMBED)
    a, t = gi.pw_inputs()
    da, dt = T.DAC(), T.DAC()
    net = e35.ProposedWrapper(da.encoder, da.quantizer, dt.encoder, dt.decoder, c_lat=1024)      # ...3.5_eval.py:485
    missing, unexpected = net.load_state_dict(sdm, strict=False)                                 # ...:487 (strict=False)
    assert not missing and not unexpected
    net.eval()
    g = {}
    g["codes"], g["codes_margin"], g["codes_scale"] = audio_codes_with_margins(da, a)
    seen = {}
    hook = net.T_DEC.register_forward_hook(lambda mod, inp, out: seen.__setitem__("z_run", inp[0].detach().clone()))
    for use in uses:
        with RecordNearest(e35.ResidualVQEMA) as rec:
            y = net.forward_eval(a, t, books_use=int(use))
        g[f"use{use}.idx"], g[f"use{use}.margin"], g[f"use{use}.scale"] = rec.stacked(use, B)
        g[f"use{use}.z_run"] = seen["z_run"].numpy()
        g[f"use{use}.y"] = y.numpy()
        g[f"use{use}.psnr"] = np.array(e5.psnr_batch(t[..., :y.shape[-1]], y), np.float64)
    hook.remove()
    np.savez_compressed(OUT / "g11_proposed_wrapper.npz", **g)


def main():
    assert ref_import.available(), "reference not mounted"
    if sys.argv[1:] == ["g11"]:
        make_g11()
        return
    tr, ev, e5 = ref_import.training(), ref_import.evaluation(), ref_import.eval5()

    # ---- G1: RVQ forward
    g1 = {}
    for name, (K, nb, use, B, Tt, seed) in gi.RVQ_CASES.items():
        z, books = gi.rvq_inputs(K, nb, B, Tt, seed)
        m = ev.ResidualVQEMA(dim=96, n_books=nb, n_embed=K)
        for p, b in zip(m.books, books):
            p.data.copy_(torch.from_numpy(b))
        q_eval = m(torch.from_numpy(z), n_books_use=use).numpy()
        mt = tr.ResidualVQEMA(dim=96, n_books=nb, n_embed=K, decay=0.99)
        for p, b in zip(mt.books, books):
            p.data.copy_(torch.from_numpy(b))
        # per-book indices exactly as the reference computes them
        x = torch.from_numpy(z).permute(0, 2, 1).reshape(-1, 96)
        res, idxs = x, []
        for cb in list(mt.books)[:(nb if use is None else min(use, nb))]:
            i = mt._nearest_l2(res, cb.detach())
            idxs.append(i.numpy().astype(np.int16))
            res = res - torch.nn.functional.embedding(i, cb.detach())
        g1[f"{name}.q"] = q_eval
        g1[f"{name}.idx"] = np.stack(idxs)
        if use is None:
            g1[f"{name}.q_train"] = mt(torch.from_numpy(z)).numpy()
    np.savez_compressed(OUT / "g1_rvq_forward.npz", **g1)

    # ---- G2: EMA step
    g2 = {}
    for name, (K, nb, B, Tt, seed) in gi.EMA_CASES.items():
        z, books = gi.rvq_inputs(K, nb, B, Tt, seed)
        m = tr.ResidualVQEMA(dim=96, n_books=nb, n_embed=K, decay=0.99)
        for p, b in zip(m.books, books):
            p.data.copy_(torch.from_numpy(b))
        m.ema_step(torch.from_numpy(z))
        g2[f"{name}.books_after"] = np.stack([p.detach().numpy() for p in m.books])
    np.savez_compressed(OUT / "g2_ema_step.npz", **g2)

    # ---- G3: CrossPredictor
    g3 = {}
    sd = gi.head_state()
    cp = tr.CrossPredictor(c=1024, heads=8, mlp_mul=2, dropout=0.1).eval()
    cp.load_state_dict({k[len("predict."):]: v for k, v in sd.items() if k.startswith("predict.")}, strict=True)
    for name, (B, Tq, Tk, seed) in gi.CP_CASES.items():
        zt_prev, za = gi.cp_inputs(B, Tq, Tk, seed)
        g3[name] = cp(torch.from_numpy(zt_prev), torch.from_numpy(za)).numpy()
    np.savez_compressed(OUT / "g3_cross_predictor.npz", **g3)

    # ---- G4: ProposedEval with restated backbones
    g4 = {}
    for name, (books, K, use, B, seed) in gi.PE_CASES.items():
        sdm = gi.model_state(seed, books, K)
        a, t = gi.pe_inputs(B, seed)
        da, dt = T.DAC(), T.DAC()
        net = ev.ProposedEval(da.encoder, da.quantizer, dt.encoder, dt.decoder, c_lat=1024, rvq_books=books, rvq_embed=K)
        net.load_state_dict(sdm, strict=True)
        net.eval()
        with RecordNearest(ev.ResidualVQEMA) as rec:
            z_run = net.encode_latents(a, t, books_use=use)
        g4[f"{name}.idx"], g4[f"{name}.margin"], g4[f"{name}.scale"] = rec.stacked(books if use is None else min(use, books), B)
        g4[f"{name}.codes"], g4[f"{name}.codes_margin"], g4[f"{name}.codes_scale"] = audio_codes_with_margins(da, a)
        y = net.forward_eval(a, t, books_use=use)
        Tm = y.shape[-1]
        g4[f"{name}.z_run"] = z_run.numpy()
        g4[f"{name}.y"] = y.numpy()
        g4[f"{name}.psnr"] = np.array(e5.psnr_batch(t[..., :Tm], y), np.float64)
    np.savez_compressed(OUT / "g4_proposed_eval.npz", **g4)

    # ---- G5: PSNR helpers
    r = np.random.default_rng(5)
    ref = r.uniform(-1, 1, (3, 1, 4000)).astype(np.float32)
    est = (ref + 0.01 * r.standard_normal(ref.shape)).astype(np.float32)
    g5 = {"ref": ref, "est": est,
          "psnr_batch": np.array(e5.psnr_batch(torch.from_numpy(ref), torch.from_numpy(est)), np.float64),
          "psnr_global": np.array([ev.psnr_global_peak_db(torch.from_numpy(ref[i]), torch.from_numpy(est[i]), 4.3857)
                                   for i in range(3)], np.float64)}
    np.savez_compressed(OUT / "g5_psnr.npz", **g5)
    # ---- G6: align_by_xcorr (Evaluation/dac_vcpwq_proposed6_latency.py:164-202)
    g6 = {}
    for name, (Tlen, shift, noise, seed) in gi.ALIGN_CASES.items():
        ref, est = gi.align_inputs(Tlen, shift, noise, seed)
        r_a, e_a, s = ev.align_by_xcorr(torch.from_numpy(ref), torch.from_numpy(est), 200)
        g6[f"{name}.shift"] = np.array(s); g6[f"{name}.ref_a"] = r_a.numpy(); g6[f"{name}.est_a"] = e_a.numpy()
    np.savez_compressed(OUT / "g6_align.npz", **g6)
    # ---- G7: training step (loss + gradients) from the reference's own classes
    from oracle import losses_torch as LT
    books, K, B, seed, Tn = gi.TRAIN_CASE
    with torch.enable_grad():
        sdm = gi.model_state(seed, books, K)
        a, t = gi.train_inputs()
        da, dt = T.DAC(), T.DAC()
        net = tr.AllPredAR(da.encoder, da.quantizer, dt.encoder, dt.decoder, c_lat=1024, rvq_books=books, rvq_embed=K)
        net.load_state_dict({k: v for k, v in sdm.items()}, strict=True)
        net.eval()
        mr, mc = tr.MultiResSTFTLoss(), tr.MelCosineLoss()
        mc.mel = LT.MelScale(n_mels=64, sample_rate=24000, n_stft=257, f_min=0.0, f_max=12000.0)
        out = net.forward_step(a, t)
        y = out["y_hat"]; y.retain_grad()
        l1, st, me = tr.safe_l1(y, out["tgt"]), mr(y, out["tgt"]), mc(y, out["tgt"])
        total = tr.W_WAV_L1 * l1 + tr.W_STFT * st + tr.W_MELCOS * me
        total.backward()
    g7 = {"losses": np.array([float(l1), float(st), float(me), float(total)], np.float64),
          "y_hat": y.detach().numpy(), "dy": y.grad.numpy(), "r_tokens": out["r_tokens"].numpy()}
    for name, p_ in net.named_parameters():
        if p_.requires_grad and not name.startswith("vq.books"):
            g7[f"norm.{name}"] = np.array(float(p_.grad.norm()), np.float64)
            g7[f"sub.{name}"] = p_.grad.reshape(-1)[::gi.GRAD_STRIDE].numpy().copy()
    assert all(p_.grad is None for n_, p_ in net.named_parameters() if n_.split(".")[0] in ("A_ENC", "A_QUANT", "T_ENC", "T_DEC"))
    # the exact-value yardstick for the tolerances: the same step in float64 (torch restatement of model + losses; the
    # reference's own loss classes cast to float32 internally, so they cannot be run in double).  Same discrete path: the
    # float64 run must pick the same codes as the float32 reference, otherwise it is a different function.
    with torch.enable_grad():
        n64 = T.ProposedEval(rvq_books=books, rvq_embed=K)
        n64.load_state_dict(sdm, strict=True)
        n64 = n64.double().eval()
        for m_ in (n64.A_ENC, n64.A_QUANT, n64.T_ENC, n64.T_DEC):
            for p_ in m_.parameters():
                p_.requires_grad_(False)
        o64 = n64.forward_step(a.double(), t.double())
        assert (o64["r_tokens"] - out["r_tokens"].double()).abs().max() < 1e-4 * out["r_tokens"].abs().max()
        assert (o64["y_hat"] - y.detach().double()).abs().max() < 1e-4                     # same codes, same waveform
        y64 = o64["y_hat"]; y64.retain_grad()
        tot64, (l1_64, st_64, me_64) = LT.total_loss(y64, o64["tgt"])
        tot64.backward()
    g7["f64.losses"] = np.array([float(l1_64), float(st_64), float(me_64), float(tot64)], np.float64)
    g7["f64.dy"] = y64.grad.numpy()
    for name, p_ in n64.named_parameters():
        if p_.requires_grad and not name.startswith("vq.books"):
            g7[f"f64.norm.{name}"] = np.array(float(p_.grad.norm()), np.float64)
            g7[f"f64.sub.{name}"] = p_.grad.reshape(-1)[::gi.GRAD_STRIDE].numpy().copy()
    np.savez_compressed(OUT / "g7_train_step.npz", **g7)
    # ---- G8: stsim_batch from the reference function
    e5.torchaudio.transforms.MelScale = LT.MelScale
    ref, est = gi.stsim_inputs()
    np.savez_compressed(OUT / "g8_stsim.npz", stsim=np.array(e5.stsim_batch(ref, est), np.float64))
    # ---- G9: aligned 3 kHz PSNR from the reference functions
    e5.torchaudio.transforms.Resample = LT.Resample
    ref, est, lags = gi.aligned_psnr_inputs()
    shifts = [e5.align_pair_24k(ref[i:i + 1], est[i:i + 1])[2] for i in range(ref.shape[0])]
    np.savez_compressed(OUT / "g9_aligned_psnr.npz", psnr=np.array(e5.psnr_3k_aligned_batch(ref, est), np.float64),
                        shifts=np.array(shifts, np.int64))
    # ---- G10: compare_dacvsproposal_3.py -- its own AllPredAR + step() body, one full segment, B = 1
    t3 = ref_import.training3()
    assert (t3.RVQ_N_BOOKS, t3.RVQ_EMBED, t3.CODE_DIM) == gi.CFG3[:3]
    sdm = gi.model_state(gi.CFG3[3], t3.RVQ_N_BOOKS, t3.RVQ_EMBED)
    a, t = gi.cfg3_inputs()
    da, dt = T.DAC(), T.DAC()
    net = t3.AllPredAR(da.encoder, da.quantizer, dt.encoder, dt.decoder, c_lat=1024)
    net.load_state_dict(sdm, strict=True)
    net.eval()
    mr, mc = t3.MultiResSTFTLoss(), t3.MelCosineLoss()
    mc.mel = LT.MelScale(n_mels=64, sample_rate=24000, n_stft=257, f_min=0.0, f_max=12000.0)
    with RecordNearest(t3.ResidualVQEMA) as rec:
        out = net.forward_step(a, t)
    yh, tg = out["y_hat"], out["tgt"]
    l1, st, me = t3.safe_l1(yh, tg), mr(yh, tg), mc(yh, tg)
    total = t3.W_WAV_L1 * l1 + t3.W_STFT * st + t3.W_MELCOS * me + t3.W_LAT * 0.0          # llat = 0.0 (...3.py:389-396)
    idx, mar, sca = rec.stacked(t3.RVQ_N_BOOKS, 1)
    codes, cmar, csca = audio_codes_with_margins(da, a)
    np.savez_compressed(OUT / "g10_config3.npz", y_hat=yh.numpy(), z_pred=out["z_pred"].numpy(),
                        z_teacher=out["z_teacher"].numpy(), r_tokens=out["r_tokens"].numpy(), idx=idx, margin=mar,
                        scale=sca, codes=codes, codes_margin=cmar, codes_scale=csca,
                        losses=np.array([float(l1), float(st), float(me), float(total)], np.float64),
                        psnr=np.array(e5.psnr_batch(tg, yh), np.float64))
    make_g11()
    for f in sorted(OUT.glob("*.npz")):
        print(f.name, f.stat().st_size // 1024, "KiB")


if __name__ == "__main__":
    main()
