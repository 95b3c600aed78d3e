#!/usr/bin/env python3
"""BASELINE.json configs[3]: the reference's corpus evaluation loop
(Evaluation/compare_dacvsproposal_5_eval.py:396-470) on the drop-in modules, over a SYNTHETIC corpus (the real
Vibrotactile_Files corpus and the checkpoints are not available offline): 1 003 clips (the published n), length uniform
in [1 s, 4 s] rounded to 320 samples, tactile generated at 3 kHz and audio at 44.1 kHz and resampled to 24 kHz on the
device (row f3), cut into 1-s segments (last one reflect-padded, ...5.py:115-124), segment i -> rank i mod world,
batches of --batch segments through ProposedEval.forward_eval, then the reference's metrics per segment -- ST-SIM
(stsim_batch) and the aligned 3 kHz PSNR (psnr_3k_aligned_batch) -- aggregated to mean +- CI95 exactly as the reference
does.  The only collective is the final gather of the per-rank metric lists.

  python tools/corpus_eval.py [--clips 1003] [--batch 64]
  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/corpus_eval.py --gpus N
"""
import argparse, json, math, os, sys, time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

SEG = 24000
PCM_KBPS_TACT_ORIG = 3000 * 16.0 / 1000.0          # Evaluation/compare_dacvsproposal_5_eval.py:66


def reflect_pad_right(x, need):                     # Training/compare_dacvsproposal_5.py:115-124
    while need > 0:
        T = x.shape[-1]
        if T <= 1:
            return torch.nn.functional.pad(x, (0, need), mode="replicate")
        step = min(need, T - 1)
        x = torch.nn.functional.pad(x, (0, step), mode="reflect")
        need -= step
    return x


@torch.no_grad()
def first_flip_margin(net, a1, t1, books_use, kind, book, token, i_exact, i_mode):
    """Top-1 / top-2 style margin of ONE flipped decision, recomputed in float64 from the exact path's own tensors: the score the
    exact path gave its choice minus the score it gave the mode's choice, next to the scale of the scores -- what classifies a flip
    as a near-tie (the reference fixtures store the same quantity, tests/golden/make_golden.py).  B = 1 re-run of the item (rows of
    a batch are bit-equal to their B = 1 runs)."""
    za = net.A_ENC(a1)
    qa, codes, *_ = net.A_QUANT(za)
    if kind == "audio":
        res = za[0, :, token].double()
        for j in range(book + 1):
            q = net.A_QUANT.quantizers[j]
            ze = q.in_proj.folded_weight().reshape(q.codebook.weight.shape[1], -1).double() @ res + q.in_proj.bias.double()
            cb = q.codebook.weight.double()
            zen = ze / ze.norm().clamp_min(1e-12); cbn = cb / cb.norm(dim=1, keepdim=True).clamp_min(1e-12)
            score = -((zen[None] - cbn) ** 2).sum(1)
            if j == book:
                return float(score[i_exact] - score[i_mode]), float(score.abs().max())
            e = cb[int(codes[0, j, token])]
            res = res - (q.out_proj.folded_weight().reshape(-1, cb.shape[1]).double() @ e + q.out_proj.bias.double())
    zt = net.T_ENC(t1)
    _, r_tok, idx = net._ar_latents(qa, zt, books_use, want_tokens=True, want_indices=True)
    res = r_tok[0, :, token].double()
    for j in range(book + 1):
        e = net.vq.books[j].detach().double()
        score = e @ res - 0.5 * (e * e).sum(1)
        if j == book:
            return float(score[i_exact] - score[i_mode]), float(score.abs().max())
        res = res - e[int(idx[j, 0, token])]
    raise AssertionError("unreachable")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--clips", type=int, default=1003)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--books", type=int, default=8)
    ap.add_argument("--embed", type=int, default=512)
    ap.add_argument("--backend", default="nccl", help="nccl == RCCL (default); gloo only for the one-device rehearsal")
    ap.add_argument("--force-collectives", action="store_true",
                    help="bring the RCCL group up and run the gather even with one rank (executes the RCCL path on a one-GPU box)")
    ap.add_argument("--arith", choices=["f32", "bf16x6", "f16x3"], default="f32",
                    help="opt-in, NON-PARITY arithmetic mode of the wide 7-tap convs (DESIGN.md section 6d); f32 = the exact path")
    ap.add_argument("--compare-arith", choices=["bf16x6", "f16x3"], default=None,
                    help="also run every batch in this mode and count, over the whole corpus, the audio codes / RVQ indices that differ "
                         "from the exact path's and the metric differences (one rank)")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if os.environ.get("MVQ_BENCH_ONE_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import multimodal_vqvae_compression_audio_tactile_amd as mvq
    from multimodal_vqvae_compression_audio_tactile_amd import dist as mdist, synth
    dist, ranks = None, None
    if args.force_collectives and not all(k in os.environ for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")):
        raise SystemExit("corpus_eval.py --force-collectives needs the launcher's rendezvous variables; run it as\n"
                         "  python3 -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nnodes=1 --nproc-per-node=1 "
                         "tools/corpus_eval.py --force-collectives ...")
    if world > 1 or args.force_collectives:
        # BASELINE.json configs[3]: "... frame-sharded across 8 MI355X via RCCL".  Ranks on distinct devices bring up an RCCL
        # group that must pass the all-reduce probe (fatal otherwise, no fallback); gloo only in the one-device rehearsal.
        ranks = mdist.bring_up(rank, world, local, backend=args.backend, one_device=os.environ.get("MVQ_BENCH_ONE_DEVICE") == "1")
        dist = ranks.dist

    net = mvq.build_proposed(synth.proposed_model_state(7, rvq_books=args.books, rvq_embed=args.embed),
                             rvq_books=args.books, rvq_embed=args.embed, device=dev)
    up_t, up_a = mvq.Resample(3000, 24000).to(dev), mvq.Resample(44100, 24000).to(dev)

    # the corpus: every rank derives the same clip list from the seed and keeps the segments it owns
    rng = np.random.default_rng(7)
    lens24 = (rng.uniform(1.0, 4.0, args.clips) * 24000 / 320).round().astype(np.int64) * 320
    seg_of_clip = np.ceil(lens24 / SEG).astype(np.int64)
    first = np.concatenate([[0], np.cumsum(seg_of_clip)])
    n_seg = int(first[-1])
    mine = set(mdist.shard_round_robin(n_seg, rank, world))
    a_segs, t_segs = [], []
    gdev = torch.Generator(device=dev)
    t0 = time.perf_counter()
    for c in range(args.clips):
        ids = [i for i in range(int(first[c]), int(first[c + 1])) if i in mine]
        if not ids:
            continue
        dur = lens24[c] / 24000.0
        # synthetic "recordings" at the corpus' native rates (3 kHz tactile, 44.1 kHz audio), drawn ON the device from a per-clip
        # seed (round 3 drew them on the host and copied: 11 s of the tool's wall clock for 4 s of codec work)
        gdev.manual_seed(1000 + c)
        n3, n44 = int(round(dur * 3000)), int(round(dur * 44100))
        t3 = torch.cumsum(torch.randn(1, n3, generator=gdev, device=dev), -1); t3 = t3 - t3.mean(); t3 = 0.9 * t3 / t3.abs().max().clamp_min(1e-6)
        a44 = torch.randn(1, n44, generator=gdev, device=dev); a44 = 0.9 * a44 / a44.abs().max()
        t24 = up_t(t3).clamp(-1, 1)[..., :lens24[c]]
        a24 = up_a(a44).clamp(-1, 1)[..., :lens24[c]]
        L = min(t24.shape[-1], a24.shape[-1])
        need = int(seg_of_clip[c]) * SEG - L
        t24, a24 = reflect_pad_right(t24[..., :L], need), reflect_pad_right(a24[..., :L], need)
        for i in ids:
            k = i - int(first[c])
            a_segs.append(a24[:, k * SEG:(k + 1) * SEG]); t_segs.append(t24[:, k * SEG:(k + 1) * SEG])
    torch.cuda.synchronize()
    t_load = time.perf_counter() - t0

    mvq.ops.set_arith(args.arith)
    cmp = None
    if args.compare_arith:
        if world > 1 or args.arith != "f32":
            raise SystemExit("--compare-arith: one rank, with --arith f32 (the exact path is the yardstick)")
        cmp = {"mode": args.compare_arith, "segments": 0, "audio_code_decisions": 0, "audio_codes_differing": 0, "rvq_index_decisions": 0,
               "rvq_indices_differing": 0, "segments_with_a_flip": 0, "first_flip_stage_histogram": {}, "z_run_max_rel_diff": 0.0,
               "psnr_batch_abs_delta_db_max": 0.0, "psnr_batch_abs_delta_db_max_among_unflipped": 0.0, "waveform_mse_vs_exact_sum": 0.0,
               "waveform_samples": 0}
    st_vals, ps_vals = [], []
    t_fwd = t_met = 0.0
    for s in range(0, len(a_segs), args.batch):
        a = torch.stack(a_segs[s:s + args.batch]); t = torch.stack(t_segs[s:s + args.batch])
        t1 = time.perf_counter()
        y = net.forward_eval(a, t, books_use=args.books)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        if cmp is not None:                                       # the same batch in the mode, against the exact path (not timed)
            z0, c0, i0 = net.encode_latents_with_indices(a, t, books_use=args.books)
            mvq.ops.set_arith(cmp["mode"])
            z1, c1, i1 = net.encode_latents_with_indices(a, t, books_use=args.books)
            y1 = net.T_DEC(z1)
            mvq.ops.set_arith("f32")
            i0b, i1b = i0.permute(1, 0, 2), i1.permute(1, 0, 2)    # [B, books, T]
            cmp["segments"] += a.shape[0]
            cmp["audio_code_decisions"] += c0.numel(); cmp["audio_codes_differing"] += int((c0 != c1).sum())
            cmp["rvq_index_decisions"] += i0.numel(); cmp["rvq_indices_differing"] += int((i0 != i1).sum())
            flip = (c0 != c1).flatten(1).any(1) | (i0b != i1b).flatten(1).any(1)
            cmp["segments_with_a_flip"] += int(flip.sum())
            for b in torch.nonzero(flip).flatten().tolist():      # first flipped decision of the item in dependency order
                bad = (c0[b] != c1[b])
                if bad.any():
                    tk = int(torch.nonzero(bad.any(0))[0]); st = "audio book %d" % int(torch.nonzero(bad[:, tk])[0])
                else:
                    bad = (i0b[b] != i1b[b]); tk = int(torch.nonzero(bad.any(0))[0]); st = "rvq book %d" % int(torch.nonzero(bad[:, tk])[0])
                cmp["first_flip_stage_histogram"][st] = cmp["first_flip_stage_histogram"].get(st, 0) + 1
                kind, bk = st.split(" book ")
                bk = int(bk)
                ie, im = (int(c0[b, bk, tk]), int(c1[b, bk, tk])) if kind == "audio" else (int(i0b[b, bk, tk]), int(i1b[b, bk, tk]))
                mg, sc = first_flip_margin(net, a[b:b + 1], t[b:b + 1], args.books, kind, bk, tk, ie, im)
                cmp.setdefault("first_flips", []).append({"segment": s + b, "kind": kind, "book": bk, "token": tk, "index_exact": ie,
                                                          "index_mode": im, "exact_path_margin": mg, "score_scale": sc,
                                                          "margin_over_scale": mg / sc if sc else None})
            cmp["z_run_max_rel_diff"] = max(cmp["z_run_max_rel_diff"], float((z1 - z0).abs().max() / z0.abs().max()))
            Tm = min(t.shape[-1], y.shape[-1])
            p0 = torch.tensor(mvq.psnr_batch(t[..., :Tm], y[..., :Tm])); p1 = torch.tensor(mvq.psnr_batch(t[..., :Tm], y1[..., :Tm]))
            dp = (p1 - p0).abs()
            cmp["psnr_batch_abs_delta_db_max"] = max(cmp["psnr_batch_abs_delta_db_max"], float(dp.max()))
            if (~flip).any():
                cmp["psnr_batch_abs_delta_db_max_among_unflipped"] = max(cmp["psnr_batch_abs_delta_db_max_among_unflipped"], float(dp[(~flip).cpu()].max()))
            cmp["waveform_mse_vs_exact_sum"] += float(((y1.double() - y.double()) ** 2).sum()); cmp["waveform_samples"] += y.numel()
            torch.cuda.synchronize(); t2 = time.perf_counter()
        Tl = min(t.shape[-1], y.shape[-1])
        st_vals += mvq.stsim_batch(t[..., :Tl], y[..., :Tl])
        ps_vals += mvq.psnr_3k_aligned_batch(t[..., :Tl], y[..., :Tl])
        t3_ = time.perf_counter()
        t_fwd += t2 - t1; t_met += t3_ - t2
    out = {"rank": rank, "n": len(st_vals), "st": st_vals, "ps": ps_vals, "t_load": t_load, "t_fwd": t_fwd, "t_met": t_met}
    parts = [out]
    if dist:
        # the one exchange of this config: per-rank metric vectors, padded to the longest shard, all-gathered on the device
        # over RCCL (three scalars + two float64 vectors per rank), then trimmed by the gathered counts
        dev_r = ranks.reduce_device
        head = torch.tensor([len(st_vals), t_load, t_fwd, t_met], dtype=torch.float64, device=dev_r)
        heads = [torch.zeros_like(head) for _ in range(world)]
        dist.all_gather(heads, head, group=ranks.group)
        nmax = int(max(h[0].item() for h in heads))
        body = torch.zeros(2, max(nmax, 1), dtype=torch.float64, device=dev_r)
        body[0, :len(st_vals)] = torch.tensor(st_vals, dtype=torch.float64)
        body[1, :len(ps_vals)] = torch.tensor(ps_vals, dtype=torch.float64)
        bodies = [torch.zeros_like(body) for _ in range(world)]
        dist.all_gather(bodies, body, group=ranks.group)
        parts = []
        for r_, (h, bdy) in enumerate(zip(heads, bodies)):
            n_ = int(h[0].item())
            parts.append({"rank": r_, "n": n_, "st": bdy[0, :n_].cpu().tolist(), "ps": bdy[1, :n_].cpu().tolist(),
                          "t_load": float(h[1]), "t_fwd": float(h[2]), "t_met": float(h[3])})
    if rank == 0:
        st = np.array([v for p in parts for v in p["st"]], np.float64); ps = np.array([v for p in parts for v in p["ps"]], np.float64)
        n = int(st.size)
        tps = 75.0
        kbps = tps * args.books * math.log2(args.embed) / 1000.0
        res = {"config": "synthetic corpus, round-robin segment sharding", "clips": args.clips, "segments": n_seg, "n": n,
               "n_gpus": world, "collective_backend": ranks.backend if ranks else None, "rccl_ranks": ranks.rccl_ranks if ranks else None,
               "books": args.books, "embed": args.embed, "tps": tps, "kbps": kbps,
               "compression_ratio": PCM_KBPS_TACT_ORIG / kbps,
               "stsim_mean": float(st.mean()), "stsim_ci95": 1.96 * float(st.std(ddof=0)) / max(1.0, math.sqrt(n)),
               "psnr_mean": float(ps.mean()), "psnr_ci95": 1.96 * float(ps.std(ddof=0)) / max(1.0, math.sqrt(n)),
               "seconds_max_over_ranks": {k: max(p[k] for p in parts) for k in ("t_load", "t_fwd", "t_met")},
               "segments_per_s_forward": n / max(p["t_fwd"] for p in parts),
               "segments_per_s_with_metrics": n / max(p["t_fwd"] + p["t_met"] for p in parts),
               "arith": args.arith, "arith_comparison": cmp,
               "note": "random-init weights: the PSNR / ST-SIM VALUES say nothing about codec quality; the loop, its metrics "
                       "and the sharding are what is exercised"}
        if cmp is not None and cmp["waveform_samples"]:
            mse = cmp.pop("waveform_mse_vs_exact_sum") / cmp.pop("waveform_samples")
            cmp["waveform_psnr_vs_exact_db"] = 10.0 * math.log10(1.0 / mse) if mse > 0 else float("inf")
            ff = cmp.get("first_flips", [])
            cmp["first_flip_margin_over_scale_max"] = max((abs(f["margin_over_scale"]) for f in ff), default=None)
            cmp["index_equal_fraction"] = 1.0 - (cmp["audio_codes_differing"] + cmp["rvq_indices_differing"]) / max(1, cmp["audio_code_decisions"] + cmp["rvq_index_decisions"])
            cmp["note"] = ("mode vs the EXACT path on every segment of the corpus; a flipped index changes everything downstream of it in "
                           "that segment (later books of the token, later tokens through the AR state), so differing counts include the "
                           "consequences of a first flip; every FIRST flip is classified by the exact path's own margin between the two candidates "
                           "(float64, from the exact path's tensors) relative to the score scale -- fp32 round-off is 6e-8; PSNR deltas are "
                           "psnr_batch(tactile, output) differences")
        print(json.dumps(res), flush=True)
    if dist:
        ranks.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
