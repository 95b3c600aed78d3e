#!/usr/bin/env python3
"""bench.py -- encode + VQ + decode throughput of the MI355X path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

N > 1 runs one rank per GPU over RCCL.  Either the caller starts the ranks (python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N: RANK / LOCAL_RANK / WORLD_SIZE in the environment) or a bare
``python bench.py --gpus N`` starts them itself: the parent process makes no GPU call, spawns torch.distributed.run as a
child with the same arguments and exits with its code (rank 0's JSON line passes through on stdout).  With N distinct
devices RCCL must come up: every rank joins one all-reduce of ones and the result (``rccl_ranks`` in the line) must equal
N, otherwise the job prints the cause and exits non-zero -- there is no silent fallback.  gloo carries the collectives only
in the explicit one-device rehearsal (MVQ_BENCH_ONE_DEVICE=1 or --backend gloo: every rank on cuda:0).

One "step" = one pass of the hot path over one batch of synthetic paired segments that is already resident in
HBM: ProposedEval.forward_eval (A_ENC + 32-book A_QUANT + T_ENC + 5 AR chunks with 8x512 RVQ + T_DEC) over
``--batch`` 1-second segments per GPU (75 token-frames each).  Segments are independent, so ranks shard them with no
data-path collective ("weak" scaling: per-GPU batch fixed).  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline     -- the dominant kernel (largest share of device time among the conv / residual-unit kernel instantiations): algorithmic
                  FLOPs per launch / its average duration measured with HIP events on the launch stream during the
                  timed steps (one event pair per kernel launch, recorded inside the library: mvq_profile_begin/_end),
                  against the fp32 MFMA peak (157.3 TFLOP/s).
  cpu_baseline -- the torch-CPU restatement of the same path (oracle/dac24_torch.py, "port") timed on this node's
                  host cores on a bounded sample, rank 0 / N=1 only.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
from collections import defaultdict
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

BF16_MFMA_PEAK_TFLOPS = 2500.0        # same guide, dense bf16 MFMA (opt-in bf16x6 mode only: six piece products per fp32 product)
FP32_MFMA_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
GFLOP_PER_SEGMENT = {"joint": 158.5, "tactile": 120.3}     # SURVEY.md section 8(d); "train" is measured, see below
TOKENS_PER_SEGMENT = 75


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="1-s segments per GPU per step")
    ap.add_argument("--torch-optim", action="store_true", help="train workload: torch.optim.AdamW + torch clip_grad_norm_ "
                    "(the reference's own calls) instead of the fused HIP update")
    ap.add_argument("--workload", choices=["joint", "tactile", "train"], default="joint",
                    help="joint / tactile: the inference round trip (the headline metric).  train: one whole training step of "
                         "BASELINE.json configs[4] (forward_step + L1/MRSTFT/MelCos + backward + clip + AdamW + codebook EMA)")
    ap.add_argument("--books", type=int, default=8)
    ap.add_argument("--embed", type=int, default=512)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl == RCCL; gloo for a "
                    "single-GPU rehearsal of the multi-rank path together with MVQ_BENCH_ONE_DEVICE=1)")
    ap.add_argument("--force-collectives", action="store_true", help="issue the training collectives even in a 1-rank group "
                    "(lets a one-GPU box execute the RCCL code path under torch.distributed.run --nproc-per-node 1)")
    ap.add_argument("--no-latency", action="store_true", help="skip the B = 1 latency section (reference protocol)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip per-launch HIP events (roofline = null)")
    ap.add_argument("--allow-overrides", action="store_true", help="run although launch-plan A/B switches are set in the environment "
                    "(MVQ_PY_PLAN, MVQ_RU_PRESNAKED, MVQ_TWO_STREAM_MAX_BATCH, ...): the line then lists them in `plan_overrides`")
    ap.add_argument("--no-sweep", action="store_true", help="skip the batch sweep (B = 1, 6, 64) behind the headline")
    ap.add_argument("--arith", choices=["f32", "bf16x6", "f16x3"], default="f32",
                    help="f32 (default, the headline): the exact fp32 fma chains of the arithmetic contract.  bf16x6: the OPT-IN, "
                         "NON-PARITY mode -- the wide units' 7-tap convs as six bf16 piece products per fp32 product (fp32-accurate, "
                         "not bit-identical to the oracle; a separate line with its own dtype, never the headline)")
    return ap.parse_args()


class KernelEvents:
    """Per-launch HIP event pairs on the launch stream, recorded INSIDE the library around every conv / residual-unit kernel
    launch (mvq_profile_begin / mvq_profile_end, include/mvq.h) together with the launch's algorithmic FLOPs -- zero-padded
    rows and tail tiles not counted.  One entry per kernel instantiation, named as rocprofv3 names it, so `seconds /
    launches` here is the same quantity as that kernel's average duration in `rocprofv3 --kernel-trace --stats`."""

    def __init__(self, ops):
        self.ops = ops
        ops.profile_begin()

    def summary(self):
        return self.ops.profile_end()


def host_cores() -> int:
    """Cores this process may actually use: min(affinity mask, cgroup CPU quota) -- the GPU box hands each job a CPU
    share (16 for one GPU); running torch with one thread per visible core (256) oversubscribes it badly."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(workload, books, embed, sd):
    """torch-CPU restatement on the host cores: bounded sample (B=6, the reference batch; 1 warm-up + timed reps
    until ~15 s)."""
    from oracle import dac24_torch as T
    from multimodal_vqvae_compression_audio_tactile_amd import synth
    threads = host_cores()
    torch.set_num_threads(threads)
    net = T.ProposedEval(rvq_books=books, rvq_embed=embed).eval()
    missing = net.load_state_dict(sd, strict=True)
    B = 6
    a, t = synth.audio_segments(B, seed=7), synth.tactile_segments(B, seed=7)
    tact = workload == "tactile"
    with torch.no_grad():
        t0 = time.perf_counter(); net.forward_eval(a[:1], t[:1], tactile_only=tact); warm = time.perf_counter() - t0
        reps, spent = 0, 0.0
        while spent < 12.0 and reps < 10:
            t0 = time.perf_counter(); net.forward_eval(a, t, tactile_only=tact); spent += time.perf_counter() - t0
            reps += 1
    seg_s = B * reps / spent
    return {"value": seg_s * TOKENS_PER_SEGMENT, "unit": "token-frames/s", "cores": threads, "kind": "port",
            "segments_per_s": seg_s,
            "sample": f"{reps} x forward_eval on B={B} synthetic 1-s segments ({workload}), torch {torch.__version__} "
                      f"CPU fp32 restatement (oracle/dac24_torch.py), {threads} threads, warm-up {warm:.1f}s"}


def cpu_baseline_train(books, embed, sd):
    """One reference training step (restated: oracle/dac24_torch.forward_step + oracle/losses_torch + torch autograd +
    AdamW) on the host cores at the reference's batch of 6."""
    from oracle import dac24_torch as T, losses_torch as LT
    from multimodal_vqvae_compression_audio_tactile_amd import synth
    threads = host_cores()
    torch.set_num_threads(threads)
    net = T.ProposedEval(rvq_books=books, rvq_embed=embed)
    net.load_state_dict(sd, strict=True)
    net.train()
    for m in (net.A_ENC, net.A_QUANT, net.T_ENC, net.T_DEC):
        for p in m.parameters():
            p.requires_grad_(False)
    params = [p for n, p in net.named_parameters() if p.requires_grad and not n.startswith("vq.books")]
    opt = torch.optim.AdamW(params, lr=2e-4, weight_decay=1e-5)
    B = 6
    a, t = synth.audio_segments(B, seed=7), synth.tactile_segments(B, seed=7)
    reps, spent = 0, 0.0
    with torch.enable_grad():
        while spent < 15.0 and reps < 4:
            t0 = time.perf_counter()
            out = net.forward_step(a, t)
            total, _ = LT.total_loss(out["y_hat"], out["tgt"])
            opt.zero_grad(set_to_none=True); total.backward()
            torch.nn.utils.clip_grad_norm_(params, 3.0); opt.step()
            dt = time.perf_counter() - t0
            if reps or dt < 8.0:
                spent += dt; reps += 1
            else:
                reps, spent = 1, dt
    seg_s = B * reps / spent
    return {"value": seg_s * TOKENS_PER_SEGMENT, "unit": "token-frames/s", "cores": threads, "kind": "port",
            "segments_per_s": seg_s,
            "sample": f"{reps} x training step on B={B} synthetic 1-s segments, torch {torch.__version__} CPU fp32 "
                      f"restatement + autograd + AdamW (codebook EMA not included), {threads} threads"}


def self_launch(args) -> int:
    """Bare ``python bench.py --gpus N`` (N > 1, no WORLD_SIZE): start the N ranks as a child job.  This process has made
    no GPU call (device_count() does not initialise HIP on this image) and never execs: it waits and returns the child's code."""
    import subprocess
    one_dev = os.environ.get("MVQ_BENCH_ONE_DEVICE") == "1"
    ndev = torch.cuda.device_count()
    if ndev < args.gpus and not one_dev:
        print(f"bench.py: --gpus {args.gpus} but only {ndev} HIP device(s) visible (set MVQ_BENCH_ONE_DEVICE=1 for the "
              "one-device rehearsal of the multi-rank path)", file=sys.stderr)
        return 2
    # the launcher picks the rendezvous port itself (--standalone: c10d store on a free port of its own choosing), so there is
    # no window in which another job on the node can take a port this process probed and released
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC (RCCL across processes on this pool)
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // args.gpus)))
    return subprocess.run(cmd, env=env).returncode


def fatal(rank, msg, code=3):
    """A rank that cannot continue: say why and leave at once (torch.distributed.run then stops the other ranks) --
    never sit in a collective the others will not reach."""
    print(f"[bench rank {rank}] FATAL: {msg}", file=sys.stderr, flush=True)
    os._exit(code)


def pmc_traffic_table(want=None):
    """profiles/pmc_traffic.json (HBM bytes per launch per kernel, from two rocprofv3 --pmc passes of this same command,
    tools/pmc_traffic.py) + where it came from: the commit it was profiled at and whether csrc/ still has the same content."""
    f = ROOT / "profiles" / "pmc_traffic.json"
    if not f.exists():
        return {}, None
    try:
        tab = json.loads(f.read_text())
    except Exception:
        return {}, None
    meta = tab.pop("_meta", {}) if isinstance(tab, dict) else {}
    from tools.pmc_traffic import csrc_digest
    now = csrc_digest(ROOT)
    # the per-launch bytes describe launches of ONE command (kernel names repeat across workloads with other shapes): tables without
    # the stamp are the default command's (round <= 4: joint, 256 segments, f32)
    ran = (meta.get("workload", "joint"), int(meta.get("batch", 256)), meta.get("arith", "f32"))
    if want is not None and ran != want:
        return {}, None
    src = {"file": "profiles/pmc_traffic.json", "profiled_at_commit": meta.get("commit"), "profiled_command": {"workload": ran[0], "batch": ran[1], "arith": ran[2]},
           "csrc_sha16_profiled": meta.get("csrc_sha16"), "csrc_sha16_now": now,
           "kernels_unchanged_since_profile": bool(meta.get("csrc_sha16")) and meta.get("csrc_sha16") == now}
    return tab, src


def vq_lds_table():
    """profiles/pmc_vq_lds.json (tools/pmc_vq.py: SQ counters of the codebook-search kernels from a rocprofv3 --pmc pass of this
    same command) -> the `vq` object of the line: LDS instructions per global read, LDS share of the operand fetches, bank-conflict
    fraction and LDS-array busy share of peak per kernel, with the stamp that says whether the kernels changed since."""
    f = ROOT / "profiles" / "pmc_vq_lds.json"
    if not f.exists():
        return None
    try:
        tab = json.loads(f.read_text())
    except Exception:
        return None
    meta = tab.pop("_meta", {})
    from tools.pmc_traffic import csrc_digest
    now = csrc_digest(ROOT)
    keys = ("launches", "lds_per_vmem_rd", "lds_operand_share", "lds_bank_conflict_frac", "lds_busy_frac_of_peak")
    return {"bound": "lds", "kernels": {k: {q: v.get(q) for q in keys} for k, v in tab.items()},
            "peak": "one LDS access cycle per CU per clock (128 B ds_read_b32, 256 B b64/b128; MI355X guide, LDS table)",
            "source": {"file": "profiles/pmc_vq_lds.json", "profiled_at_commit": meta.get("commit"),
                       "csrc_sha16_profiled": meta.get("csrc_sha16"), "csrc_sha16_now": now,
                       "kernels_unchanged_since_profile": bool(meta.get("csrc_sha16")) and meta.get("csrc_sha16") == now}}


def batch_sweep(net, synth, dev, tact, gflop_per_segment, headline_batch, y_headline, a_headline, t_headline):
    """The reference's own operating points next to the headline (B in {1, 6, 64, 256}; the eval
    loader and the training loop run batches of 6, Evaluation/compare_dacvsproposal_5_eval.py:487-489, Training/...5.py:62): the
    same forward_eval on the first B segments of the headline batch, 3 warm-ups + timed repeats with the inputs resident, ms per
    step, segments/s and the algorithmic fraction of the fp32 MFMA peak -- each under the headline's rule: rows of the B-segment
    output must equal the same segments' rows of the headline output bit for bit (batch independence), else the run fails."""
    out = {}
    # Rows 0..B-1 of the HEADLINE's own input tensors whenever the headline batch has them.  (A fresh draw of another size is not the
    # same data: synth shapes its noise through an FFT whose rounding can depend on the batch it is called with -- round 5,
    # `--batch 6`: every comparison failed on inputs that differed in the last place, gpurun_out/f9small.)  A sweep size beyond the
    # headline batch gets a draw of its own and is timed but not compared.
    big = max(64, headline_batch)
    a_all, t_all = (a_headline, t_headline) if headline_batch >= 64 else (synth.audio_segments(big, seed=7).to(dev), synth.tactile_segments(big, seed=7).to(dev))
    fwd = (lambda a, t: net.forward_eval_tactile_only(t, books_use=None)) if tact else (lambda a, t: net.forward_eval(a, t, books_use=None))
    for B in (1, 6, 64):
        src_a, src_t = (a_headline, t_headline) if B <= headline_batch else (a_all, t_all)
        a, t = src_a[:B].contiguous(), src_t[:B].contiguous()
        for _ in range(3):
            y = fwd(a, t)
        torch.cuda.synchronize()
        reps = {1: 50, 6: 30, 64: 8}[B]
        t0 = time.perf_counter()
        for _ in range(reps):
            y = fwd(a, t)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / reps
        n = min(B, y_headline.shape[0])
        same = None if B > headline_batch else (bool(torch.equal(y[:n], y_headline[:n])) if y.shape[1:] == y_headline.shape[1:] else False)
        tf = B * gflop_per_segment / ms                          # GFLOP / ms = TFLOP/s
        out[f"B{B}"] = {"ms_per_step": ms, "segments_per_s": 1e3 * B / ms, "path_tflops": tf, "frac_of_fp32_mfma_peak": tf / FP32_MFMA_PEAK_TFLOPS,
                        "rows_bit_equal_to_headline_batch": same, "repeats": reps}
    out["note"] = ("rows 0..B-1 of the headline batch's own inputs (a sweep size beyond the headline batch: its own draw, not compared: null); eager calls, inputs resident, "
                   "device synchronised around the timed repeats")
    return out


def latency_b1(mvq, synth, dev, books, embed, sd):
    """B = 1 latency in the reference's own protocol (Evaluation/dac_vcpwq_proposed6_latency.py:489-525): 1 s of zeros,
    3 warm-ups, 10 repeats, device synchronised before each clock read; encode_latents and T_DEC timed separately -- eager
    (one host call per kernel launch) and replayed as ONE hipGraph each (graphs.GraphedCall; outputs bit-equal to eager)."""
    from multimodal_vqvae_compression_audio_tactile_amd.graphs import GraphedCall
    net = mvq.build_proposed(sd, rvq_books=books, rvq_embed=embed, device=dev)
    a = torch.zeros(1, 1, 24000, device=dev); t = torch.zeros(1, 1, 24000, device=dev)
    for _ in range(3):
        z = net.encode_latents(a, t, books_use=books); net.T_DEC(z)
    torch.cuda.synchronize()

    def timed(fn):
        ts = []
        for _ in range(10):
            t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return ts, out
    enc, z = timed(lambda: net.encode_latents(a, t, books_use=books))
    dec, y = timed(lambda: net.T_DEC(z))
    res = {"encode_ms": sum(enc) / len(enc), "decode_ms": sum(dec) / len(dec), "encode_min_ms": min(enc), "decode_min_ms": min(dec),
           "protocol": "B=1, 1 s of zeros @ 24 kHz, 3 warm-ups + 10 repeats, sync before each clock read; fp32 exact path",
           "reference_published": {"encode_ms": [12.8, 16.3], "decode_ms": [2.75, 2.86], "hardware": "unstated CUDA GPU, AMP",
                                   "source": "Evaluation/eval_vs_dac24_with_vcpwq_rawPSNR_latency/eval_all_vs_dac24_vcpwq_rawPSNR_latency.json:89-90"}}
    try:
        g_enc = GraphedCall(lambda aa, tt: net.encode_latents(aa, tt, books_use=books), a, t)
        g_dec = GraphedCall(lambda zz: net.T_DEC(zz), z)
        genc, zg = timed(lambda: g_enc(a, t))
        gdec, yg = timed(lambda: g_dec(z))
        res["graph_replay"] = {"encode_ms": sum(genc) / len(genc), "decode_ms": sum(gdec) / len(gdec), "encode_min_ms": min(genc),
                               "decode_min_ms": min(gdec), "bit_equal_to_eager": bool(torch.equal(zg, z) and torch.equal(yg, y))}
    except Exception as ex:
        res["graph_replay"] = {"error": repr(ex)}
    return res


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    # stdout carries exactly ONE line (rank 0's JSON): whatever libraries print on fd 1 while they come up (gloo's rank
    # banner, RCCL's version banner) is sent to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a line for the wrong GPU count", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (HIP device); there is no CPU fallback for the product path")
    one_dev = os.environ.get("MVQ_BENCH_ONE_DEVICE") == "1"      # rehearsal: every rank on cuda:0 (never on the 8-GPU node)
    if one_dev:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        fatal(rank, f"LOCAL_RANK {local_rank} has no device ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    under_launcher = "WORLD_SIZE" in os.environ
    dist, grp, backend_used, rccl_ranks, ranks = None, None, None, None, None
    if under_launcher:
        # gloo rendezvous, then an RCCL group that has to complete one all-reduce of ones on every rank (dist.bring_up); failure
        # is fatal.  MVQ_BENCH_REHEARSE_RCCL_FAILURE=1 (test hook): keep RCCL although every rank sits on cuda:0, which RCCL
        # refuses -- exercises the fatal path (the job must end non-zero within seconds, not hang)
        from multimodal_vqvae_compression_audio_tactile_amd import dist as mdist_
        ranks = mdist_.bring_up(rank, world, local_rank, backend=args.backend, one_device=one_dev,
                                rehearse_failure=os.environ.get("MVQ_BENCH_REHEARSE_RCCL_FAILURE") == "1")
        dist, grp, backend_used, rccl_ranks = ranks.dist, ranks.group, ranks.backend, ranks.rccl_ranks
    red_dev = ranks.reduce_device if ranks else torch.device("cpu")

    def barrier():
        if ranks:
            ranks.barrier()

    import multimodal_vqvae_compression_audio_tactile_amd as mvq
    from multimodal_vqvae_compression_audio_tactile_amd import ops, synth
    overrides = mvq.plan_overrides()
    if overrides and not args.allow_overrides:
        print(f"bench.py: launch-plan A/B switches in the environment ({', '.join(overrides)}): not the product's plan; "
              "pass --allow-overrides to measure it anyway (the line lists them)", file=sys.stderr)
        sys.exit(2)
    ops.set_arith(args.arith)                                    # "f32" unless the opt-in, non-parity mode was asked for

    sd = synth.proposed_model_state(7, rvq_books=args.books, rvq_embed=args.embed)
    train = args.workload == "train"
    net = mvq.build_proposed(sd, rvq_books=args.books, rvq_embed=args.embed, device=dev, cls=mvq.AllPredAR if train else None)
    B = args.batch
    a = synth.audio_segments(B, seed=7 + rank).to(dev)
    t = synth.tactile_segments(B, seed=7 + rank).to(dev)
    tact = args.workload == "tactile"

    if train:
        from multimodal_vqvae_compression_audio_tactile_amd import dist as mdist
        mdist.FORCE_COLLECTIVES = bool(args.force_collectives)
        net.train()                                              # ctx dropout on, as in the reference's epoch loop
        crit = mvq.TrainingLoss()
        params = [p for n, p in net.named_parameters() if p.requires_grad and not n.startswith("vq.books")]
        opt = (torch.optim.AdamW if args.torch_optim else mvq.optim.AdamW)(params, lr=2e-4, weight_decay=1e-5)   # ...5.py:54-55,367

    phase_ev = []                # train: per step, (name, start event, end event) on the current stream = the stream of every launch
    # the phase events of the timed region exist BEFORE it starts (torch creates the hipEvent at the first record(), so each one
    # is recorded once here): no hipEventCreate inside the timed loop, like the library's own per-launch events
    phase_pool = []
    if train:
        for _ in range(2 * 3 * args.steps):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            phase_pool.append(e)

    def phase(name, fn):
        if not train or not collect_phases:
            return fn()
        e0 = phase_pool.pop() if phase_pool else torch.cuda.Event(enable_timing=True)
        e1 = phase_pool.pop() if phase_pool else torch.cuda.Event(enable_timing=True)
        e0.record(); r = fn(); e1.record()
        phase_ev.append((name, e0, e1))
        return r

    def step():
        if train:                                                # Training/compare_dacvsproposal_5.py:379-397
            with torch.enable_grad():
                out = net.forward_step(a, t)
                total = crit(out["y_hat"], out["tgt"])
                opt.zero_grad(set_to_none=True)
                total.backward()
            if dist:
                phase("grad_allreduce", lambda: mdist.allreduce_grads(params, B, group=grp))
            if args.torch_optim:
                torch.nn.utils.clip_grad_norm_(params, 3.0)
                opt.step()
            else:
                _, coef = mvq.optim.clip_coef(params, 3.0)      # clip_grad_norm_(params, 3.0) fused into the update
                opt.step(clip_coef=coef)
            toks = out["r_tokens"]
            if dist:
                toks = phase("ema_token_allgather", lambda: mdist.gather_tokens(toks, group=grp))
            phase("ema_assign_and_update", lambda: net.vq.ema_step(toks))       # identical update on every rank
            return out["y_hat"].detach()
        if tact:
            return net.forward_eval_tactile_only(t, books_use=None)
        return net.forward_eval(a, t, books_use=None)

    collect_phases = False
    launches_per_step = 0
    for w in range(args.warmup):
        if w == args.warmup - 1 and not args.no_kernel_events:  # count the launches of one step: the event pool is sized from it
            ops.profile_begin()
            y = step()
            launches_per_step = sum(v["launches"] for v in ops.profile_end().values())
        else:
            y = step()
    torch.cuda.synchronize()

    kev = None
    if not args.no_kernel_events:                          # train: forward (saving form) + decoder input-gradient convs
        # every event pair of the timed region exists before it starts: no hipEventCreate inside the timed loop
        ops.profile_reserve((launches_per_step or 256) * args.steps + 64)
        kev = KernelEvents(ops)
    collect_phases = train

    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    summ = kev.summary() if kev else None
    collect_phases = False

    # the same steps once more WITHOUT the per-launch events (the two event records per launch sit between back-to-back
    # kernels): reported next to the headline so the instrumentation's cost is a number, not an assumption
    unprof_ms = None
    if kev and args.steps > 0:
        n_un = min(args.steps, 3)
        barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n_un):
            y = step()
        torch.cuda.synchronize()
        unprof_ms = 1e3 * (time.perf_counter() - t1) / n_un

    # Parity spot check of the HEADLINE configuration (inference workloads): sampled segments of this batch, re-run one at a
    # time (B = 1: other tiles, other launch plan, same fma chains), must equal their rows of the B-segment output bit for bit.
    # A mismatch fails the run: a throughput number for wrong results is not reported.
    spot = None
    if not train:
        idx = sorted({0, B // 3, B - 1})
        bad = []
        for i in idx:
            yi = (net.forward_eval_tactile_only(t[i:i + 1], books_use=None) if tact
                  else net.forward_eval(a[i:i + 1], t[i:i + 1], books_use=None))
            if yi.shape[1:] != y.shape[1:] or not torch.equal(yi[0], y[i]):
                bad.append(i)
        torch.cuda.synchronize()
        spot = {"segments": idx, "bit_equal_to_B1": not bad, "mismatching": bad}
        if bad:
            print(f"[bench rank {rank}] parity spot check FAILED: segments {bad} of the B={B} output differ from their B=1 runs", file=sys.stderr, flush=True)

    if dist:
        tt = torch.tensor([elapsed], device=red_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=grp)              # grp is None (default gloo group) in the rehearsal
        elapsed = float(tt.item())

    # Opt-in arithmetic mode: what it changes against the EXACT path on the first segments of this very batch (never a parity
    # claim: the line carries its own dtype).  Indices compared entry by entry; an item counts from its first flipped decision on.
    arith_check = None
    if args.arith != "f32" and rank == 0 and not train and not tact:
        nb = min(B, 16)
        def both():
            z, codes, idx = net.encode_latents_with_indices(a[:nb], t[:nb], books_use=None)
            return z, codes, idx, net.T_DEC(z)
        z1, c1, i1, y1 = both()
        ops.set_arith("f32")
        z0, c0, i0, y0 = both()
        ops.set_arith(args.arith)
        Tm = min(y0.shape[-1], t.shape[-1])
        p0 = torch.tensor(mvq.psnr_batch(t[:nb, :, :Tm], y0[..., :Tm])); p1 = torch.tensor(mvq.psnr_batch(t[:nb, :, :Tm], y1[..., :Tm]))
        mse = float(((y1.double() - y0.double()) ** 2).mean())
        arith_check = {"segments": nb, "against": "the exact f32 path on the same segments",
                       "audio_codes_equal_fraction": float((c1 == c0).double().mean()),
                       "rvq_indices_equal_fraction": float((i1 == i0).double().mean()),
                       "items_with_any_flipped_index": int(((c1 != c0).flatten(1).any(1) | (i1 != i0).permute(1, 0, 2).flatten(1).any(1)).sum()),
                       "z_run_max_rel_diff": float((z1 - z0).abs().max() / z0.abs().max()),
                       "waveform_psnr_vs_exact_db": (10.0 * math.log10(1.0 / mse) if mse > 0 else float("inf")),
                       "recon_psnr_delta_db_max": float((p1 - p0).abs().max()), "recon_psnr_db_exact_mean": float(p0.mean())}
    out_ok = bool(torch.isfinite(y).all().item()) and y.shape[0] == B
    spot_ok = spot is None or spot["bit_equal_to_B1"]
    if dist:                                                   # a failing rank fails the job
        so = torch.tensor([1.0 if spot_ok else 0.0], device=red_dev, dtype=torch.float64)
        dist.all_reduce(so, op=dist.ReduceOp.MIN, group=grp)
        spot_ok = bool(so.item() == 1.0)

    if rank == 0:
        segs = B * world * args.steps
        seg_s = segs / elapsed
        line = {
            "metric": "encode+VQ+decode token-frames/sec (1 frame = 13.33 ms of paired 24 kHz audio+tactile)",
            "value": seg_s * TOKENS_PER_SEGMENT, "unit": "token-frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.arith, "data": "synthetic",
            "segments_per_s": seg_s, "output_finite": out_ok, "rccl_ranks": rccl_ranks,
            "parity_spot_check": (spot_ok if spot is not None else None), "parity_spot_check_detail": spot,
            "build_flags": ops.build_flags(), "plan_overrides": overrides, "arith": args.arith, "arith_check": arith_check,
            "config": {"workload": ("joint audio+tactile ProposedEval.forward_eval (compare_dacvsproposal_5 config: "
                                    "2x DAC-24k encoder, 32x1024x8 audio RVQ, CrossPredictor AR x5 chunks, "
                                    f"RVQ {args.books}x{args.embed}x96, DAC-24k decoder)") if not tact else
                                   (f"tactile-only chain: T_ENC -> TokenNorm/tanh -> proj_down -> RVQ {args.books}x{args.embed}x96 "
                                    "-> proj_up -> T_DEC"),
                       "segments_per_gpu_per_step": B, "segment": "1 s @ 24 kHz = 75 token-frames",
                       "sharding": f"segments sharded over {world} GPU(s), no data-path collective",
                       "collective_backend": backend_used, "rccl_ranks": rccl_ranks,
                       "weights": "seeded variance-preserving random init of the DAC-24k architecture"},
        }
        if train:
            line["metric"] = "training step token-frames/sec (forward_step + losses + backward + AdamW + codebook EMA)"
            line["config"]["workload"] = ("AllPredAR training step (BASELINE.json configs[4], compare_dacvsproposal_5.py:379-397): "
                                          "frozen DAC encoders/quantiser/decoder, trainable CrossPredictor/TokenNorm/scale/proj, "
                                          f"RVQ {args.books}x{args.embed}x96 with EMA update, loss 0.55 L1 + 0.25 MRSTFT + 0.20 MelCos, fp32")
            line["config"]["sharding"] = (f"data parallel over {world} GPU(s): one flat-bucket gradient all-reduce (34 MB) "
                                          "+ one token all-gather for the codebook EMA per step")
        else:
            gf = GFLOP_PER_SEGMENT[args.workload]
            line["path_tflops"] = seg_s * gf * 1e-3 / world            # per GPU, algorithmic
        if unprof_ms is not None:
            # rank 0, a SEPARATE measurement: min(steps, 3) further steps behind a barrier, after the measured region (for the
            # train workload these steps go on updating the model and codebooks, so it is a like-for-like cost, not the same data)
            line["ms_per_step_without_kernel_events"] = unprof_ms
        ms_ = torch.cuda.memory_stats(dev)
        line["device_memory"] = {"peak_allocated_GB": ms_.get("allocated_bytes.all.peak", 0) / 1e9,
                                 "peak_reserved_GB": ms_.get("reserved_bytes.all.peak", 0) / 1e9,
                                 "alloc_retries": ms_.get("num_alloc_retries", 0)}
        if train and phase_ev:
            acc = defaultdict(float)
            for name, e0, e1 in phase_ev:
                acc[name] += e0.elapsed_time(e1)
            line["train_phases_ms_per_step"] = {k: v / args.steps for k, v in acc.items()}
            line["train_phases_ms_per_step"]["note"] = ("HIP events on the launch stream, rank 0; ema_assign_and_update = codebook "
                                                        f"search of all {args.books} books + counting-sort update over "
                                                        f"{B * world * TOKENS_PER_SEGMENT} gathered tokens")
        if kev:
            dom = max(summ.items(), key=lambda kv: kv[1]["seconds"])
            name, d = dom
            ach = d["flops"] / d["seconds"] * 1e-12
            pmc_all, pmc_src = pmc_traffic_table((args.workload, args.batch, args.arith))
            traffic = pmc_all.get(name, {}).get("hbm_bytes_per_launch")       # null when this kernel is not in the profile
            # a bf16x6 kernel (opt-in mode only) spends six bf16 MFMA products per algorithmic fp32 product: its ceiling is the dense
            # bf16 MFMA peak / 6, in the same algorithmic TFLOP/s the other kernels are quoted in
            peak = FP32_MFMA_PEAK_TFLOPS
            if name.startswith("conv_k7_pieces_kernel"):           # last template argument = pieces per operand: 3 -> six products, 2 -> three
                peak = BF16_MFMA_PEAK_TFLOPS / (3.0 if name.rstrip(">").endswith(", 2") else 6.0)
            line["roofline"] = {"bound": "mfma", "kernel": name, "achieved": ach, "peak": peak,
                                "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic,
                                "traffic_source": pmc_src if traffic is not None else None,
                                "launches": d["launches"], "avg_launch_us": 1e6 * d["seconds"] / d["launches"],
                                "flop_per_launch": d["flops"] / d["launches"],
                                "share_of_conv_time": d["seconds"] / sum(v["seconds"] for v in summ.values())}
            conv_s = sum(v["seconds"] for v in summ.values()); conv_f = sum(v["flops"] for v in summ.values())
            line["conv_stack"] = {"tflops": conv_f / conv_s * 1e-12, "frac_of_fp32_mfma_peak": conv_f / conv_s * 1e-12 / FP32_MFMA_PEAK_TFLOPS,
                                  "seconds_per_step": conv_s / args.steps,
                                  "kernels": {k: {"tflops": v["flops"] / v["seconds"] * 1e-12,
                                                  "ms_per_step": 1e3 * v["seconds"] / args.steps,
                                                  "launches_per_step": v["launches"] // args.steps}
                                              for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["seconds"])}}
            # achieved HBM rate of the conv stacks: PMC bytes per launch (profiles/pmc_traffic.json, collected on this same
            # command as the MI355X guide prescribes) x the launches timed here, over their summed HIP-event durations
            if pmc_all:
                by = sum(pmc_all[k]["hbm_bytes_per_launch"] * v["launches"] for k, v in summ.items() if k in pmc_all)
                cov = sum(v["seconds"] for k, v in summ.items() if k in pmc_all) / conv_s
                if cov > 0.95:
                    line["conv_stack"]["hbm"] = {"achieved_GBps": by / conv_s * 1e-9, "peak_GBps": 8000.0,
                                                 "frac": by / conv_s * 1e-9 / 8000.0,
                                                 "GB_per_step": by / args.steps * 1e-9, "source": pmc_src}
        if not train:
            vq = vq_lds_table()
            if vq:
                line["vq"] = vq
        if world == 1 and not train and not args.no_sweep and args.arith == "f32":
            try:
                line["batch_sweep"] = batch_sweep(net, synth, dev, tact, GFLOP_PER_SEGMENT[args.workload], B, y, a, t)
            except Exception as ex:
                line["batch_sweep"] = {"error": repr(ex)}
        if world == 1 and not train and not args.no_latency:
            try:
                line["latency_b1"] = latency_b1(mvq, synth, dev, args.books, args.embed, sd)
            except Exception as ex:
                line["latency_b1"] = {"error": repr(ex)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = (cpu_baseline_train(args.books, args.embed, sd) if train else
                                        cpu_baseline(args.workload, args.books, args.embed, sd))
                line["speedup_vs_cpu"] = line["value"] / line["cpu_baseline"]["value"]
            except Exception as ex:       # the baseline is reported, never required for the GPU number
                line["cpu_baseline"] = {"value": None, "unit": "token-frames/s", "cores": torch.get_num_threads(),
                                        "kind": "port", "sample": f"failed: {ex!r}"}
        sweep_ok = all(v.get("rows_bit_equal_to_headline_batch", True) is not False for v in line.get("batch_sweep", {}).values() if isinstance(v, dict))
        if not sweep_ok:
            print("[bench] batch sweep: a small-batch output differs from the same segments of the headline batch", file=sys.stderr, flush=True)
            spot_ok = False
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if not spot_ok:
        sys.exit(4)


if __name__ == "__main__":
    main()
