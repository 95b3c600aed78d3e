"""Multi-GPU use of the path: one process per GPU, ``torch.distributed`` ("nccl" == RCCL over xGMI on ROCm).

Segments are independent (the AR dependency is inside a segment), so inference shards them across ranks with NO
data-path collective.  The only exchange the algorithm has is the codebook EMA update of the
training config (``ResidualVQEMA.ema_step``, Training/compare_dacvsproposal_5.py:266-277, called at ...:396-397):
every rank must apply the SAME update.  ``ema_step_all_ranks`` all-gathers the detached ``r_tokens`` (B*75*96 fp32
per rank: ~170 KB at the reference batch of 6 -- latency-bound on xGMI, one collective) in rank order and then runs
the identical deterministic ``ema_step`` on every rank, which preserves the reference's token-order sums bit for bit
(an all-reduce of partial sums would change the summation order).

Data-parallel training adds the one collective every DP job has: the gradients of the 21 trainable tensors (8.6 M fp32 =
34 MB).  ``allreduce_grads`` packs them into ONE flat bucket and issues ONE all-reduce after backward -- on the
point-to-point xGMI ring that is bandwidth-bound at ~2*(N-1)/N * 34 MB per link (~0.4 ms at 8 ranks), far below the
~0.5 s step, so finer bucketing / overlap with backward buys nothing here.  Every loss term of the reference is a mean
over batch items, so the global-batch gradient is sum_r (B_r / B_total) * grad_r; uneven shards are weighted accordingly.
"""
from __future__ import annotations

from typing import List, Tuple

import torch

FORCE_COLLECTIVES = False     # True: issue the collectives even in a 1-rank group (lets a one-GPU box execute the RCCL path)


def _skip(dist, group) -> bool:
    if not (dist.is_available() and dist.is_initialized()):
        return True
    return dist.get_world_size(group) == 1 and not FORCE_COLLECTIVES


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, end) slice of n_items for `rank` (sizes differ by at most one)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_round_robin(n_items: int, rank: int, world: int) -> List[int]:
    """Segment i -> rank i mod world (the corpus config of SURVEY.md section 8d)."""
    return list(range(rank, n_items, world))


def gather_tokens(r_tokens: torch.Tensor, group=None) -> torch.Tensor:
    """All-gather [B,D,T] token tensors along the batch axis in rank order (every rank gets the same tensor).
    Per-rank batch sizes may differ (uneven shards): sizes are exchanged first."""
    import torch.distributed as dist
    if _skip(dist, group):
        return r_tokens
    world = dist.get_world_size(group)
    r_tokens = r_tokens.contiguous()
    nb = torch.tensor([r_tokens.shape[0]], device=r_tokens.device, dtype=torch.int64)
    sizes = [torch.zeros_like(nb) for _ in range(world)]
    dist.all_gather(sizes, nb, group=group)
    sizes = [int(s.item()) for s in sizes]
    bmax = max(sizes)
    pad = r_tokens
    if r_tokens.shape[0] < bmax:
        pad = torch.zeros((bmax,) + tuple(r_tokens.shape[1:]), device=r_tokens.device, dtype=r_tokens.dtype)
        pad[: r_tokens.shape[0]] = r_tokens
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[:n] for b, n in zip(bufs, sizes)], dim=0)


def ema_step_all_ranks(vq, r_tokens: torch.Tensor, group=None) -> None:
    """Distributed form of `net.vq.ema_step(out["r_tokens"])`: identical codebooks on every rank afterwards."""
    vq.ema_step(gather_tokens(r_tokens, group))


def allreduce_grads(params, local_items: int, group=None) -> None:
    """Make every rank's ``.grad`` the gradient of the GLOBAL-batch mean loss: one flat-bucket all-reduce.
    Call between ``total.backward()`` and ``clip_grad_norm_`` / ``opt.step()`` (Training/...5.py:393-395)."""
    import torch.distributed as dist
    if _skip(dist, group):
        return
    params = [p for p in params if p.grad is not None]
    if not params:
        return
    dev = params[0].grad.device
    n = torch.tensor([float(local_items)], device=dev, dtype=torch.float64)
    dist.all_reduce(n, group=group)
    flat = torch.cat([p.grad.reshape(-1).to(torch.float32) for p in params])
    flat.mul_(float(local_items) / float(n.item()))
    dist.all_reduce(flat, group=group)
    off = 0
    for p in params:
        k = p.grad.numel()
        p.grad.copy_(flat[off:off + k].view_as(p.grad))
        off += k


class Ranks:
    """What ``bring_up`` returns: the process group the data path uses and how it came up."""

    def __init__(self, dist, group, backend, rccl_ranks, local_rank):
        self.dist, self.group, self.backend, self.rccl_ranks, self.local_rank = dist, group, backend, rccl_ranks, local_rank

    @property
    def reduce_device(self):
        return torch.device("cuda", self.local_rank) if self.backend == "nccl" else torch.device("cpu")

    def barrier(self):
        if self.dist is None:
            return
        if self.backend == "nccl":
            self.dist.barrier(group=self.group, device_ids=[self.local_rank])
        else:
            self.dist.barrier()


def fatal(rank, msg, code=3):
    """A rank that cannot continue says why and leaves at once (torch.distributed.run then stops the other ranks) -- it never
    sits in a collective the others will not reach."""
    import os
    import sys
    print(f"[rank {rank}] FATAL: {msg}", file=sys.stderr, flush=True)
    os._exit(code)


def bring_up(rank: int, world: int, local_rank: int, backend: str = "nccl", one_device: bool = False,
             rehearse_failure: bool = False) -> Ranks:
    """One process per GPU under torch.distributed.run.  The rendezvous goes over gloo (host only: it cannot fail for GPU
    reasons and gives every rank a way to learn that a peer died).  The data-path group is RCCL (``"nccl"`` on ROCm) and has
    to PROVE itself: one all-reduce of ones must return the world size on every rank; anything else is fatal -- there is no
    silent fallback to gloo.  gloo carries the collectives only in the explicit one-device rehearsal (``one_device``: every
    rank on cuda:0, which RCCL refuses) or when the caller asks for ``backend="gloo"``."""
    import datetime
    import sys
    import torch.distributed as dist
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=600))
    use_gloo = (one_device and not rehearse_failure) or backend == "gloo"
    if world > 1 and not one_device and backend == "gloo":
        print(f"[rank {rank}] backend gloo with distinct devices: collectives go over the host", file=sys.stderr)
    if use_gloo:
        return Ranks(dist, None, "gloo (one-device rehearsal)" if one_device else "gloo", None, local_rank)
    try:
        grp = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=180))
        probe = torch.ones(1, device=torch.device("cuda", local_rank))
        dist.all_reduce(probe, group=grp)
        torch.cuda.synchronize()
        n = int(round(float(probe.item())))
    except Exception as ex:                                    # noqa: BLE001 -- whatever RCCL raises is fatal here
        fatal(rank, f"RCCL group did not come up ({type(ex).__name__}: {ex})")
    if n != world:
        fatal(rank, f"RCCL all-reduce of ones returned {n}, expected {world}")
    return Ranks(dist, grp, "nccl", n, local_rank)
