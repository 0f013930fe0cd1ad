"""Multi-GPU sharding of independent node-AVIs.

One process per GPU.  Rank g owns the contiguous node range [g*B/G, (g+1)*B/G) and assembles its own
stacked blocks on its own GPU, so no M/q traffic crosses GPUs.  The only exchange per outer sweep is
the all-gather of the primal blocks (count x n fp64 per rank) so that every rank holds the full
iterate x for the next sweep's R_i w / B_i w terms -- RCCL over xGMI on the GPU box
(torch.distributed backend "nccl"), gloo in the CPU tests.  The message is small (config 4:
320 KB per rank) and latency-bound; one collective per sweep, no per-node messages.

What does NOT shard: a single Nash pool is ONE AVI (src/avi.jl:399-400) -- shard over instances
instead.
"""
from __future__ import annotations


def node_range(total: int, world: int, rank: int):
    """Contiguous, balanced node range of `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def all_gather_primal(x_all, x_local, ranges, dist):
    """Reassemble the primal iterate: x_all[lo_r:hi_r] <- rank r's x_local, on every rank.

    `ranges` = [node_range(total, world, r) for r in range(world)].  Equal shard sizes use one
    all_gather_into_tensor straight into x_all (no staging); ragged shards gather padded blocks."""
    sizes = [hi - lo for lo, hi in ranges]
    if len(set(sizes)) == 1:
        dist.all_gather_into_tensor(x_all, x_local.contiguous())
        return x_all
    import torch
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(x_local.shape[1:]), dtype=x_local.dtype, device=x_local.device)
    pad[: x_local.shape[0]].copy_(x_local)
    bufs = [torch.empty_like(pad) for _ in sizes]
    dist.all_gather(bufs, pad)
    for (lo, hi), b in zip(ranges, bufs):
        x_all[lo:hi].copy_(b[: hi - lo])
    return x_all


def all_reduce_status(n_failed_local: int, max_resid_local: float, device, dist):
    """Tiny all-reduce of (any failure, max residual) -- the outer loop's stop/raise decision."""
    import torch
    t = torch.tensor([float(n_failed_local), float(max_resid_local)], dtype=torch.float64, device=device)
    s = t.clone()
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(s[0].item()), float(t[1].item())
