"""Proof-index sharding of a batch over ranks and the one exchange step of the path (mode A).

Proofs are independent units (reference src/range/mod.rs:503-509), so a batch of `count` proofs is cut into
contiguous blocks, one per rank, and nothing is exchanged on the data path.  The only collective is the
batch verdict: an all-reduce (SUM) of the per-rank failure counts.  Works with any torch.distributed
backend ("nccl" = RCCL on MI355X; "gloo" in the CPU tests)."""

from __future__ import annotations


def shard_bounds(count: int, world: int, rank: int):
    """Contiguous block [lo, hi) of proof indices for `rank`; sizes differ by at most one."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(count, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def batch_verdict(local_ok, dist=None, device=None):
    """local_ok: 1-D integer tensor of this rank's per-proof verdicts (0 = Ok, 1 = VerificationError).
    Returns (total_failures over all ranks, batch_ok)."""
    import torch
    fails = (local_ok != 0).sum().to(torch.int64).reshape(1)
    if device is not None:
        fails = fails.to(device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(fails, op=dist.ReduceOp.SUM)
    total = int(fails.item())
    return total, total == 0


def gather_verdicts(local_ok, count: int, dist=None):
    """All ranks obtain the full per-proof verdict vector (used by tests; the bench only needs the sum)."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local_ok.clone()
    world = dist.get_world_size()
    sizes = [shard_bounds(count, world, r) for r in range(world)]
    maxlen = max(hi - lo for lo, hi in sizes)
    pad = torch.full((maxlen,), -1, dtype=local_ok.dtype, device=local_ok.device)
    pad[: local_ok.numel()] = local_ok
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(outs, sizes)])
