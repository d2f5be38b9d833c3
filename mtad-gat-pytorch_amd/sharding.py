"""Data-parallel sharding of sliding windows across the GPUs of a node.

Windows are independent units: rank r gets the contiguous block
[r*M/N, (r+1)*M/N) of window indices, runs the forward on it with its own replica of the
(1.7 MB) weights, and no collective touches the data path.  `torch.distributed` (backend "nccl"
= RCCL on ROCm, "gloo" in the CPU tests) is used only for the timing barrier / max-over-ranks and,
optionally, to gather per-window outputs.
"""
import torch
import torch.distributed as dist


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous block of window indices owned by `rank` (sizes differ by at most one)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_over_ranks(seconds: float, device=None) -> float:
    """Slowest rank's time (the job's time); identity when not running distributed."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_windows(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """Concatenate per-rank outputs (first dim = this rank's windows) back into window order."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    sizes = [hi - lo for lo, hi in (shard_range(n_total, r, world) for r in range(world))]
    pad = max(sizes)          # all_gather wants equal shapes: pad the short shards, trim after
    mine = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    mine[: local.shape[0]] = local
    bufs = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(bufs, mine)
    return torch.cat([b[:n] for b, n in zip(bufs, sizes)], dim=0)
