"""Multi-GPU: environments shard embarrassingly (one process per GPU, contiguous env blocks, no exchange while
stepping).  The only collective on the path is an all-gather of per-env returns (RCCL over xGMI on ROCm:
torch.distributed backend "nccl"; gloo in the CPU tests)."""
from __future__ import annotations

import torch


def shard_range(total_envs: int, rank: int, world: int):
    """Contiguous block of env ids owned by `rank`."""
    per = total_envs // world
    extra = total_envs % world
    lo = rank * per + min(rank, extra)
    return lo, lo + per + (1 if rank < extra else 0)


def gather_returns(local_returns: torch.Tensor) -> torch.Tensor:
    """All ranks' per-env returns, rank-major.  Single-process: identity."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_returns.clone()
    world = dist.get_world_size()
    out = torch.empty(world * local_returns.numel(), dtype=local_returns.dtype, device=local_returns.device)
    dist.all_gather_into_tensor(out, local_returns.contiguous())
    return out
