"""Multi-GPU: environments shard embarrassingly (one process per GPU, contiguous env blocks, no exchange while
stepping -- the reference itself is one env per process with no coupling, stretch_mujoco_simulator.py:102-118).

The only collective on the path is an all-gather of per-env returns.  On the GPU it goes through the library's own
`smj_allgather_returns` (RCCL's ncclAllGather over xGMI, called inside libsmj.so on the caller's stream, include/smj.h);
`gather_returns` is the torch.distributed twin (backend "nccl" = RCCL on ROCm, gloo in the CPU tests) that the tests use as
the comparator.
"""
from __future__ import annotations

import ctypes
import os
import secrets
import tempfile

import torch

from . import lib as _lib


def shard_range(total_envs: int, rank: int, world: int):
    """Contiguous block of env ids owned by `rank`; the first `total % world` ranks hold one more."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"rank {rank} / world {world}")
    per = total_envs // world
    extra = total_envs % world
    lo = rank * per + min(rank, extra)
    return lo, lo + per + (1 if rank < extra else 0)


def shard_sizes(total_envs: int, world: int):
    return [shard_range(total_envs, r, world)[1] - shard_range(total_envs, r, world)[0] for r in range(world)]


def gather_returns(local_returns: torch.Tensor) -> torch.Tensor:
    """All ranks' per-env returns, rank-major, through torch.distributed.  Single-process: a copy."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_returns.clone()
    world = dist.get_world_size()
    out = torch.empty(world * local_returns.numel(), dtype=local_returns.dtype, device=local_returns.device)
    dist.all_gather_into_tensor(out, local_returns.contiguous())
    return out


def _job_token() -> str:
    """A token every rank of the job agrees on: rank 0 draws it, torch.distributed carries it (object broadcast)."""
    import torch.distributed as dist

    tok = [secrets.token_hex(8) if dist.get_rank() == 0 else None]
    dist.broadcast_object_list(tok, src=0)
    return tok[0]


def init_comm(sim, rank: int | None = None, world: int | None = None, id_path: str | None = None, timeout_s: float = 120.0) -> int:
    """Join `sim`'s context to the job's RCCL communicator (smj_comm_init).  rank / world default to torch.distributed's (or
    RANK / WORLD_SIZE); the ncclUniqueId travels through the file `id_path`, which defaults to a per-job path in the temp
    directory agreed on over torch.distributed.  Returns the world size."""
    import torch.distributed as dist

    have_dist = dist.is_available() and dist.is_initialized()
    if rank is None:
        rank = dist.get_rank() if have_dist else int(os.environ.get("RANK", "0"))
    if world is None:
        world = dist.get_world_size() if have_dist else int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and id_path is None:
        if not have_dist:
            raise ValueError("init_comm: pass id_path (a file path shared by all ranks) when torch.distributed is not initialised")
        id_path = os.path.join(tempfile.gettempdir(), f"smj_rccl_id_{_job_token()}")
    if world > 1 and have_dist:
        # a file an earlier job left at this path must be gone before any rank looks for the new one (the library also checks
        # the record's job nonce and lets rank 0 remove the path first; this barrier closes the window for good)
        if rank == 0:
            try:
                os.remove(id_path)
            except OSError:
                pass
        dist.barrier()
    rc = sim._L.smj_comm_init(sim._ctx, int(rank), int(world), id_path.encode() if id_path else None, float(timeout_s))
    _lib.check(sim._L, sim._ctx, rc, "smj_comm_init")
    sim._comm_world = world
    if world > 1 and have_dist:
        dist.barrier()
        if rank == 0:
            try:
                os.remove(id_path)
            except OSError:
                pass
    return world


def allgather_returns(sim, local_returns: torch.Tensor) -> torch.Tensor:
    """All ranks' per-env returns, rank-major, by the library's RCCL all-gather on the current stream."""
    world = getattr(sim, "_comm_world", 1)
    send = local_returns.contiguous().to(torch.float32)
    out = torch.empty(world * send.numel(), dtype=torch.float32, device=send.device)
    stream = ctypes.c_void_p(torch.cuda.current_stream(send.device).cuda_stream)
    rc = sim._L.smj_allgather_returns(sim._ctx, ctypes.c_void_p(send.data_ptr()), ctypes.c_void_p(out.data_ptr()), send.numel(), stream)
    _lib.check(sim._L, sim._ctx, rc, "smj_allgather_returns")
    return out


def gather_returns_native(sim, local_returns: torch.Tensor):
    """bench.py's gather: the C-ABI path (RCCL inside libsmj.so); needs equal shard sizes (ncclAllGather).  Returns
    (tensor, description of the path taken)."""
    import torch.distributed as dist

    world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    if world > 1:
        nmax = torch.tensor([local_returns.numel()], device=local_returns.device)
        dist.all_reduce(nmax, op=dist.ReduceOp.MAX)
        if int(nmax.item()) != local_returns.numel():
            pad = torch.zeros(int(nmax.item()), dtype=local_returns.dtype, device=local_returns.device)
            pad[: local_returns.numel()] = local_returns
            local_returns = pad   # ragged shards: padded to the largest shard
        if getattr(sim, "_comm_world", 1) != world:
            init_comm(sim)
    how = "RCCL ncclAllGather inside libsmj.so" if world > 1 else "world 1: a device-to-device copy, no collective"
    return allgather_returns(sim, local_returns), f"smj_allgather_returns ({how}), world {world}"
