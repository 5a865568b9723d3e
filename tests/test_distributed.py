"""Multi-rank path on CPU: gloo, world_size 2.  Envs shard contiguously; the only collective is the all-gather of
per-env returns (RCCL on the GPU box)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from stretch_mujoco_amd.parallel import gather_returns, shard_range


def test_shard_range_partitions_exactly():
    for total, world in ((4096, 8), (4096, 3), (10, 4), (7, 8)):
        spans = [shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        for (a, b), (c, d) in zip(spans, spans[1:]):
            assert b == c and b >= a
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(16, rank, world)
    local = torch.arange(lo, hi, dtype=torch.float32) * 10 + rank
    allr = gather_returns(local)
    # max-over-ranks timing as bench.py does it
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        torch.save(dict(allr=allr, tmax=float(t)), out)
    dist.barrier()
    dist.destroy_process_group()


def test_gather_returns_world2(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "r.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out)
    exp = torch.cat([torch.arange(0, 8, dtype=torch.float32) * 10, torch.arange(8, 16, dtype=torch.float32) * 10 + 1])
    assert torch.equal(r["allr"], exp) and r["tmax"] == 2.0


def test_gather_returns_single_process_is_identity():
    x = torch.arange(5.0)
    y = gather_returns(x)
    assert torch.equal(x, y) and y.data_ptr() != x.data_ptr()
