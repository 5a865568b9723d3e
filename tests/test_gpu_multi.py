"""Two ranks on two GPUs of one node: envs shard, each rank steps its own shard, the per-env returns
are gathered by the library's own RCCL all-gather (smj_allgather_returns inside libsmj.so) and must equal the
torch.distributed gather and the single-process result.  Skipped on a box with fewer than two GPUs (gpurun leases one):
the CPU twin of this test is tests/test_distributed.py (gloo)."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rank(rank, world, port, out, total):
    import torch.distributed as dist

    from stretch_mujoco_amd import StretchBatchSimulator
    from stretch_mujoco_amd.parallel import gather_returns, gather_returns_native, shard_range

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    lo, hi = shard_range(total, rank, world)
    sim = StretchBatchSimulator(num_envs=hi - lo, device=f"cuda:{rank}", solver="newton")
    sim.start(home=False)
    ids = torch.arange(lo, hi, device=sim.device, dtype=torch.float32)
    sim.ctrl[:] = 0.0
    sim.ctrl[2] = 0.3 + 0.5 * ids / total          # lift target differs per GLOBAL env id: the shards are distinguishable
    sim.ctrl[3] = 0.1
    sim.step(200)
    local = sim.pull_status().lift.pos.float().contiguous()
    native, how = gather_returns_native(sim, local)
    via_torch = gather_returns(local)
    torch.cuda.synchronize()
    if rank == 0:
        torch.save(dict(native=native.cpu(), via_torch=via_torch.cpu(), how=how), out)
    dist.barrier()
    sim.stop()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs on the node (gpurun leases one)")
def test_two_ranks_gather_returns_through_the_library(tmp_path):
    import torch.multiprocessing as mp

    from stretch_mujoco_amd import StretchBatchSimulator

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    total, out = 128, str(tmp_path / "g.pt")
    mp.spawn(_rank, args=(2, port, out, total), nprocs=2, join=True)
    r = torch.load(out)
    assert "RCCL" in r["how"] and r["native"].numel() == total
    assert torch.equal(r["native"], r["via_torch"])
    # the same 128 envs in one process on one GPU: sharding changes nothing (no coupling between envs)
    sim = StretchBatchSimulator(num_envs=total, device="cuda:0", solver="newton")
    sim.start(home=False)
    ids = torch.arange(total, device=sim.device, dtype=torch.float32)
    sim.ctrl[:] = 0.0
    sim.ctrl[2] = 0.3 + 0.5 * ids / total
    sim.ctrl[3] = 0.1
    sim.step(200)
    one = sim.pull_status().lift.pos.float().cpu()
    sim.stop()
    assert torch.equal(one, r["native"])
