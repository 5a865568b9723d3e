"""bench.py's N-rank control flow on the CPU (VERDICT r5 "next" 7): shards, the barrier / synchronize brackets, the max over ranks, the
gather of per-env returns, `ranks_seen`, and the strong / weak arithmetic of the JSON line -- world 2 and 4 under gloo with a stand-in
for the simulator (a step costs wall time in proportion to envs x steps, rank 1 is the slow one).  No GPU, no libsmj.so: what is tested
is bench.measure_ranks / line_head, the code the driver's `torch.distributed.run ... bench.py --gpus N` executes on every rank."""
import argparse
import json
import os
import socket
import sys
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SEC_PER_ENV_STEP = 2e-7   # the stand-in's cost: 4096 envs x 50 steps = 41 ms


class FakeSim:
    """What measure_one touches of StretchBatchSimulator."""

    def __init__(self, B, rank):
        self.num_envs, self.nu, self.rank = B, 10, rank
        self.model = {"actuator_ctrlrange": np.stack([-np.ones(10), np.ones(10)], 1), "key_ctrl": np.zeros((1, 10))}
        self.ctrl = torch.zeros(10, B)
        self.base_pose = torch.zeros(3, B)
        self.info = torch.zeros(4, B, dtype=torch.int32)
        self.nstep = torch.zeros(B, dtype=torch.int32)
        self.stopped = False

    def step(self, k):
        time.sleep(SEC_PER_ENV_STEP * self.num_envs * k * (1.5 if self.rank == 1 else 1.0))   # rank 1 is 50 % slower: the max over ranks must show it
        self.nstep += k
        self.base_pose[0] += 1e-3 * k * (1 + self.rank)

    def stop(self):
        self.stopped = True


class WallEvent:
    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


def _worker(rank, world, port, out, scaling, steps):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from stretch_mujoco_amd import parallel

    sims = []

    class Hooks:
        device = torch.device("cpu")

        @staticmethod
        def make_sim(B):
            sims.append(FakeSim(B, rank))
            return sims[-1]

        sync = staticmethod(lambda: None)
        event = staticmethod(WallEvent)
        barrier = staticmethod(dist.barrier)

        @staticmethod
        def max_over_ranks(x):
            t = torch.tensor([x], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        @staticmethod
        def gather(sim, returns):
            return parallel.gather_returns(returns), "gloo all_gather_into_tensor (CPU test)"

    args = argparse.Namespace(envs=4096, hold=50, steps=steps, warmup=50, scaling=scaling, solver="newton")
    m = bench.measure_ranks(args, rank, world, Hooks)
    head = bench.line_head(args, world, m)
    own = sum(a.elapsed_time(b) for a, b, _ in m["events"]) / 1e3
    res = dict(rank=rank, head=head, own_seconds=own, B=m["B"], nsims=len(sims), other_stopped=all(s.stopped for s in sims[1:]),
               returns_sum=float(m["all_returns"].sum()), nstep=int(m["sim"].nstep[0]))
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    if rank == 0:
        with open(out, "w") as f:
            json.dump(gathered, f)
    dist.barrier()
    dist.destroy_process_group()


def _run(tmp_path, world, scaling, steps):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / f"ranks_{world}_{scaling}.json")
    mp.spawn(_worker, args=(world, port, out, scaling, steps), nprocs=world, join=True)
    with open(out) as f:
        return json.load(f)


def test_bench_line_at_world_2_strong_is_4096_envs_in_total_with_the_weak_figure_beside_it(tmp_path):
    ranks = _run(tmp_path, 2, "strong", 100)
    h = ranks[0]["head"]
    assert h["scaling"] == "strong" and h["n_gpus"] == 2 and h["metric"].endswith("in total")
    c = h["config"]
    assert c["envs_total"] == 4096 and c["envs_per_gpu"] == 2048 and c["ranks_seen"] == 2 and c["returns_gathered"] == 4096
    assert [r["B"] for r in ranks] == [2048, 2048]
    # value = all envs x K steps / the SLOWEST rank's bracket: rank 1's kernels take 1.5 x rank 0's, and every rank reports the same maximum
    assert all(abs(r["head"]["ms_per_step"] - h["ms_per_step"]) < 1e-9 for r in ranks)
    assert h["ms_per_step"] * 100 / 1e3 >= ranks[1]["own_seconds"] > 1.3 * ranks[0]["own_seconds"]
    assert abs(h["value"] - 4096 * 100 / (h["ms_per_step"] * 100 / 1e3)) < 1e-6 * h["value"]
    # the weak figure of the same ranks: 4096 envs per GPU, 8192 in total, every rank seen, its sims stopped again
    w = h["weak"]
    assert w["scaling"] == "weak" and w["envs_total"] == 8192 and w["envs_per_gpu"] == 4096 and w["ranks_seen"] == 2
    assert abs(w["value"] - 8192 * 100 / (w["ms_per_step"] * 100 / 1e3)) < 1e-6 * w["value"]
    assert 1.3 < w["ms_per_step"] / h["ms_per_step"] < 4.5          # twice the envs per rank on the stand-in: twice the time (the slow rank bounds both)
    assert all(r["nsims"] == 2 and r["other_stopped"] for r in ranks)
    # exactly warm-up + K steps after the 500 settle + 200 pre-roll on every rank; the gathered returns are rank-major and complete
    assert all(r["nstep"] == 500 + 200 + 50 + 100 for r in ranks)
    # (the synthetic return adds the base's x after every timed launch: 0.80 + 0.85 on rank 0's envs, twice that on rank 1's)
    assert abs(ranks[0]["returns_sum"] - (2048 * 1.65 * 1 + 2048 * 1.65 * 2)) < 1e-1


def test_bench_line_at_world_4_short_region_takes_the_median_repetition(tmp_path):
    ranks = _run(tmp_path, 4, "strong", 20)      # K below one action interval: three brackets of exactly K steps, the median reported
    h = ranks[0]["head"]
    c = h["config"]
    assert c["envs_total"] == 4096 and c["envs_per_gpu"] == 1024 and c["ranks_seen"] == 4 and len(c["timed_region_ms"]) == 3
    assert abs(h["ms_per_step"] * 20 - sorted(c["timed_region_ms"])[1]) < 2e-3
    assert h["weak"]["envs_total"] == 4 * 4096 and h["weak"]["ranks_seen"] == 4
    assert all(r["B"] == 1024 for r in ranks)


def test_weak_as_the_headline_when_asked(tmp_path):
    ranks = _run(tmp_path, 2, "weak", 50)
    h = ranks[0]["head"]
    assert h["scaling"] == "weak" and h["config"]["envs_total"] == 8192 and h["config"]["envs_per_gpu"] == 4096 and h["metric"].endswith("per MI355X")
    assert h["strong"]["envs_total"] == 4096 and h["strong"]["envs_per_gpu"] == 2048
