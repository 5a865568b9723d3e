"""The two-wavefront Newton kernel of the 16-satellite build (smj_kernels_sat2.hip) beside the one-wavefront one: the same rollout
on both (states compared bit for bit), then the throughput of each at 4096 envs (random actions every 50 steps, as bench.py)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from stretch_mujoco_amd import StretchBatchSimulator  # noqa: E402


def rollout_pair(scene, B=256, launches=6):
    sims = []
    for two in (1, 0):
        sim = StretchBatchSimulator(num_envs=B, device="cuda:0", scene=scene)
        sim.start(home=False)
        sim.set_option("newton_two_waves", two)
        sim.home(settle=False)
        sims.append(sim)
    a, b = sims
    cr = torch.tensor(np.asarray(a.model["actuator_ctrlrange"]), dtype=torch.float32, device=a.device)
    g = torch.Generator(device=a.device); g.manual_seed(5)
    for w in range(launches):
        if w:
            a.ctrl[:] = cr[:, :1] + (cr[:, 1:] - cr[:, :1]) * torch.rand(a.nu, B, generator=g, device=a.device)
            b.ctrl[:] = a.ctrl
        a.step(50); b.step(50)
        torch.cuda.synchronize()
        dq = (a.qpos - b.qpos).abs().amax().item(); dv = (a.qvel - b.qvel).abs().amax().item()
        same = bool(torch.equal(a.qpos, b.qpos) and torch.equal(a.qvel, b.qvel))
        print(f"[{scene}] launch {w}: identical {same}, max |dqpos| {dq:.2e} |dqvel| {dv:.2e}; ncon mean {a.info[1].float().mean():.1f} / {b.info[1].float().mean():.1f}; "
              f"flags {int(a.info[3].ne(0).sum())} / {int(b.info[3].ne(0).sum())}", flush=True)
    for s in sims:
        s.stop()


def rate(scene, two, B=4096, n=400, hold=50):
    sim = StretchBatchSimulator(num_envs=B, device="cuda:0", scene=scene)
    sim.start(home=False)
    sim.set_option("newton_two_waves", two)
    sim.home(settle=False)
    cr = torch.tensor(np.asarray(sim.model["actuator_ctrlrange"]), dtype=torch.float32, device=sim.device)
    g = torch.Generator(device=sim.device); g.manual_seed(99)
    sim.step(500)
    torch.cuda.synchronize()
    t = time.perf_counter(); sim.step(200); torch.cuda.synchronize()
    settled = B * 200 / (time.perf_counter() - t)
    for _ in range(4):
        sim.ctrl[:] = cr[:, :1] + (cr[:, 1:] - cr[:, :1]) * torch.rand(sim.nu, B, generator=g, device=sim.device)
        sim.step(hold)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n // hold):
        sim.ctrl[:] = cr[:, :1] + (cr[:, 1:] - cr[:, :1]) * torch.rand(sim.nu, B, generator=g, device=sim.device)
        sim.step(hold)
    torch.cuda.synchronize()
    r = B * n / (time.perf_counter() - t)
    print(f"[{scene}] newton_two_waves {two}: settled {settled / 1e6:.3f} M, random actions {r / 1e6:.3f} M env-steps/s; flags {int(sim.info[3].ne(0).sum())}", flush=True)
    sim.stop()


if __name__ == "__main__":
    scenes = sys.argv[1:] or ["stretch_kitchen_robocasa", "stretch_kitchen4_sat"]
    for sc in scenes:
        rollout_pair(sc)
    for sc in scenes:
        for two in (0, 1, 0, 1):
            rate(sc, two)
