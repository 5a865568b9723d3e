"""step(17) (one 30 Hz camera interval of physics) in the kitchen stand-in under different chunk lengths of the pipelined dispatch."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from stretch_mujoco_amd import StretchBatchSimulator
B = 4096
dev = torch.device("cuda:0")
sim = StretchBatchSimulator(num_envs=B, device=str(dev), solver="newton", scene="stretch_kitchen_standin")
sim.start(home=False)
lo = torch.tensor(sim.model["actuator_ctrlrange"][:, 0], dtype=torch.float32, device=dev).unsqueeze(1)
hi = torch.tensor(sim.model["actuator_ctrlrange"][:, 1], dtype=torch.float32, device=dev).unsqueeze(1)
g = torch.Generator(device=dev); g.manual_seed(1)
for _ in range(6):
    sim.ctrl[:] = lo + (hi - lo) * torch.rand(sim.ctrl.shape, device=dev, generator=g); sim.step(50)
def timed(n, reps=12):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): sim.step(n)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
for p in (5, 0, 2, 3, 4, 6, 9):
    sim.set_option("pipeline", p)
    print("pipeline", p, "step(17): %.2f ms" % timed(17), " step(50): %.2f ms" % timed(50, 4))
