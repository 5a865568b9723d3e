"""Timing experiment: does a render of both depth cameras overlap with the physics launch when issued on a second stream?
(No hazard handling: the render may read poses the concurrent step is rewriting -- times only.)"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stretch_mujoco_amd import StretchBatchSimulator
from stretch_mujoco_amd.enums import StretchCameras

B = 4096
dev = torch.device("cuda:0")
sim = StretchBatchSimulator(num_envs=B, device=str(dev), solver="newton", scene="stretch_kitchen_standin", cameras_to_use=StretchCameras.depth())
sim.start(home=False)
lo = torch.tensor(sim.model["actuator_ctrlrange"][:, 0], dtype=torch.float32, device=dev).unsqueeze(1)
hi = torch.tensor(sim.model["actuator_ctrlrange"][:, 1], dtype=torch.float32, device=dev).unsqueeze(1)
g = torch.Generator(device=dev); g.manual_seed(1)
def act():
    sim.ctrl[:] = lo + (hi - lo) * torch.rand(sim.ctrl.shape, device=dev, generator=g)
for _ in range(6):
    act(); sim.step(50)
sim.pull_camera_data()
torch.cuda.synchronize()
def timed(f, n=6):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
print("physics 17 steps: %.2f ms" % timed(lambda: sim.step(17)))
print("render both cameras: %.2f ms" % timed(lambda: sim.pull_camera_data()))
def serial():
    sim.step(17); sim.pull_camera_data()
print("serial: %.2f ms" % timed(serial))
s2 = torch.cuda.Stream(device=dev)
def overlapped(first):
    ev = torch.cuda.Event(); ev.record()
    if first == "render":
        with torch.cuda.stream(s2):
            s2.wait_event(ev); sim.pull_camera_data()
        sim.step(17)
    else:
        sim.step(17)
        with torch.cuda.stream(s2):
            s2.wait_event(ev); sim.pull_camera_data()
    torch.cuda.current_stream().wait_stream(s2)
print("overlapped, render issued first: %.2f ms" % timed(lambda: overlapped("render")))
print("overlapped, physics issued first: %.2f ms" % timed(lambda: overlapped("physics")))
