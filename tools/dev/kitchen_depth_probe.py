import sys, time, torch
sys.path.insert(0, ".")
from stretch_mujoco_amd import StretchBatchSimulator
from stretch_mujoco_amd.enums import StretchCameras
B = 4096
sim = StretchBatchSimulator(num_envs=B, device="cuda:0", scene="stretch_kitchen_robocasa", cameras_to_use=StretchCameras.depth())
sim.start(home=False); sim.step(100); sim.pull_camera_data(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(3): sim.pull_camera_data()
torch.cuda.synchronize(); print("render ms", (time.perf_counter() - t) / 3 * 1e3)
t = time.perf_counter(); n = 0
for _ in range(6): sim.step(17); sim.pull_camera_data(); n += 17
torch.cuda.synchronize(); print("loop M env-steps/s", B * n / (time.perf_counter() - t) / 1e6)
