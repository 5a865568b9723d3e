"""Time of one render of both depth cameras (4096 envs at random-action poses), kitchen stand-in and empty scene.
   [SMJ_LIB_PATH=...] python tools/gpu_render_time.py [tag]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from stretch_mujoco_amd import StretchBatchSimulator
from stretch_mujoco_amd.enums import StretchCameras
tag = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("SMJ_LIB_PATH", "default")
B = 4096
for scene in ("stretch_kitchen_standin", "stretch_empty"):
    sim = StretchBatchSimulator(num_envs=B, device="cuda:0", scene=scene, cameras_to_use=StretchCameras.depth()); sim.start(home=False)
    g = torch.Generator(device=sim.device).manual_seed(1234)
    lo = torch.tensor(np.asarray(sim.model["actuator_ctrlrange"], np.float32)[:, 0], device=sim.device)
    hi = torch.tensor(np.asarray(sim.model["actuator_ctrlrange"], np.float32)[:, 1], device=sim.device)
    for _ in range(6):
        sim.ctrl[:] = lo[:, None] + (hi - lo)[:, None] * torch.rand(sim.nu, B, generator=g, device=sim.device)
        sim.step(50)
    cd = sim.pull_camera_data()
    chk = [float(getattr(cd, c.name).double().sum()) for c in StretchCameras.depth()]
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): sim.pull_camera_data()
    torch.cuda.synchronize()
    print(tag, scene, "render both cameras: %.2f ms" % ((time.perf_counter() - t) / 5 * 1e3), "checksum", ["%.6e" % c for c in chk])
    sim.stop()
