"""Work counters / shader cycles per phase of the meshlet rasteriser (a build of smj_render.hip with -DSMJ_MESHLET_STATS, selected
with SMJ_LIB_PATH; the library prints one line per dynamic render).
   SMJ_LIB_PATH=.../build/exp/libsmj_mlstats.so python tools/gpu_meshlet_stats.py [scene]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from stretch_mujoco_amd import StretchBatchSimulator
from stretch_mujoco_amd.enums import StretchCameras
scene = sys.argv[1] if len(sys.argv) > 1 else "stretch_kitchen_standin"
B = 4096
sim = StretchBatchSimulator(num_envs=B, device="cuda:0", scene=scene, cameras_to_use=StretchCameras.depth()); sim.start(home=True)
g = torch.Generator(device=sim.device).manual_seed(1234)
lo = torch.tensor(np.asarray(sim.model["actuator_ctrlrange"], np.float32)[:, 0], device=sim.device)
hi = torch.tensor(np.asarray(sim.model["actuator_ctrlrange"], np.float32)[:, 1], device=sim.device)
sim.ctrl[:] = lo[:, None] + (hi - lo)[:, None] * torch.rand(sim.nu, B, generator=g, device=sim.device)
sim.step(400)
sim.pull_camera_data()
sim.pull_camera_data()
torch.cuda.synchronize()
