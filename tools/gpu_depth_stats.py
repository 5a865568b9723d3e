"""Work counters of the depth kernel (a build of smj_render.hip with -DSMJ_DEPTH_STATS, selected with SMJ_LIB_PATH): per ray, the
geoms in the tile's list, those past the bounding-sphere reject, mesh walks, inner-node visits and leaves, per camera.
   SMJ_LIB_PATH=.../libsmj_depthstats.so python tools/gpu_depth_stats.py [scene]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from stretch_mujoco_amd import StretchBatchSimulator
from stretch_mujoco_amd.enums import StretchCameras
NAMES = ["geoms in tile list", "past sphere reject", "mesh walks", "node visits", "leaves"]
scene = sys.argv[1] if len(sys.argv) > 1 else "stretch_kitchen_standin"
res = {}
B = 256
sim = StretchBatchSimulator(num_envs=B, device="cuda:0", scene=scene, cameras_to_use=StretchCameras.depth()); sim.start(home=True)
g = torch.Generator(device=sim.device).manual_seed(1234)
lo = torch.tensor(np.asarray(sim.model["actuator_ctrlrange"], np.float32)[:, 0], device=sim.device)
hi = torch.tensor(np.asarray(sim.model["actuator_ctrlrange"], np.float32)[:, 1], device=sim.device)
sim.ctrl[:] = lo[:, None] + (hi - lo)[:, None] * torch.rand(sim.nu, B, generator=g, device=sim.device)
sim.step(400)
for k in range(5):
    os.environ["SMJ_DEPTH_STAT"] = str(k)   # (the library reads the variable at every launch)
    cd = sim.pull_camera_data()
    for cam in StretchCameras.depth():
        img = getattr(cd, cam.name)
        res.setdefault(cam.name, []).append((float(img.mean()), float(img.max())))
for cam, v in res.items():
    print(scene, cam, {n: f"mean {a:.2f} max {b:.0f}" for n, (a, b) in zip(NAMES, v)})
