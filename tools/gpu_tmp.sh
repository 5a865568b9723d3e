#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for lib in default xcd; do for sp in 4 8 16; do
if [ $lib = xcd ]; then export SMJ_LIB_PATH=$PWD/stretch_mujoco_amd/csrc/build/exp/libsmj_xcd.so; fi
python - $lib $sp <<'PY'
import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from stretch_mujoco_amd import StretchBatchSimulator
from stretch_mujoco_amd.enums import StretchCameras
B=4096
sim = StretchBatchSimulator(num_envs=B, device="cuda:0", cameras_to_use=StretchCameras.depth(), scene="stretch_kitchen_standin"); sim.start(home=True)
g = torch.Generator(device=sim.device).manual_seed(1234)
lo = torch.tensor(np.asarray(sim.model["actuator_ctrlrange"], np.float32)[:, 0], device=sim.device)
hi = torch.tensor(np.asarray(sim.model["actuator_ctrlrange"], np.float32)[:, 1], device=sim.device)
sim.ctrl[:] = lo[:, None] + (hi - lo)[:, None] * torch.rand(sim.nu, B, generator=g, device=sim.device)
sim.step(400)
sim.set_option("depth_raster_splits", int(sys.argv[2]))
sim.pull_camera_data(); torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(4): d=sim.pull_camera_data()
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/4*1e3
sim.set_option("depth_raster", 0); r=sim.pull_camera_data(); 
bad=[float(((getattr(d,c.name)-getattr(r,c.name)).abs() > 1e-4*getattr(r,c.name).abs()).float().mean()) for c in StretchCameras.depth()]
print("lib", sys.argv[1], "splits", sys.argv[2], "ms per render of both cameras %.2f" % dt, "mismatch vs ray caster", bad, flush=True)
PY
done; done 2>&1 | grep "ms per"
