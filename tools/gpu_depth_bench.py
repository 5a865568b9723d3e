"""Time the depth-camera kernel (smj_render_depth) on the GPU: B envs, both depth cameras, random head/arm poses."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from stretch_mujoco_amd import StretchBatchSimulator  # noqa: E402
from stretch_mujoco_amd.enums import StretchCameras  # noqa: E402


def main(B=4096, reps=5):
    sim = StretchBatchSimulator(num_envs=B, device="cuda:0", cameras_to_use=StretchCameras.depth())
    sim.start(home=True)
    g = torch.Generator(device=sim.device).manual_seed(1234)
    lo = torch.tensor(np.asarray(sim.model["actuator_ctrlrange"], np.float32)[:, 0], device=sim.device)
    hi = torch.tensor(np.asarray(sim.model["actuator_ctrlrange"], np.float32)[:, 1], device=sim.device)
    sim.ctrl[:] = lo[:, None] + (hi - lo)[:, None] * torch.rand(sim.nu, B, generator=g, device=sim.device)
    sim.step(400)
    torch.cuda.synchronize()
    out = {}
    for cam in StretchCameras.depth():
        sim._cameras = [cam]
        sim.pull_camera_data()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            img = getattr(sim.pull_camera_data(), cam.name)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / reps
        st = cam.initial_camera_settings
        rays = B * st.width * st.height
        out[cam.name] = dict(ms=dt * 1e3, grays_per_s=rays / dt / 1e9, write_GBps=rays * 4 / dt / 1e9,
                             nonzero_frac=float((img > 0).float().mean()))
    print(json.dumps(dict(B=B, **out)))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 4096)
