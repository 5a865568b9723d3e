"""Workload for the rocprofv3 passes over the ray-casting kernels (tools/gpu_profile.sh): kitchen stand-in (or the scene named second), B envs at random
arm / head poses, a few renders of both depth cameras and a few lidar readouts."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from stretch_mujoco_amd import StretchBatchSimulator, StretchSensors  # noqa: E402
from stretch_mujoco_amd.enums import StretchCameras  # noqa: E402


def main(B=4096, reps=3, scene="stretch_kitchen_standin"):
    sim = StretchBatchSimulator(num_envs=B, device="cuda:0", scene=scene, cameras_to_use=StretchCameras.depth(),
                                sensors_to_use=StretchSensors.all())
    sim.start(home=False)
    g = torch.Generator(device=sim.device).manual_seed(1234)
    lo = torch.tensor(np.asarray(sim.model["actuator_ctrlrange"], np.float32)[:, 0], device=sim.device)
    hi = torch.tensor(np.asarray(sim.model["actuator_ctrlrange"], np.float32)[:, 1], device=sim.device)
    sim.ctrl[:] = torch.tensor(np.asarray(sim.model["key_ctrl"], np.float32)[0, : sim.nu], device=sim.device)[:, None]
    sim.step(300)
    sim.ctrl[:] = lo[:, None] + (hi - lo)[:, None] * torch.rand(sim.nu, B, generator=g, device=sim.device)
    sim.step(300)
    for _ in range(reps):
        sim.pull_camera_data()
        sim.step(1)          # one physics step + the lidar readout
    torch.cuda.synchronize()
    sim.stop()


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 4096, scene=sys.argv[2] if len(sys.argv) > 2 else "stretch_kitchen_standin")
