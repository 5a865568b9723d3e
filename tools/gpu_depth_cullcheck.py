"""Depth renderer self-check on the GPU: the image with the per-tile rectangle culling must equal, bit for bit, the image with the
culling switched off (SMJ_DEPTH_NOCULL=1 in a child process).  Random robot poses, both depth cameras."""
import os
import subprocess
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from stretch_mujoco_amd import StretchBatchSimulator  # noqa: E402
from stretch_mujoco_amd.enums import StretchCameras  # noqa: E402


def render(B, scene):
    sim = StretchBatchSimulator(num_envs=B, device="cuda:0", cameras_to_use=StretchCameras.depth(), scene=scene)
    sim.start(home=True)
    g = torch.Generator(device=sim.device).manual_seed(4321)
    lo = torch.tensor(np.asarray(sim.model["actuator_ctrlrange"], np.float32)[:, 0], device=sim.device)
    hi = torch.tensor(np.asarray(sim.model["actuator_ctrlrange"], np.float32)[:, 1], device=sim.device)
    sim.ctrl[:] = lo[:, None] + (hi - lo)[:, None] * torch.rand(sim.nu, B, generator=g, device=sim.device)
    sim.step(300)
    d = sim.pull_camera_data()
    out = {c.name: getattr(d, c.name).cpu().numpy().copy() for c in StretchCameras.depth()}
    sim.stop()
    return out


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    scene = sys.argv[2] if len(sys.argv) > 2 else "stretch_kitchen_standin"
    if os.environ.get("SMJ_DEPTH_CHILD"):
        np.savez(os.environ["SMJ_DEPTH_CHILD"], **render(B, scene))
        sys.exit(0)
    a = render(B, scene)
    path = "/tmp/smj_depth_nocull.npz"
    subprocess.check_call([sys.executable, __file__, str(B), scene], env=dict(os.environ, SMJ_DEPTH_NOCULL="1", SMJ_DEPTH_CHILD=path))
    b = np.load(path)
    for k in a:
        diff = np.argwhere(a[k] != b[k])
        print(k, "pixels", a[k].size, "differing", len(diff))
        for e, v, u in diff[:10]:
            print("   env", e, "pixel", (u, v), "culled", a[k][e, v, u], "unculled", b[k][e, v, u])
