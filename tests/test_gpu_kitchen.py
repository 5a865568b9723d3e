"""Kitchen stand-in on the HIP path: trajectory, contacts, lidar and depth against the fp64 oracle (tolerances as in
test_gpu_parity.py / test_gpu_depth.py)."""
import json

import numpy as np
import pytest
import torch

from conftest import home_qpos
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu


def _sim(B, **kw):
    from stretch_mujoco_amd import StretchBatchSimulator

    sim = StretchBatchSimulator(num_envs=B, device="cuda:0", scene="stretch_kitchen_standin", solver="newton", **kw)
    sim.start(home=False)
    return sim


def test_arm_stalls_on_the_counter_like_the_oracle():
    from stretch_mujoco_amd import StretchSensors

    sim = _sim(8, sensors_to_use=[StretchSensors.base_lidar])
    ctrl = [0, 0, 0.6, 0.5, 0, 0, 0, 0, 0, 0]
    o = Oracle(sim._blob); o.set_option("solver", 2)
    o.arr("ctrl")[:] = ctrl
    q0 = home_qpos(o.arr("qpos").copy())
    o.arr("qpos")[:] = q0
    sim.qpos[:] = torch.tensor(q0, dtype=torch.float32, device=sim.device).unsqueeze(1)
    sim.ctrl[:] = torch.tensor(ctrl, dtype=torch.float32, device=sim.device).unsqueeze(1)
    # free flight until the gripper reaches the counter: tight parity
    o.step(60); sim.step(60)
    torch.cuda.synchronize()
    assert np.abs(sim.qpos[:, 0].cpu().numpy() - o.arr("qpos")).max() < 1e-5
    o.step(340); sim.step(340)
    torch.cuda.synchronize()
    arm_gpu = sim.qpos[10:14].sum(0).cpu().numpy()
    arm_ref = o.arr("qpos")[10:14].sum()
    assert 0.3 < arm_ref < 0.45 and np.abs(arm_gpu - arm_ref).max() < 0.02       # stalled at the same place
    assert int(sim.info[1].max()) > 5 and torch.equal(sim.qpos[:, 0], sim.qpos[:, 7])
    # lidar of the final pose against the oracle on the same pose
    o.arr("qpos")[:] = sim.qpos[:, 0].cpu().numpy().astype(np.float64)
    sim.step(1)
    torch.cuda.synchronize()
    o.forward(); o.sensors(True)
    L = sim.pull_sensor_data().lidar[0].cpu().numpy()
    bad = np.abs(L - o.arr("lidar")) > 1e-3
    assert bad.sum() <= 2 and (L > 0).all() and L.max() < 3.8
    sim.stop()


def test_depth_sees_the_fixtures():
    from stretch_mujoco_amd.enums import StretchCameras

    cams = StretchCameras.depth()
    sim = _sim(2, cameras_to_use=cams)
    o = Oracle(sim._blob)
    q = home_qpos(o.arr("qpos").copy())
    names = {n: i for i, n in enumerate(json.loads(bytes(sim.model["names_json"]).decode())["joint"])}
    adr = sim.model["jnt_qposadr"]
    q2 = q.copy()
    q2[adr[names["joint_head_pan"]]] = -1.5            # look toward the counter run
    q2[adr[names["joint_head_tilt"]]] = -0.5
    q2[adr[names["joint_lift"]]] = 1.0
    q32 = np.stack([q, q2], 1).astype(np.float32)
    sim.qpos[:] = torch.tensor(q32, device=sim.device)
    sim.step(1)
    imgs = sim.pull_camera_data()
    torch.cuda.synchronize()
    cam_names = json.loads(bytes(sim.model["names_json"]).decode())["camera"]
    for cam in cams:
        st = cam.initial_camera_settings
        g = getattr(imgs, cam.name).cpu().numpy()
        for e in range(2):
            o.arr("qpos")[:] = q32[:, e].astype(np.float64)
            o.forward()
            ref = o.render_depth(cam_names.index(cam.camera_name_in_mjcf), st.width, st.height,
                                 st.field_of_view_vertical_in_degrees, cam.depth_limit)
            ok = np.abs(g[e] - ref) <= 1e-4 + 1e-4 * np.abs(ref)
            assert 1.0 - ok.mean() < 5e-3, (cam, e, 1.0 - ok.mean())
    d435 = imgs.cam_d435i_depth[1].cpu().numpy()
    assert ((d435 > 0.5) & (d435 < 3.0)).mean() > 0.3      # cabinets / counter fill the head camera's view
    sim.stop()
