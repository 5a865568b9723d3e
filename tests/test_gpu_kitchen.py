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


def test_pipelined_chunks_on_the_tall_variant_are_bit_identical():
    """The kitchen stand-in runs the tall variant as its primary kernel; its launches are pipelined in chunks of 8 steps
    (smj_step, DevState::pipe_len).  Scheduling only: 2048 envs under random actions, 2 x 37 steps -- every state word and the
    sensor readouts equal those of the one-workgroup-per-env launch."""
    from stretch_mujoco_amd import StretchSensors

    B, ref = 2048, None
    for pipe in (0, 5, 3):
        sim = _sim(B, sensors_to_use=StretchSensors.all())
        sim.set_option("pipeline", pipe)
        g = torch.Generator(device=sim.device).manual_seed(11)
        lo = torch.tensor(sim.model["actuator_ctrlrange"][:, 0], dtype=torch.float32, device=sim.device).unsqueeze(1)
        hi = torch.tensor(sim.model["actuator_ctrlrange"][:, 1], dtype=torch.float32, device=sim.device).unsqueeze(1)
        sim.ctrl[:] = torch.tensor([0, 0, 0.6, 0.1, 0, 0, 0, 0, 0, 0], dtype=torch.float32, device=sim.device).unsqueeze(1)
        sim.step(200)
        for _ in range(2):
            sim.ctrl.copy_(lo + (hi - lo) * torch.rand(10, B, generator=g, device=sim.device))
            sim.step(37)
        torch.cuda.synchronize()
        got = [t.clone() for t in (sim.qpos, sim.qvel, sim.qacc_warmstart, sim.actuator_length, sim.base_pose, sim.info, sim.nstep, sim.gyro, sim.accel, sim.lidar, sim.xpose)]
        assert int(sim.nstep.min()) == int(sim.nstep.max()) == 274 and (int(sim.info[3].max()) & 8) == 0
        if ref is None:
            ref = got
        else:
            for a, b in zip(ref, got):
                assert torch.equal(a, b), pipe
        sim.stop()


def test_three_envs_per_cu_build_hands_over_to_the_160_row_build():
    """The kitchen stand-in's primary kernel is the 128-row / 44-contact build of the tall variant (53 KB of LDS per env: three
    envs per CU instead of two); an env whose step needs more is parked and the 160-row build finishes the launch's remaining
    steps, as for the standard variant.  Forced with `primary_rows` just under the rows of the arm-on-the-counter scenario;
    state-synchronised against the fp64 oracle: no flag, its row / contact counts, its velocities."""
    sim = _sim(3)
    assert sim.nefc_max == 128
    o = Oracle(sim._blob); o.set_option("solver", 2)
    ctrl = [0.5, -0.5, 0.9, 0.5, 1.0, -0.5, 0.3, 0.0, 0.2, -0.3]
    o.arr("ctrl")[:10] = ctrl
    sim.ctrl[:] = torch.tensor(ctrl, dtype=torch.float32, device=sim.device).unsqueeze(1)
    o.step(150)
    limit = max(30, o.nefc - 4)
    sim.set_option("primary_rows", limit)
    over, errs = 0, []
    for k in range(40):
        for e in range(3):
            sim.qpos[:, e] = torch.tensor(o.arr("qpos"), dtype=torch.float32, device=sim.device)
            sim.qvel[:, e] = torch.tensor(o.arr("qvel"), dtype=torch.float32, device=sim.device)
            sim.qacc_warmstart[:, e] = torch.tensor(o.arr("qacc_warmstart"), dtype=torch.float32, device=sim.device)
        o.step(1); sim.step(1)
        torch.cuda.synchronize()
        assert int(sim.info[3].max()) == 0, k
        assert torch.equal(sim.qpos[:, 0], sim.qpos[:, 2])
        over += o.nefc > limit
        if (int(sim.info[0, 0]), int(sim.info[1, 0])) == (o.nefc, o.ncon):
            errs.append(np.abs(sim.qvel[:, 0].cpu().numpy() - o.arr("qvel")).max() / max(1.0, np.abs(o.arr("qvel")).max()))
    assert over >= 25
    errs = np.sort(np.array(errs))
    assert len(errs) >= 30 and errs[int(0.8 * len(errs))] < 2e-3 and errs[-1] < 0.3, errs[-8:]
    sim.set_option("escalate", 0)
    sim.step(1)
    torch.cuda.synchronize()
    assert int(sim.info[3].max()) & 1   # without the hand-over the same step is flagged
    sim.stop()


@pytest.mark.parametrize("scene", ["stretch_scene", "stretch_kitchen4"])
def test_big_builds_hand_over_to_the_224_row_build(scene):
    """The two-envs-per-CU builds of the big variant (38 / 50 dof columns, 160 rows / 48 contacts) park an env whose step needs
    more and the 64-column build (224 rows / 64 contacts, one env per CU) finishes the launch's remaining steps on it -- the
    standard -> tall mechanism one size up.  Forced here with the `primary_rows` option: the primary kernel hands over beyond
    90 rows (kitchen4 settles at 96), the escalation build keeps its full capacity.  State-synchronised against the fp64 oracle
    like every contact-rich check: no flag, the oracle's row / contact counts, its velocities."""
    from stretch_mujoco_amd import StretchBatchSimulator

    sim = StretchBatchSimulator(num_envs=3, device="cuda:0", scene=scene, solver="newton")
    sim.start(home=False)
    assert sim.nefc_max == 160
    o = Oracle(sim._blob); o.set_option("solver", 2)
    ctrl = [1.5, -1.5, 0.3, 0.4, 0, -1.2, 0, 0, 0.5, -0.5]   # driving, arm out and down towards the table / counter
    o.arr("ctrl")[:10] = ctrl
    sim.ctrl[:] = torch.tensor(ctrl, dtype=torch.float32, device=sim.device).unsqueeze(1)
    o.step(150)     # past the start transient, objects settled on their supports
    limit = max(40, o.nefc - 6)
    sim.set_option("primary_rows", limit)
    over, errs = 0, []
    for k in range(40):
        for e in range(3):
            sim.qpos[:, e] = torch.tensor(o.arr("qpos"), dtype=torch.float32, device=sim.device)
            sim.qvel[:, e] = torch.tensor(o.arr("qvel"), dtype=torch.float32, device=sim.device)
            sim.qacc_warmstart[:, e] = torch.tensor(o.arr("qacc_warmstart"), dtype=torch.float32, device=sim.device)
        o.step(1); sim.step(1)
        torch.cuda.synchronize()
        assert int(sim.info[3].max()) == 0, k
        assert torch.equal(sim.qpos[:, 0], sim.qpos[:, 2])
        if o.nefc > limit:
            over += 1
        if (int(sim.info[0, 0]), int(sim.info[1, 0])) == (o.nefc, o.ncon):
            errs.append(np.abs(sim.qvel[:, 0].cpu().numpy() - o.arr("qvel")).max() / max(1.0, np.abs(o.arr("qvel")).max()))
    assert over >= 30     # the hand-over was exercised on nearly every step
    # same statistics as the state-synchronised rollout tests: the bulk of the steps at fp32 rounding, the odd step at an MPR
    # facet bifurcation (a free object resting on a face: the contact normal of two equally close facets) well above it
    errs = np.sort(np.array(errs))
    assert len(errs) >= 30 and errs[int(0.8 * len(errs))] < 2e-3 and errs[-1] < 0.3, errs[-8:]
    sim.set_option("escalate", 0)
    sim.step(1)
    torch.cuda.synchronize()
    assert int(sim.info[3].max()) & 1   # without it the same step is flagged
    sim.stop()


def test_kitchen_export_on_the_device():
    """The converted robosuite-style export (tests/test_compiler_generality.py has the CPU side) through StretchBatchSimulator on
    the device, the robot started at the removed robosuite robot's pose (`start_translation`, like the reference's
    change_start_pose): state-synchronised against the fp64 oracle for 600 steps of the reach / press / sweep script."""
    from stretch_mujoco_amd import StretchBatchSimulator
    from test_compiler_generality import KX_SCRIPT, kx_start

    sim = StretchBatchSimulator(num_envs=2, device="cuda:0", scene="stretch_kitchen_export", solver="newton",
                                start_translation=[0.0, -0.2, 0.0], start_rotation_quat=[1.0, 0.0, 0.0, 0.0])
    sim.start(home=False)
    assert sim.nv == 46 and sim.nefc_max == 160 and float(sim.qpos[1, 0]) == pytest.approx(-0.2)
    o = Oracle(sim._blob); o.set_option("solver", 2)
    kx_start(o)
    errs, same = [], 0
    for k in range(600):
        for k0, c in KX_SCRIPT:
            if k == k0:
                o.arr("ctrl")[:] = c
                sim.ctrl[:] = torch.tensor(c, dtype=torch.float32, device=sim.device).unsqueeze(1)
        for e in range(2):
            sim.qpos[:, e] = torch.tensor(o.arr("qpos"), dtype=torch.float32, device=sim.device)
            sim.qvel[:, e] = torch.tensor(o.arr("qvel"), dtype=torch.float32, device=sim.device)
            sim.qacc_warmstart[:, e] = torch.tensor(o.arr("qacc_warmstart"), dtype=torch.float32, device=sim.device)
        o.step(1); sim.step(1)
        torch.cuda.synchronize()
        assert int(sim.info[3].max()) == 0, k
        if (int(sim.info[0, 0]), int(sim.info[1, 0])) == (o.nefc, o.ncon):
            same += 1
            errs.append(np.abs(sim.qvel[:, 0].cpu().numpy() - o.arr("qvel")).max() / max(1.0, np.abs(o.arr("qvel")).max()))
    errs = np.sort(np.array(errs))
    assert same >= 570 and errs[int(0.9 * len(errs))] < 5e-4 and torch.equal(sim.qpos[:, 0], sim.qpos[:, 1]), (same, errs[-5:])
    sim.stop()
