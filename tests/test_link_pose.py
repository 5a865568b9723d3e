"""get_link_pose with the reference's semantics (stretch_mujoco_simulator.py:468-486): FK relative to base_link at the STATUS
joint positions, placed at the planar base pose.  Host math only (no GPU): the simulator object is fed the oracle's readout of
a pose with the base flat on z = 0 and the fingers at zero, where the reference's value and the simulated body pose coincide."""
import numpy as np
import torch

from oracle.oracle import Oracle
from stretch_mujoco_amd.simulator import StretchBatchSimulator


def _fake_running_sim(o, B=3):
    sim = StretchBatchSimulator(num_envs=B, device="cpu")
    sim._ctx = 1   # host-side math only; nothing below touches the library
    sim.nstep = torch.zeros(B, dtype=torch.int32)
    sim.actuator_length = torch.tensor(o.arr("actuator_length"), dtype=torch.float32).unsqueeze(1).expand(10, B).clone()
    sim.actuator_velocity = torch.zeros(10, B)
    R1 = o.arr("xmat")[1].reshape(3, 3)
    sim.base_pose = torch.tensor([o.arr("xpos")[1][0], o.arr("xpos")[1][1], np.arctan2(R1[1, 0], R1[0, 0])], dtype=torch.float32).unsqueeze(1).expand(3, B).clone()
    sim.xpose = torch.tensor(np.concatenate([np.concatenate([o.arr("xpos")[b], o.arr("xmat")[b]]) for b in range(o.dim("nbody"))]),
                             dtype=torch.float32).unsqueeze(1).expand(-1, B).clone()
    return sim


def test_reference_semantics_equal_the_simulated_pose_on_a_flat_base(blob_fused):
    o = Oracle(blob_fused)
    q = o.arr("qpos")
    th = 0.6
    q[0], q[1], q[2], q[3], q[6] = 1.0, -2.0, 0.0, np.cos(th / 2), np.sin(th / 2)
    q[9] = 0.7; q[10:14] = 0.05; q[14], q[15], q[16] = 0.3, -0.4, 0.2; q[24], q[25] = 0.5, -0.3
    o.forward()
    sim = _fake_running_sim(o)
    for link in ("link_grasp_center", "link_head_tilt", "base_link", "link_arm_l0", "link_wrist_yaw", "link_lift"):
        Tr, Ts = sim.get_link_pose(link), sim.get_link_pose(link, simulated=True)
        assert torch.allclose(Tr, Ts, atol=2e-5), link
        assert torch.allclose(Tr[:, 3], torch.tensor([0, 0, 0, 1.0]).expand(3, 4))
    sim._ctx = None


def test_reference_semantics_ignore_base_tilt_height_and_finger_joints(blob_fused):
    """What the reference's URDF evaluation cannot see: base roll / pitch / height, gripper and finger joints."""
    o = Oracle(blob_fused)
    q = o.arr("qpos")
    q[2] = 0.05                                      # base lifted 5 cm
    q[9] = 0.6; q[17] = 0.02; q[18] = q[21] = 0.2    # gripper slide and finger joints moved
    o.forward()
    sim = _fake_running_sim(o)
    ee_ref, ee_sim = sim.get_ee_pose(), sim.get_link_pose("link_grasp_center", simulated=True)
    assert abs(float(ee_sim[0, 2, 3] - ee_ref[0, 2, 3]) - 0.05) < 1e-5      # height of the base is not in the reference's value
    fl_ref, fl_sim = sim.get_link_pose("link_gripper_finger_left"), sim.get_link_pose("link_gripper_finger_left", simulated=True)
    assert float((fl_ref[0, :3, :3] - fl_sim[0, :3, :3]).abs().max()) > 0.05   # the finger's own joint is left at zero
    sim._ctx = None
