"""Host glue vs golden vectors captured from the reference's own Python (tools/gen_glue_golden.py).

The fixture holds inputs/outputs of utils.*, StatusCommand merge rules, MujocoServer.push_command +
BaseController, MujocoServer.pull_status and the client-side validation of StretchMujocoSimulator.
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from stretch_mujoco_amd import config, utils
from stretch_mujoco_amd.enums import Actuators, StretchSensors
from stretch_mujoco_amd.glue import Glue

with open(os.path.join(GOLDEN, "glue_golden.json")) as f:
    G = json.load(f)

KEY_CTRL = torch.tensor([[0, 0, 0.6, 0.1, 0, 0, 0, 0, 0, 0], [0, 0, 0.23, 0, 3.14, -0.4, 0, 0, 0, 0]], dtype=torch.float64)


def test_config_constants():
    assert list(config.robot_settings["gripper_min_max"]) == G["config"]["robot_settings"]["gripper_min_max"]
    assert list(config.robot_settings["sim_gripper_min_max"]) == G["config"]["robot_settings"]["sim_gripper_min_max"]
    assert config.robot_settings["wheel_diameter"] == G["config"]["robot_settings"]["wheel_diameter"]
    assert config.robot_settings["wheel_separation"] == G["config"]["robot_settings"]["wheel_separation"]
    assert config.depth_limits == G["config"]["depth_limits"]
    assert config.base_motion == G["config"]["base_motion"]


def test_pure_functions():
    P = G["pure"]
    for v, w, wl, wr in P["inv"]:
        assert utils.diff_drive_inv_kinematics(v, w) == pytest.approx((wl, wr), rel=1e-15, abs=1e-15)
    for a, b, V, om in P["fwd"]:
        assert utils.diff_drive_fwd_kinematics(a, b) == pytest.approx((V, om), rel=1e-15, abs=1e-15)
    sim_r, real_r = config.robot_settings["sim_gripper_min_max"], config.robot_settings["gripper_min_max"]
    for x, to_real, to_sim in P["map"]:
        assert utils.map_between_ranges(x, sim_r, real_r) == pytest.approx(to_real, rel=1e-15, abs=1e-15)
        assert utils.map_between_ranges(x, real_r, sim_r) == pytest.approx(to_sim, rel=1e-15, abs=1e-15)
        assert utils.to_real_gripper_range(x) == pytest.approx(to_real, rel=1e-15, abs=1e-15)
        assert utils.to_sim_gripper_range(x) == pytest.approx(to_sim, rel=1e-15, abs=1e-15)
    for fovy, w, h, K in P["K"]:
        np.testing.assert_allclose(utils.compute_K(fovy, w, h), np.array(K), rtol=1e-15)
    np.testing.assert_array_equal(utils.limit_depth_distance(np.array(P["depth_in"]), 1), np.array(P["depth_out_1"]))
    np.testing.assert_array_equal(utils.limit_depth_distance(np.array(P["depth_in"]), 10), np.array(P["depth_out_10"]))
    t = utils.limit_depth_distance(torch.tensor(P["depth_in"], dtype=torch.float64), 1)
    np.testing.assert_array_equal(t.numpy(), np.array(P["depth_out_1"]))
    names = StretchSensors.lidar_names(360)
    assert len(names) == 360 and [names[0], names[1], names[359]] == P["lidar_names_360"]


def _replay(ticks, dtype=torch.float64):
    g = Glue(1, 10, KEY_CTRL, ["home", "stow"], "cpu", dtype=dtype)
    ctrl = torch.zeros(10, 1, dtype=dtype)
    out = []
    for t in ticks:
        for op in t["ops"]:
            if op[0] == "move_to":
                g.move_to(op[1], op[2])
            elif op[0] == "move_by":
                g.move_by(op[1], op[2])
            elif op[0] == "base_velocity":
                g.set_base_velocity(op[1], op[2])
            elif op[0] == "keyframe":
                g.set_keyframe(op[1])
        act_len = torch.tensor(t["length"], dtype=dtype).unsqueeze(1)
        pose = torch.tensor(t["pose"], dtype=dtype).unsqueeze(1)
        g.push_command(ctrl, act_len, pose)
        out.append(dict(ctrl=ctrl[:, 0].tolist(), mode=int(g.bc_mode[0]), start=g.bc_start[:, 0].tolist()))
    return out


@pytest.mark.parametrize("idx", range(len(G["push_command"])))
def test_push_command_sequences(idx):
    sc = G["push_command"][idx]
    got = _replay(sc["ticks"])
    for k, (g, e) in enumerate(zip(got, sc["expect"])):
        np.testing.assert_allclose(g["ctrl"], e["ctrl"], rtol=1e-12, atol=1e-12, err_msg=f"scenario {idx} tick {k}")
        assert g["mode"] == e["mode"], f"scenario {idx} tick {k}: controller mode"
        if e["mode"] != 0:
            np.testing.assert_allclose(g["start"], e["start"], rtol=1e-12, atol=1e-12)


def test_push_command_batched_equals_single():
    """All scenarios of equal length replayed as one batch give the same per-env results."""
    by_len = {}
    for sc in G["push_command"]:
        by_len.setdefault(len(sc["ticks"]), []).append(sc)
    for n, group in by_len.items():
        B = len(group)
        g = Glue(B, 10, KEY_CTRL.float(), ["home", "stow"], "cpu")
        ctrl = torch.zeros(10, B)
        for k in range(n):
            for e, sc in enumerate(group):
                for op in sc["ticks"][k]["ops"]:
                    if op[0] == "move_to":
                        g.move_to(op[1], op[2], env_ids=[e])
                    elif op[0] == "move_by":
                        g.move_by(op[1], op[2], env_ids=[e])
                    elif op[0] == "base_velocity":
                        g.set_base_velocity(op[1], op[2], env_ids=[e])
                    else:
                        g.set_keyframe(op[1], env_ids=[e])
            act_len = torch.tensor([sc["ticks"][k]["length"] for sc in group]).t().float()
            pose = torch.tensor([sc["ticks"][k]["pose"] for sc in group]).t().float()
            g.push_command(ctrl, act_len, pose)
            exp = np.array([sc["expect"][k]["ctrl"] for sc in group]).T
            np.testing.assert_allclose(ctrl.numpy(), exp, rtol=2e-6, atol=2e-6)


def test_pull_status():
    for case in G["pull_status"]:
        L = torch.tensor(case["length"], dtype=torch.float64).unsqueeze(1)
        V = torch.tensor(case["velocity"], dtype=torch.float64).unsqueeze(1)
        P = torch.tensor(case["pose"], dtype=torch.float64).unsqueeze(1)
        st = Glue.pull_status(torch.tensor([case["time"]], dtype=torch.float64), L, V, P)
        e = case["expect"]
        for k in ("lift", "arm", "head_pan", "head_tilt", "wrist_yaw", "wrist_pitch", "wrist_roll", "gripper"):
            assert float(st[k].pos[0]) == pytest.approx(e[k][0], rel=1e-12, abs=1e-12), k
            assert float(st[k].vel[0]) == pytest.approx(e[k][1], rel=1e-12, abs=1e-12), k
        got = [float(st.base.x[0]), float(st.base.y[0]), float(st.base.theta[0]), float(st.base.x_vel[0]), float(st.base.theta_vel[0])]
        assert got == pytest.approx(e["base"], rel=1e-12, abs=1e-12)
        assert float(st.time[0]) == e["time"]


def test_client_validation_matches_reference():
    g = Glue(2, 10, KEY_CTRL.float(), ["home", "stow"], "cpu")
    for case in G["validation"]["cases"]:
        fn = getattr(g, case["method"])
        if case["ok"] and case["actuator"].startswith("gripper_") and case["actuator"].endswith("_finger"):
            # accepted by the reference's client, then KeyError in its server (no MJCF actuator of that name);
            # in-process here, so the KeyError surfaces at the call
            with pytest.raises(KeyError):
                fn(case["actuator"], 0.1)
        elif case["ok"]:
            fn(case["actuator"], 0.1)
        else:
            exc = {"Exception": Exception, "KeyError": KeyError}[case["exc"]]
            with pytest.raises(exc) as ei:
                fn(case["actuator"], 0.1)
            if case["exc"] == "Exception":
                assert type(ei.value) is Exception and str(ei.value) == case["msg"]


def test_finger_pseudo_actuators_have_no_mjcf_actuator():
    """The reference accepts these names client-side and fails server-side (no such MJCF actuator); here the
    failure surfaces at the call as a KeyError."""
    g = Glue(1, 10, KEY_CTRL.float(), ["home", "stow"], "cpu")
    with pytest.raises(KeyError):
        g.move_to(Actuators.gripper_left_finger, 0.1)
