"""The MJCF export of the compiled model (mjcf_export.py, input of the live-MuJoCo harness tools/mujoco_harness.py): every
number MuJoCo needs is in the XML and round-trips to the blob; the harness itself runs when `import mujoco` works and says
UNVERIFIED, loudly, when it does not (SURVEY.md 8(c) run-time plan)."""
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np
import pytest

from conftest import MODELS, ROOT
from stretch_mujoco_amd import model_blob
from stretch_mujoco_amd.mjcf_export import export_mjcf

sys.path.insert(0, os.path.join(ROOT, "tools"))
import mujoco_harness  # noqa: E402


def _fl(s):
    return np.array([float(x) for x in s.split()])


@pytest.mark.parametrize("scene", ["stretch_empty", "stretch_empty_full", "stretch_scene"])
def test_export_round_trips_to_the_blob(scene):
    with open(os.path.join(MODELS, scene + ".smjb"), "rb") as f:
        m = model_blob.loads(f.read())
    root = ET.fromstring(export_mjcf(m))
    nb, ngeom = int(m["dims"][3]), int(m["dims"][5])
    bodies = root.findall(".//body")
    assert len(bodies) == nb - 1
    # body tree order = depth-first = the blob's ids; frames, inertials, gravcomp
    for i, b in enumerate(bodies, start=1):
        assert np.array_equal(_fl(b.get("pos")), m["body_pos"][i]) and np.array_equal(_fl(b.get("quat")), m["body_quat"][i])
        assert float(b.get("gravcomp")) == float(m["body_gravcomp"][i])
        ine = b.find("inertial")
        if float(m["body_mass"][i]) > 0:
            assert float(ine.get("mass")) == float(m["body_mass"][i]) and np.array_equal(_fl(ine.get("diaginertia")), m["body_inertia"][i])
            assert np.array_equal(_fl(ine.get("pos")), m["body_ipos"][i]) and np.array_equal(_fl(ine.get("quat")), m["body_iquat"][i])
    # joints in blob order, with every dof parameter
    joints = [j for b in bodies for j in b if j.tag in ("joint", "freejoint")]
    assert len(joints) == int(m["dims"][4])
    for j, e in enumerate(joints):
        if int(m["jnt_type"][j]) == 0:
            assert e.tag == "freejoint"
            continue
        d = int(m["jnt_dofadr"][j])
        assert np.array_equal(_fl(e.get("axis")), m["jnt_axis"][j]) and np.array_equal(_fl(e.get("range")), m["jnt_range"][j])
        assert float(e.get("armature")) == float(m["dof_armature"][d]) and float(e.get("damping")) == float(m["dof_damping"][d])
        assert float(e.get("frictionloss")) == float(m["dof_frictionloss"][d]) and (e.get("limited") == "true") == bool(m["jnt_limited"][j])
    # collision geoms are all there, switched to explicit pairs; hull vertices intact
    geoms = {g.get("name"): g for g in root.findall(".//geom")}
    collide = set(int(g) for g in m["pair_geom1"]) | set(int(g) for g in m["pair_geom2"])
    assert all(f"g{g}" in geoms for g in collide)
    assert all(g.get("contype") == "0" and g.get("conaffinity") == "0" for g in geoms.values())
    hv = np.asarray(m["hull_vert"]).reshape(-1, 3)
    for me in root.findall(".//asset/mesh"):
        g = int(me.get("name")[4:])
        a, n = int(m["geom_hulladr"][g]), int(m["geom_hullnum"][g])
        assert np.array_equal(_fl(me.get("vertex")).reshape(-1, 3), hv[a:a + n])
    pairs = root.findall(".//contact/pair")
    assert len(pairs) == len(m["pair_geom1"])
    for p in (0, len(pairs) // 2, len(pairs) - 1):
        assert pairs[p].get("geom1") == f"g{int(m['pair_geom1'][p])}" and int(pairs[p].get("condim")) == int(m["pair_condim"][p])
        assert np.array_equal(_fl(pairs[p].get("friction")), m["pair_friction"][p]) and np.array_equal(_fl(pairs[p].get("solimp")), m["pair_solimp"][p])
    # actuators, tendon, equalities, sensors, keyframes
    acts = root.findall(".//actuator/general")
    assert [a.get("name") for a in acts] == ["left_wheel_vel", "right_wheel_vel", "lift", "arm", "wrist_yaw", "wrist_pitch", "wrist_roll", "gripper", "head_pan", "head_tilt"]
    for a, e in enumerate(acts):
        assert np.array_equal(_fl(e.get("gainprm")), m["actuator_gainprm"][a]) and np.array_equal(_fl(e.get("biasprm")), m["actuator_biasprm"][a])
        assert np.array_equal(_fl(e.get("ctrlrange")), m["actuator_ctrlrange"][a]) and float(e.get("gear")) == float(m["actuator_gear"][a])
    assert acts[3].get("tendon") == "extend" and acts[2].get("joint") == "joint_lift"
    assert len(root.findall(".//tendon/fixed/joint")) == 4 and len(root.findall(".//equality/joint")) == 5
    assert len(root.findall(".//sensor/rangefinder")) == 360 and root.find(".//sensor/gyro") is not None
    opt = root.find("option")
    assert opt.get("integrator") == "implicitfast" and opt.get("cone") == "elliptic" and opt.find("flag").get("multiccd") == "enable"
    assert [k.get("name") for k in root.findall(".//keyframe/key")] == ["home", "stow"]


def test_harness_reports_mujoco_parity_or_says_unverified():
    """With MuJoCo importable this IS the parity test against the reference's physics (Appendix D.10: qpos drift of the fp64
    restatement < 1e-4 over 1000 steps on the home / mixed scripts, identical contact and row counts on > 99 % of the steps);
    without it the result is a skip that says so -- never a silent pass."""
    if not mujoco_harness.have_mujoco():
        print("\n" + mujoco_harness.UNVERIFIED)
        pytest.skip(mujoco_harness.UNVERIFIED)
    res = mujoco_harness.compare("stretch_empty", steps=1000)
    print(res["verdict"])
    assert all(a == b for a, b in res["model"]["dims"].values())
    for k in ("qpos0", "body_mass", "body_subtreemass", "dof_invweight0", "stat_meaninertia"):
        assert res["model"][k] < 1e-9, (k, res["model"][k])
    for name in ("home", "mixed"):
        t = res["trajectories"][name]
        assert t["max_qpos_drift"] < 1e-4 and t["steps_nefc_differs"] < 10, (name, t)
