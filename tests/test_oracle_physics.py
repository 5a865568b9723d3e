"""The fp64 oracle against closed-form answers and the weak sanity bands the reference's docs print.

MuJoCo itself is absent from this image (parity unpinned); these tests pin the oracle to
physics that has an analytic answer, to the discrete-time recurrences MuJoCo's integrator implies, and to the
printed post-home() status in the reference's README / notebook.
"""
import numpy as np
import pytest

import os

from conftest import HOME_CTRL, MODELS
from oracle.oracle import Oracle
from stretch_mujoco_amd import mjcf_compiler as C
from stretch_mujoco_amd import model_blob as B

OPT = '<option integrator="implicitfast" cone="elliptic" impratio="20"/>'


def make(xml):
    return Oracle(B.dumps(C.compile_string(xml)))


def test_free_fall_recurrence():
    o = make(f'<mujoco><compiler angle="radian"/>{OPT}<worldbody><body pos="0 0 5"><freejoint/>'
             '<geom type="box" size=".1 .2 .3"/></body></worldbody></mujoco>')
    h, g, n = 0.002, 9.81, 500
    o.step(n)
    assert o.arr("qvel")[2] == pytest.approx(-g * h * n, rel=1e-12)
    assert o.arr("qpos")[2] == pytest.approx(5 - g * h * h * n * (n + 1) / 2, rel=1e-12)
    assert abs(o.arr("qpos")[3] - 1) < 1e-14 and o.nefc == 0


def test_spinning_free_body_conserves_angular_momentum_direction():
    o = make(f'<mujoco><compiler angle="radian"/>{OPT}<option gravity="0 0 0"/><worldbody><body><freejoint/>'
             '<geom type="box" size=".1 .2 .3" mass="2"/></body></worldbody></mujoco>')
    o.arr("qvel")[3:6] = [0.3, 2.0, 0.1]  # near the (unstable) intermediate axis: exercises the gyroscopic bias term
    I = np.array([2 / 3 * (.04 + .09), 2 / 3 * (.01 + .09), 2 / 3 * (.01 + .04)])
    o.forward()
    R0 = o.arr("xmat")[1].reshape(3, 3).copy()
    L0 = R0 @ (I * o.arr("qvel")[3:6])
    E0 = 0.5 * np.sum(I * o.arr("qvel")[3:6] ** 2)
    o.step(2000)
    o.forward()
    R = o.arr("xmat")[1].reshape(3, 3)
    L = R @ (I * o.arr("qvel")[3:6])
    assert np.linalg.norm(L - L0) / np.linalg.norm(L0) < 2e-2   # first-order integrator, 4 s
    assert abs(0.5 * np.sum(I * o.arr("qvel")[3:6] ** 2) - E0) / E0 < 2e-2


def test_pendulum_small_angle_period():
    # point-like mass on a massless arm: T = 2 pi sqrt(L/g)
    o = make(f'<mujoco><compiler angle="radian"/>{OPT}<worldbody><body pos="0 0 2"><joint type="hinge" axis="0 1 0"/>'
             '<geom type="sphere" size="0.01" pos="0 0 -1" mass="1"/></body></worldbody></mujoco>')
    o.arr("qpos")[0] = 0.02
    zero = []
    prev = o.arr("qpos")[0]
    for k in range(3000):
        o.step(1)
        cur = o.arr("qpos")[0]
        if prev > 0 >= cur:
            zero.append(k * 0.002 + 0.002 * prev / (prev - cur))
        prev = cur
    T = np.mean(np.diff(zero))
    assert T == pytest.approx(2 * np.pi * np.sqrt(1.0 / 9.81), rel=2e-3)


def test_position_servo_steady_state_with_gravity():
    # slide joint along z, position actuator kp: q_ss = ctrl - m g / kp
    o = make(f'<mujoco><compiler angle="radian"/>{OPT}<worldbody><body><joint name="j" type="slide" axis="0 0 1" damping="20"/>'
             '<geom type="sphere" size="0.05" mass="2"/></body></worldbody>'
             '<actuator><position joint="j" kp="400"/></actuator></mujoco>')
    o.arr("ctrl")[0] = 0.3
    o.step(5000)
    assert o.arr("qpos")[0] == pytest.approx(0.3 - 2 * 9.81 / 400, abs=1e-9)
    assert abs(o.arr("qvel")[0]) < 1e-9


def test_sphere_rests_on_plane_with_weight_as_normal_force():
    o = make(f'<mujoco><compiler angle="radian"/>{OPT}<worldbody><geom type="plane" size="0 0 1" condim="1"/><body pos="0 0 0.1"><freejoint/>'
             '<geom type="sphere" size="0.1" mass="3" condim="1"/></body></worldbody></mujoco>')
    o.step(3000)
    o.forward()
    assert o.ncon == 1 and o.nefc == 1
    assert o.arr("efc_force")[0] == pytest.approx(3 * 9.81, rel=1e-6)
    assert -2e-3 < o.arr("qpos")[2] - 0.1 < 0 and abs(o.arr("qvel")[2]) < 1e-8


def test_box_sticks_below_and_slides_above_the_coulomb_limit():
    xml = (f'<mujoco><compiler angle="radian"/>{OPT}<worldbody><geom type="plane" size="0 0 1" friction="0.5 0.005 0.0001"/>'
           '<body pos="0 0 0.05"><freejoint/><geom type="box" size="0.1 0.1 0.05" mass="1" friction="0.5 0.005 0.0001"/></body>'
           '</worldbody></mujoco>')
    mu_mg = 0.5 * 9.81
    o = make(xml)
    o.step(300)
    o.arr("qfrc_applied")[0] = 3.0  # below mu*m*g: sticks (the soft friction model only creeps)
    o.step(500)
    assert abs(o.arr("qvel")[0]) < 2e-3
    o = make(xml)
    o.step(300)
    o.arr("qfrc_applied")[0] = 8.0  # above: slides with a = (F - mu m g)/m
    o.step(100)
    v0 = o.arr("qvel")[0]
    o.step(200)
    acc = (o.arr("qvel")[0] - v0) / 0.4
    assert acc == pytest.approx(8.0 - mu_mg, rel=0.02)


def test_joint_limit_stops_motion():
    o = make(f'<mujoco><compiler angle="radian"/>{OPT}<worldbody><body><joint name="j" type="slide" axis="1 0 0" range="-0.1 0.2"/>'
             '<geom type="sphere" size="0.05" mass="1"/></body></worldbody>'
             '<actuator><motor joint="j"/></actuator></mujoco>')
    o.arr("ctrl")[0] = 5.0
    o.step(2000)
    assert 0.2 < o.arr("qpos")[0] < 0.205 and abs(o.arr("qvel")[0]) < 1e-6
    o.forward()
    assert o.arr("efc_force")[0] == pytest.approx(5.0, rel=1e-5)


def test_stretch_home_settles_inside_the_documented_bands(blob_full):
    """README.md:136-145 / docs/getting_started.ipynb:729-750 print the settled status after home()."""
    o = Oracle(blob_full)
    o.arr("ctrl")[:] = HOME_CTRL
    o.step(4000)
    o.forward()
    L = o.arr("actuator_length")
    assert 0.5885 <= L[2] <= 0.5912            # lift: 0.58897 (README) / 0.59055 (notebook) for ctrl 0.6
    assert 0.0975 <= L[3] <= 0.1001            # arm: 0.09806 / 0.10000 for ctrl 0.1
    assert L[9] == pytest.approx(-0.004519, abs=2e-5)   # head_tilt sag: -0.004519 in both
    assert abs(L[8]) < 5e-5                    # head_pan -4.97e-06
    assert -0.006 <= L[5] <= -0.003            # wrist_pitch -0.003345 / -0.005325
    assert abs(o.arr("qvel")).max() < 1e-3
    assert o.nefc == 42 and o.ncon == 5        # 5 eq + 12 friction + 2 wheels x 2 x 6 + caster 1


def test_stretch_head_tilt_limit(blob_full):
    """notebook cell 23: head_tilt commanded -2.0 "did not reach -2.0. Actual: -1.522573472981672" (limit -1.53): ctrl is clamped
    to ctrlrange, the servo then holds at the range edge minus the gravity sag.  The oracle's steady state is the printed number
    to nine digits -- joint limit row, position servo, gravity compensation and the solver's force balance as MuJoCo has them."""
    o = Oracle(blob_full)
    c = np.array(HOME_CTRL, float)
    c[9] = -2.0
    o.arr("ctrl")[:] = c
    o.step(4000)
    o.forward()
    assert abs(o.arr("actuator_velocity")[9]) < 1e-8
    assert o.arr("actuator_length")[9] == pytest.approx(-1.522573472981672, abs=2e-9)


def test_mpr_penetration_against_closed_forms():
    """Convex narrowphase (MPR, restating libccd's ccdMPRPenetration): depth / normal / position of primitive pairs
    with a closed-form answer.  geom1 is the lower MuJoCo type id (sphere 2 < cylinder 5 < box 6)."""
    opt = '<option integrator="implicitfast" cone="elliptic" impratio="20" gravity="0 0 0"/>'

    def first_contact(xml):
        o = make(xml)
        o.set_option("multiccd", 0)    # the single-point query itself; the multi-point manifolds are tested below
        o.forward()
        assert o.ncon == 1
        c = o.arr("contact").reshape(1, -1)[0]
        return c[0], c[1:4], c[4:7]

    d, p, n = first_contact(f'<mujoco><compiler angle="radian"/>{opt}<worldbody><body><freejoint/><geom type="sphere" size="0.1"/></body>'
                            '<body pos="0.15 0 0"><freejoint/><geom type="sphere" size="0.1"/></body></worldbody></mujoco>')
    assert d == pytest.approx(-0.05, abs=1e-9) and np.allclose(n, [1, 0, 0], atol=1e-9) and np.allclose(p, [0.075, 0, 0], atol=1e-9)
    d, p, n = first_contact(f'<mujoco><compiler angle="radian"/>{opt}<worldbody><body><freejoint/><geom type="box" size="0.2 0.2 0.1"/></body>'
                            '<body pos="0.05 0.03 0.18"><freejoint/><geom type="box" size="0.1 0.1 0.1"/></body></worldbody></mujoco>')
    assert d == pytest.approx(-0.02, abs=1e-6) and np.allclose(n, [0, 0, 1], atol=1e-6) and p[2] == pytest.approx(0.09, abs=1e-6)
    d, p, n = first_contact(f'<mujoco><compiler angle="radian"/>{opt}<worldbody><body><freejoint/><geom type="box" size="0.2 0.2 0.1"/></body>'
                            '<body pos="0 0 0.19" euler="1.5708 0 0"><freejoint/><geom type="cylinder" size="0.1 0.3"/></body></worldbody></mujoco>')
    assert d == pytest.approx(-0.01, abs=1e-5) and np.allclose(n, [0, 0, -1], atol=1e-4)    # cylinder is geom1: normal towards the box
    o = make(f'<mujoco><compiler angle="radian"/>{opt}<worldbody><body><freejoint/><geom type="box" size="0.2 0.2 0.1"/></body>'
             '<body pos="0 0 0.25"><freejoint/><geom type="box" size="0.1 0.1 0.1"/></body></worldbody></mujoco>')
    o.forward()
    assert o.ncon == 0


def test_box_box_polygon_and_multiccd_manifolds():
    """Multi-point contacts (stretch.xml:8 enables multiccd): a box on a box gets the clipped face polygon -- here the four
    bottom corners of the small box, all 2 cm deep, midway between the two faces; partially overlapping faces give the
    intersection polygon; an edge-edge crossing gives one point.  A cylinder lying on a box gets a line of points from the
    counter-rotated queries, all sharing the first normal.  A box resting on a box does not rock."""
    opt = '<option integrator="implicitfast" cone="elliptic" impratio="20" gravity="0 0 0"/>'
    o = make(f'<mujoco><compiler angle="radian"/>{opt}<worldbody><body><freejoint/><geom type="box" size="0.2 0.2 0.1"/></body>'
             '<body pos="0.05 0.03 0.18"><freejoint/><geom type="box" size="0.1 0.1 0.1"/></body></worldbody></mujoco>')
    o.forward()
    c = o.arr("contact").reshape(o.ncon, -1)
    assert o.ncon == 4 and np.allclose(c[:, 0], -0.02, atol=1e-12) and np.allclose(c[:, 4:7], [0, 0, 1], atol=1e-12)
    assert np.allclose(c[:, 3], 0.09, atol=1e-12)
    corners = sorted((round(x, 6), round(y, 6)) for x, y in c[:, 1:3])
    assert corners == [(-0.05, -0.07), (-0.05, 0.13), (0.15, -0.07), (0.15, 0.13)]
    # faces overlapping partially: the small box hangs over the edge x = 0.2 of the big one -> polygon [0.15, 0.2] x [-0.1, 0.1]
    o = make(f'<mujoco><compiler angle="radian"/>{opt}<worldbody><body><freejoint/><geom type="box" size="0.2 0.2 0.1"/></body>'
             '<body pos="0.25 0 0.19"><freejoint/><geom type="box" size="0.1 0.1 0.1"/></body></worldbody></mujoco>')
    o.forward()
    c = o.arr("contact").reshape(o.ncon, -1)
    pts = sorted((round(x, 6), round(y, 6)) for x, y in c[:, 1:3])
    assert o.ncon == 4 and pts == [(0.15, -0.1), (0.15, 0.1), (0.2, -0.1), (0.2, 0.1)] and np.allclose(c[:, 0], -0.01, atol=1e-12)
    # edge against edge: the upper box turned 45 deg about x and about z, its lowest edge crossing the top edge region
    o = make(f'<mujoco><compiler angle="radian"/>{opt}<worldbody><body><freejoint/><geom type="box" size="0.2 0.2 0.1"/></body>'
             '<body pos="0.2 0 0.235" euler="0.7853981634 0 0.7853981634"><freejoint/><geom type="box" size="0.1 0.1 0.1"/></body></worldbody></mujoco>')
    o.forward()
    assert 1 <= o.ncon <= 2
    # cylinder lying on the box (axis along y): the first contact plus the counter-rotated queries spread along the line
    o = make(f'<mujoco><compiler angle="radian"/>{opt}<worldbody><body><freejoint/><geom type="box" size="0.2 0.2 0.1"/></body>'
             '<body pos="0 0 0.19" euler="1.5708 0 0"><freejoint/><geom type="cylinder" size="0.1 0.15"/></body></worldbody></mujoco>')
    o.forward()
    c = o.arr("contact").reshape(o.ncon, -1)
    assert 2 <= o.ncon <= 5 and np.allclose(c[:, 4:7], c[0, 4:7], atol=1e-12) and np.ptp(c[:, 2]) > 0.2   # spread along y
    assert np.all(np.abs(c[:, 0] + 0.01) < 1e-3)    # depths are those of the counter-rotated (1e-3 rad) configurations
    # a box put on a box stays put (a single contact point would let it rock)
    o = make(f'<mujoco><compiler angle="radian"/>{OPT}<worldbody><geom type="plane" size="0 0 1"/>'
             '<body pos="0 0 0.1"><freejoint/><geom type="box" size="0.2 0.2 0.1" mass="2"/></body>'
             '<body pos="0.05 0.03 0.2995"><freejoint/><geom type="box" size="0.05 0.04 0.1" mass="0.5"/></body></worldbody></mujoco>')
    o.set_option("solver", 2)
    o.step(1000)
    q = o.arr("qpos")
    assert np.abs(o.arr("qvel")).max() < 1e-6 and abs(q[9] - 0.2995) < 2e-3 and np.allclose(q[10:14], [1, 0, 0, 0], atol=1e-3)
    assert abs(q[7] - 0.05) < 1e-3 and abs(q[8] - 0.03) < 1e-3


def test_sphere_rests_on_a_free_box():
    """Convex pair in the loop: a sphere (MPR contact with the box, one contact per pair) on a box on the plane."""
    o = make(f'<mujoco><compiler angle="radian"/>{OPT}<worldbody><geom type="plane" size="0 0 1"/>'
             '<body pos="0 0 0.1"><freejoint/><geom type="box" size="0.2 0.2 0.1" mass="2"/></body>'
             '<body pos="0.02 0.01 0.3"><freejoint/><geom type="sphere" size="0.1" mass="1"/></body></worldbody></mujoco>')
    o.set_option("solver", 2)
    o.step(1500)
    o.forward()
    q = o.arr("qpos")
    assert abs(q[2] - 0.1) < 3e-3 and abs(q[9] - 0.3) < 6e-3 and np.abs(o.arr("qvel")).max() < 0.2   # the ball may roll slowly
    c = o.arr("contact").reshape(o.ncon, -1)
    assert o.ncon == 5 and -3e-3 < c[4][0] < 0 and np.allclose(c[4][4:7], [0, 0, -1], atol=1e-3)   # sphere (geom1) -> box: down


def test_status_at_t_8_26_against_the_notebook(blob_full):
    """docs/getting_started.ipynb cell 20 prints pull_status() at time = 8.26 s of a robot that was started (qpos0, home
    keyframe targets) and left alone.  Joint by joint, the oracle at step 4130 against the printed values (the reference's own
    MuJoCo, fp64): lift and arm to 2e-5, the wrist and head joints to 1e-6 -- tighter than any band, and at a stated time.
    wrist_yaw is still creeping there (printed velocity -3.3e-05 rad/s): 1e-4.  The base pose is not compared: x, y, theta of
    the settled base are what is left of the first contact transient of the drop from qpos0 (1 cm, 4 degrees in the notebook)."""
    o = Oracle(blob_full)
    o.arr("ctrl")[:] = HOME_CTRL
    o.step(4130)
    o.forward()
    assert o.time == pytest.approx(8.26, abs=1e-9)
    L = o.arr("actuator_length")
    nb = {2: (0.5905520090306994, 2e-5), 3: (0.09999622635034094, 2e-5), 8: (-5.005046374741913e-06, 1e-6), 9: (-0.004519272499335126, 1e-6),
          4: (9.232975816659571e-05, 1e-4), 5: (-0.005324523093874352, 1e-6), 6: (-9.586627571896982e-05, 1e-6)}
    for a, (val, tol) in nb.items():
        assert abs(L[a] - val) < tol, (a, L[a], val)
    # the lift is still creeping towards its target at that time: printed velocity 2.2064e-4 m/s, here 2.193e-4
    assert o.arr("actuator_velocity")[2] == pytest.approx(0.00022063552289719744, rel=0.02)
    # gripper: printed -0.06399746756801022 = the reference's sim -> real range map of the simulated 2.5e-6
    from stretch_mujoco_amd.utils import to_real_gripper_range
    assert float(to_real_gripper_range(np.array([L[7]]))[0]) == pytest.approx(-0.06399746756801022, abs=2e-5)


# docs/getting_started.ipynb cell 20: the base pose the reference's own MuJoCo printed at t = 8.26 s
NOTEBOOK_BASE = (-0.012182561444183192, 0.004419350400411598, -0.06498666843943465)


def _start_transient(blob, k, eps, steps=600):
    """The reference's start sequence (stretch_mujoco_simulator.py:126-136, mujoco_server.py:457-463): the server steps with
    ctrl = 0 until the client has seen a status and a camera frame, only then is the home keyframe sent -- k physics steps after
    the reset.  eps perturbs the lift's start position (round-off stand-in).  Returns base (x, y, theta) once it has settled."""
    o = Oracle(blob)
    o.arr("qpos")[9] += eps
    if k:
        o.step(k)
    o.arr("ctrl")[:] = HOME_CTRL
    o.step(steps - k)
    o.forward()
    p, R = o.arr("xpos")[1], o.arr("xmat")[1].reshape(3, 3)
    return p[0], p[1], float(np.arctan2(R[1, 0], R[0, 0]))


def test_base_pose_at_t_8_26_lies_inside_the_start_transient_ensemble():
    """The one reference-held number that depends on contact dynamics: the notebook's base pose at t = 8.26 s,
    (x, y, theta) = (-0.0122, 0.0044, -0.0650).  It is what is left of the START transient: at qpos0 the lift is down and the
    wrist sits 6.5 cm inside the base hull; while the lift drives out (the first ~30 steps) the deep, degenerate penetration
    kicks the base, which comes to rest within 250 steps and does not move again (asserted below).  That transient is CHAOTIC:
    a 1e-12 m change of the start state, or sending `home` one physics step later, moves the final theta by 1e-2 rad; in one
    branch the base is tipped 5 mm off a wheel and lands 3-9 degrees away.  So the printed pose is not a function of the algorithm
    but of round-off (two MuJoCo builds would not agree on it either); what CAN be asserted is that it is a member of the
    oracle's outcome distribution, and that the eight joint values of the same printout (test above) hold in every branch.
    Ensemble: k = 0..4 steps of ctrl = 0 before `home` (the lift's creep at t = 8.26 s pins k below ~13 steps) x 4 perturbations."""
    blob = open(os.path.join(MODELS, "stretch_scene.smjb"), "rb").read()
    runs = np.array([_start_transient(blob, k, eps) for k in range(5) for eps in (0.0, 1e-10, 1e-6, 1e-4)])
    lo, hi = runs.min(0), runs.max(0)
    for name, v, a, b in zip("x y theta".split(), NOTEBOOK_BASE, lo, hi):
        assert a <= v <= b, f"printed base {name} = {v} outside the ensemble [{a}, {b}]"
    # the spread itself is the finding: theta varies by more than the printed value's magnitude
    assert hi[2] - lo[2] > 0.06 and hi[0] - lo[0] > 0.015
    # at least one member lands near the printed pose in all three coordinates at once (the tipped branch)
    near = (np.abs(runs[:, 2] - NOTEBOOK_BASE[2]) < 0.02) & (np.abs(runs[:, 0] - NOTEBOOK_BASE[0]) < 0.012) & (np.abs(runs[:, 1] - NOTEBOOK_BASE[1]) < 0.003)
    assert near.any(), runs
    # settled: nothing moves the base between step 600 and t = 8.26 s
    a = _start_transient(blob, 2, 0.0, steps=600)
    b = _start_transient(blob, 2, 0.0, steps=1500)
    assert np.allclose(a, b, atol=2e-5)


@pytest.mark.parametrize("k", [2, 4])
def test_status_at_t_8_26_holds_when_home_is_sent_k_steps_after_the_reset(k):
    """The eight printed joint values with the reference's start sequence (ctrl = 0 for k steps, then `home`), in the default
    scene: they do not depend on the branch the base took (k = 2: the tipped branch, theta = -0.055)."""
    blob = open(os.path.join(MODELS, "stretch_scene.smjb"), "rb").read()
    o = Oracle(blob)
    o.step(k)
    o.arr("ctrl")[:] = HOME_CTRL
    o.step(4130 - k)
    o.forward()
    L = o.arr("actuator_length")
    nb = {2: (0.5905520090306994, 2e-5), 3: (0.09999622635034094, 2e-5), 8: (-5.005046374741913e-06, 1e-6), 9: (-0.004519272499335126, 1e-6),
          4: (9.232975816659571e-05, 1e-4), 5: (-0.005324523093874352, 1e-6), 6: (-9.586627571896982e-05, 1e-6)}
    for a, (val, tol) in nb.items():
        assert abs(L[a] - val) < tol, (a, L[a], val)


def test_status_at_t_6_422_against_the_readme(blob_full):
    """README.md:136-145 prints a second pull_status(): 'time': 6.421999999999515, a robot that was started and left alone (the
    commands above the printout in the README were evidently not what produced it: head_pan -5e-6 after `move_by('head_pan',
    -1.1)`, base velocity 3e-7 after `set_base_velocity(0.3, -0.1)`).  What it pins:
      * the clock: 6.421999999999515 is the fp64 sum of 3211 timesteps of 0.002 -- digit for digit what the oracle's `time` holds
        after 3211 steps (MuJoCo adds the timestep every step; nstep * dt would print 6.422);
      * head_tilt -0.00451929555883404 and head_pan -4.968686850480367e-06 to 1e-7: the same quasi-static balance (position
        servo + gravity sag + equality / friction rows) as the notebook's t = 8.26 s printout, from another MuJoCo run.
    The lift / arm / wrist values of this printout (0.58897, 0.09806, ...) contradict the notebook's own MuJoCo output at a LATER
    time (lift 0.59055 still creeping upwards at 2.2e-4 m/s): the README predates the current actuator gains of stretch.xml, so
    only the documented band is asserted for them (test_stretch_home_settles_inside_the_documented_bands)."""
    o = Oracle(blob_full)
    o.arr("ctrl")[:] = HOME_CTRL
    o.step(3211)
    o.forward()
    assert repr(float(o.time)) == "6.421999999999515"
    L = o.arr("actuator_length")
    assert abs(L[9] - (-0.00451929555883404)) < 1e-7, L[9]
    assert abs(L[8] - (-4.968686850480367e-06)) < 1e-7, L[8]
    assert 0.5885 <= L[2] <= 0.5912 and 0.0975 <= L[3] <= 0.1005


def _lidar_figure():
    import json
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "lidar_figure.json")))


def lidar_figure_check(scan, fig, absent_rays=()):
    """Compare a 360-ray scan (metres, -1 = no hit, clipped to the cutoff) with the notebook's figure (tests/golden/
    lidar_figure.json, tools/gen_lidar_golden.py): every ray, drawn the way cell 18 draws it (x = -r cos i, y = -r sin i,
    negative -> 0), must land on a red pixel of the figure -- within 3 px: 0.13 m at the figure's scale, 0.8 degree at r = 10 m.
    Returns (rays off the drawing, fraction of the figure's red pixels within 3.5 px of a drawn ray)."""
    x0, x1, y0, y1 = fig["axes_px"]
    lim = fig["lim"]
    red = np.array([(r, c) for r, c0, n in fig["red_runs"] for c in range(c0, c0 + n)], float)
    r = np.where(scan < 0, 0.0, scan)
    a = np.radians(np.arange(360))
    px = x0 + (-r * np.cos(a) + lim) / (2 * lim) * (x1 - x0)
    py = y1 - (-r * np.sin(a) + lim) / (2 * lim) * (y1 - y0)
    d = np.sqrt((red[:, 1][None, :] - px[:, None]) ** 2 + (red[:, 0][None, :] - py[:, None]) ** 2)
    off = [i for i in range(360) if d[i].min() > 3.0 and i not in absent_rays]
    return off, float((d.min(0) <= 3.5).mean())


def test_lidar_scan_reproduces_the_notebook_figure():
    """docs/getting_started.ipynb cell 18 keeps the reference's lidar scan as a figure (MuJoCo's own rangefinder output, default
    scene, settled robot).  The oracle's scan of the compiled default scene, drawn the same way, falls on it ray by ray:
      * rays 144..270 return EXACTLY the cutoff (10 m): the base rests pitched forward by a fraction of a degree, these rays
        meet the infinite floor plane beyond the cutoff and the sensor clips to it ([MJ] rangefinder + cutoff) -- the figure's
        arc spans rays 144..271: the same half plane, its edge within one degree (sign and direction of the settled tilt, the
        180-degree turn of the laser body, the sense of the replicate's rotation);
      * rays pointing upwards return -1 (drawn at the origin after `scan_data[scan_data < 0] = 0`), nothing returns between 1.5 and
        9.6 m in either;
      * rays 42..138 meet the table's front edge (y = -0.5) where the figure draws its line, rays 290..306 the robot's own mast
        at 0.13-0.15 m (the cluster at the origin).
    Round 5: NO ray is excused.  Rays 140..143 graze the table's front-right corner (0.6, -0.5): whether they meet the table or pass it
    depends on the base's yaw, and they are on the drawing exactly when the robot stands at the base pose cell 20 prints
    (theta = -0.065; at theta = -0.04 ray 143 is off it, at the oracle's own +0.01 all four are) -- the third MuJoCo output, after
    the wrist depth map and the nav frame (tests/test_notebook_images.py), that the printed pose makes fall into place.  Rounds 3-4
    blamed those rays on the docking station; with today's scene.xml (docking station at (-1, 0), `stretch_scene_docking`) six rays
    behind the robot return 0.9 m where the figure has none: the notebook's scene had no docking station there, which is asserted too."""
    import notebook_images as nbi

    fig = _lidar_figure()
    o = Oracle(open(os.path.join(MODELS, "stretch_scene.smjb"), "rb").read())
    o.arr("ctrl")[:] = HOME_CTRL
    o.step(1600)
    own = o.arr("qpos").copy()
    o.sensors(True)
    off_own, _ = lidar_figure_check(o.arr("lidar").copy(), fig)
    assert off_own == [140, 141, 142, 143], off_own        # (the oracle's own start transient leaves the robot at theta = +0.01)
    o.arr("qpos")[:] = nbi.place_base(own, *nbi.NB_BASE_POSE)
    o.forward()
    o.sensors(True)
    scan = o.arr("lidar").copy()
    off, covered = lidar_figure_check(scan, fig)
    assert off == [], off
    assert covered > 0.85, covered
    at_cut = [i for i in range(360) if scan[i] == fig["cutoff"]]
    assert at_cut == list(range(at_cut[0], at_cut[-1] + 1))                      # one contiguous arc
    assert set(range(145, 271)) <= set(at_cut) and abs(at_cut[-1] - fig["rays_at_cutoff"][-1]) <= 2 and abs(at_cut[0] - fig["rays_at_cutoff"][0]) <= 1   # (ray 144 grazes the table's corner)
    assert not np.any((scan > 1.5) & (scan < 9.6))
    assert np.all(scan[list(range(272, 290)) + list(range(310, 360)) + list(range(0, 40))] == -1.0)
    assert np.all((scan[290:307] > 0.12) & (scan[290:307] < 0.16))
    x, y, th = nbi.NB_BASE_POSE
    o.arr("qpos")[:] = nbi.place_base(own, x, y, -0.04)
    o.forward(); o.sensors(True)
    assert lidar_figure_check(o.arr("lidar").copy(), fig)[0] == [143]
    dock = os.path.join(MODELS, "stretch_scene_docking.smjb")
    if os.path.exists(dock):
        od = Oracle(open(dock, "rb").read())
        od.arr("ctrl")[:] = HOME_CTRL
        od.step(1600)
        od.arr("qpos")[:27] = nbi.place_base(od.arr("qpos")[:27].copy(), *nbi.NB_BASE_POSE)
        od.forward(); od.sensors(True)
        offd, _ = lidar_figure_check(od.arr("lidar").copy(), fig)
        assert len(offd) >= 5 and all(i <= 10 or i >= 350 for i in offd), offd     # the rays that meet the docking station behind the robot
