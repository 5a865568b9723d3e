"""Kitchen stand-in scene (24 static boxes around the robot): model tables, contact parity of
the kernel logic (lane emulator) against the fp64 oracle when the arm runs into the counter, and what the lidar sees."""
import numpy as np

from conftest import home_qpos
from emul.emul import Emul
from oracle.oracle import Oracle
from stretch_mujoco_amd import model_blob

DIMS = dict(nq=34, nv=32, nu=10, nlidar=360)   # robot 27/26 + one free ball 7/6


def test_scene_tables(blob_kitchen):
    m = model_blob.loads(blob_kitchen)
    assert int(m["dims"][5]) == 126 + 24 + 1 and int(m["dims"][0]) == 34 and int(m["dims"][1]) == 32
    assert int(m["k_ncgeom"][0]) == 51 + 24 + 1 and int(m["k_nconvpair"][0]) == 848 + 51 * 24 + 51 + 24   # ball vs robot, ball vs fixtures
    assert int(m["k_nplanepair"][0]) == 51 + 1                  # static boxes do not pair with the static floor, the ball does
    assert int(m["k_nrgeom"][0]) == 74 + 24 + 1 and int(m["k_nlgeom"][0]) == 96 + 24 + 1
    assert int(m["k_nroot"][0]) == 2                            # two kinematic trees: robot and ball
    boxes = [g for g in range(151) if m["geom_type"][g] == 6 and m["geom_bodyid"][g] == 0]
    assert len(boxes) == 24


def test_arm_runs_into_the_counter(blob_kitchen):
    """ctrl arm 0.5 at lift 0.6: the gripper meets the counter front (y = -0.78) and the arm stalls.  State-synchronised
    along the oracle's trajectory: same contacts (count, depth, point; normals of box faces are exact) and, where depths
    agree to 1e-6, the same acceleration."""
    ctrl = [0, 0, 0.6, 0.5, 0, 0, 0, 0, 0, 0]
    o = Oracle(blob_kitchen); o.set_option("solver", 2)
    o.arr("ctrl")[:] = ctrl
    o.arr("qpos")[:] = home_qpos(o.arr("qpos"))
    e = Emul(blob_kitchen, DIMS, num_envs=1); e.set_option("solver", 2)
    e.ctrl[:, 0] = ctrl
    touching = compared = 0
    for k in range(0, 400, 10):
        e.qpos[:, 0] = o.arr("qpos"); e.qvel[:, 0] = o.arr("qvel"); e.warm[:, 0] = o.arr("qacc_warmstart")
        e.step(1)
        o.forward()
        n = o.ncon
        assert int(e.info[1, 0]) == n, k
        co = o.arr("contact").reshape(n, -1)
        ce = e.debug[1600:1600 + 8 * n, 0].reshape(n, 8)
        np.testing.assert_allclose(ce[:, 0], co[:, 0], atol=2e-3)
        np.testing.assert_allclose(ce[:, 1:4], co[:, 1:4], atol=1e-2)
        cosn = np.sum(ce[:, 4:7] * co[:, 4:7], axis=1)
        assert (cosn > 0.0).all(), (k, cosn)
        touching += int(n > 5)
        if int(e.info[0, 0]) == o.nefc and (cosn > 0.9999).all() and np.abs(ce[:, 0] - co[:, 0]).max() < 1e-6:
            qa = o.arr("qacc")
            assert np.abs(e.debug[1056:1088, 0] - qa)[:18].max() < 2e-2 * max(1.0, np.abs(qa[:18]).max()), k
            compared += 1
        o.step(10)
    arm = o.arr("qpos")[10:14].sum()
    assert touching >= 10 and compared >= 10, (touching, compared)
    assert 0.3 < arm < 0.45, arm                         # stalled against the counter, short of the 0.5 target
    # free-running fp32 trajectory ends in the same place
    e = Emul(blob_kitchen, DIMS, num_envs=1); e.set_option("solver", 2)
    e.qpos[:, 0] = home_qpos(o.arr("qpos")); e.ctrl[:, 0] = ctrl
    e.qpos[:, 0] = home_qpos(model_blob.loads(blob_kitchen)["qpos0"])
    e.step(400)
    assert abs(e.qpos[10:14, 0].sum() - arm) < 0.02


def test_lidar_sees_the_room(blob_kitchen):
    m = model_blob.loads(blob_kitchen)
    o = Oracle(blob_kitchen)
    o.arr("qpos")[:] = home_qpos(m["qpos0"])
    o.forward(); o.sensors(True)
    L = o.arr("lidar")
    assert (L > 0).all() and L.max() < 3.8                        # walls all around: no ray escapes the 5 m room
    sid = m["sensor_lidar_site"]
    P = o.arr("site_xpos").reshape(-1, 3)[sid]
    Z = o.arr("site_xmat").reshape(-1, 3, 3)[sid][:, :, 2]
    i = int(np.argmin(Z[:, 1]))                                    # the ray pointing most nearly along -y: counter front at y = -0.78
    assert abs(Z[i, 1] + 1) < 1e-3 and abs(L[i] - (P[i, 1] + 0.78)) < 2e-3, (Z[i], L[i])
    j = int(np.argmax(Z[:, 1]))                                    # +y: table leg? no -- clear path to the wall at y = 2.5
    assert abs(L[j] - (2.5 - P[j, 1])) < 0.02 or L[j] < 2.5


def test_ball_rests_on_the_counter_and_is_pushed(blob_kitchen):
    """The free ball (second kinematic tree, sphere-box contact in closed form) settles on the countertop at the height the
    soft contact allows; kernel logic and oracle agree along the way, and a shove makes it roll (its dofs are live)."""
    m = model_blob.loads(blob_kitchen)
    o = Oracle(blob_kitchen); o.set_option("solver", 2)
    q = home_qpos(m["qpos0"])
    ctrl = [0, 0, 0.6, 0.1, 0, 0, 0, 0, 0, 0]
    o.arr("qpos")[:] = q; o.arr("ctrl")[:] = ctrl
    e = Emul(blob_kitchen, DIMS, num_envs=1); e.set_option("solver", 2)
    e.qpos[:, 0] = q; e.ctrl[:, 0] = ctrl
    o.step(400); e.step(400)
    zb = o.arr("qpos")[29]
    assert 0.9590 < zb < 0.9601                                   # countertop at 0.92, radius 0.04, sub-millimetre sink
    assert abs(e.qpos[29, 0] - zb) < 2e-5 and np.abs(e.qpos[:, 0] - o.arr("qpos")).max() < 1e-4
    assert np.abs(o.arr("qvel")[26:]).max() < 1e-3
    o.arr("qvel")[26] = 0.5; e.qvel[26, 0] = 0.5                  # shove along x
    o.step(100); e.step(100)
    assert o.arr("qpos")[27] > -0.3 + 0.02 and abs(e.qpos[27, 0] - o.arr("qpos")[27]) < 2e-3
    assert abs(o.arr("qvel")[30]) > 1.0                           # rolling: spin about y has built up from friction



def test_three_envs_per_cu_build_equals_the_tall_variant(blob_kitchen):
    """smj_create runs contact-rich 32-dof scenes on the 128-row / 44-contact build of the tall variant (53 KB of LDS: three envs
    per CU) and hands steps beyond that to the 160-row build.  Same source, smaller arrays: while a rollout stays inside 128 rows
    the two builds must produce the same states, bit for bit -- the arm driven into the counter, 120 steps."""
    from stretch_mujoco_amd import model_blob

    m = model_blob.loads(blob_kitchen)
    out = {}
    for variant in ("tall", "mid"):
        e = Emul(blob_kitchen, DIMS, num_envs=2, variant=variant); e.set_option("solver", 2)
        e.qpos[:] = np.asarray(m["qpos0"], np.float32)[:, None]
        e.ctrl[:, 0] = np.array([0.5, -0.5, 0.9, 0.5, 1.0, -0.5, 0.3, 0.0, 0.2, -0.3], np.float32)
        e.ctrl[:, 1] = np.array([-1.0, 1.0, 0.6, 0.3, -1.0, 0.2, -0.3, 0.02, -0.5, 0.3], np.float32)
        rows = 0
        for _ in range(120):
            e.step(1)
            rows = max(rows, int(e.info[0].max()))
        out[variant] = (e.qpos.copy(), e.qvel.copy(), e.info.copy(), rows)
    assert out["tall"][3] <= 128 and int((out["tall"][2][3] & 3).max()) == 0
    for k in range(3):
        assert np.array_equal(out["tall"][k], out["mid"][k])
