"""Model compiler, blob container, static-body fusion."""
import json
import os

import numpy as np
import pytest

from conftest import MIX_CTRL, home_qpos
from oracle.oracle import Oracle
from stretch_mujoco_amd import mjcf_compiler as C
from stretch_mujoco_amd import model_blob as B
from stretch_mujoco_amd import model_fuse as F

REF_XML = "/root/reference/stretch_mujoco/models/stretch.xml"


def test_blob_roundtrip():
    a = {"x": np.arange(6, dtype=np.float64).reshape(2, 3), "i": np.array([1, -2, 3], np.int32), "s": np.frombuffer(b"hi", np.uint8)}
    b = B.loads(B.dumps(a))
    assert set(b) == set(a)
    for k in a:
        np.testing.assert_array_equal(a[k], b[k])
        assert a[k].dtype == b[k].dtype
    with pytest.raises(ValueError):
        B.loads(b"not a blob at all")


def test_committed_model_dimensions(blob_full, blob_fused):
    """nq 27, nv 26, nu 10, nbody 38, 21 joints, 5 equalities, 1 tendon, 2 keyframes."""
    m = B.loads(blob_full)
    d = dict(zip("nq nv nu nbody njnt ngeom nsite ncam neq ntendon nwrap nkey npair nhull".split(), m["dims"]))
    assert (d["nq"], d["nv"], d["nu"], d["nbody"], d["njnt"]) == (27, 26, 10, 38, 21)
    assert (d["nsite"], d["ncam"], d["neq"], d["ntendon"], d["nkey"]) == (361, 5, 5, 1, 2)
    assert m["body_mass"].sum() == pytest.approx(33.70, abs=0.01)  # A.3
    names = json.loads(bytes(m["names_json"]).decode())
    assert names["joint"][1:3] == ["joint_right_wheel", "joint_left_wheel"] and names["joint"][-1] == "joint_head_nav_cam"
    assert sorted(names["missing_meshes"]) == ["base_link_8", "link_head_0"]  # .MISSING_LARGE_BLOBS, visual only
    # actuators (A.4): <position> sets biasprm[1] = -kp also when kp is inherited
    np.testing.assert_allclose(m["actuator_gainprm"][:, 0], [20, 20, 400, 150, 20, 50, 20, 4000, 10, 10])
    np.testing.assert_allclose(m["actuator_biasprm"][2], [0, -400, -100])
    np.testing.assert_allclose(m["actuator_biasprm"][3], [0, -150, -10])
    np.testing.assert_allclose(m["key_ctrl"][0], [0, 0, 0.6, 0.1, 0, 0, 0, 0, 0, 0])
    f = B.loads(blob_fused)
    assert f["dims"][3] == 20 and f["dims"][1] == 26
    np.testing.assert_allclose(f["dof_invweight0"], m["dof_invweight0"], rtol=1e-8)
    assert f["body_mass"].sum() == pytest.approx(m["body_mass"].sum(), rel=1e-12)


def test_lidar_site_ordering(blob_full):
    """Index i looks i degrees CCW about the laser body's z from its +X."""
    m = B.loads(blob_full)
    sites = m["sensor_lidar_site"]
    assert len(sites) == 360
    for i in (0, 1, 90, 180, 359):
        z = C.quat2mat(m["site_quat"][sites[i]])[:, 2]
        a = np.deg2rad(i) * (0.0174533 / np.deg2rad(1.0))
        np.testing.assert_allclose(z, [np.cos(a), np.sin(a), 0], atol=1e-9)


def test_fusion_preserves_dynamics(blob_full, blob_fused):
    a, b = Oracle(blob_full), Oracle(blob_fused)
    for o in (a, b):
        o.arr("ctrl")[:] = MIX_CTRL
        o.arr("qpos")[:] = home_qpos(o.arr("qpos"))
    for _ in range(3):
        a.step(100); b.step(100)
        assert np.abs(a.arr("qpos") - b.arr("qpos")).max() < 1e-12
        assert np.abs(a.arr("qvel") - b.arr("qvel")).max() < 1e-11
        assert a.nefc == b.nefc


def test_kernel_tables_are_consistent(blob_fused):
    f = B.loads(blob_fused)
    nv, nb = 26, 20
    assert int(f["k_nldl"][0]) == int((f["k_ldl_i"] >= 0).sum()) <= 5 * 64
    # subtree ranges are contiguous and nested
    par, size = f["body_parentid"], f["k_body_subtreesize"]
    for b in range(1, nb):
        assert par[b] < b and b + size[b] <= par[b] + size[par[b]] if par[b] else True
    # every dof's ancestors are in its body's dof mask
    mask = f["k_body_dofmask_lo"].astype(np.int64) & 0xFFFFFFFF
    for i in range(nv):
        m = int(mask[f["dof_bodyid"][i]])
        assert (m >> i) & 1
        for a in f["k_dof_anc"][f["k_dof_anc_adr"][i]: f["k_dof_anc_adr"][i] + f["k_dof_anc_num"][i]]:
            assert (m >> int(a)) & 1
    # free-joint rotational dofs see only the translational ones in cdof_dot
    vm = f["k_dof_velmask_lo"].astype(np.int64) & 0xFFFFFFFF
    assert [int(v) for v in vm[:6]] == [0, 1, 3, 7, 7, 7]


@pytest.mark.skipif(not os.path.exists(REF_XML), reason="reference MJCF not mounted (GPU box): committed blobs are used")
def test_committed_blob_matches_a_fresh_compile(blob_full):
    m = C.compile_string(C.empty_scene_xml(REF_XML))
    old = B.loads(blob_full)
    for k in ("body_mass", "body_inertia", "body_ipos", "hull_vert", "pair_geom1", "pair_geom2", "dof_invweight0"):
        np.testing.assert_allclose(m[k], old[k], rtol=1e-9, atol=1e-12, err_msg=k)


def test_mesh_inertia_of_a_cube_mesh(tmp_path):
    """Exact and legacy mesh integration agree on a convex solid with the closed form."""
    v = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], float) * [0.1, 0.2, 0.3]
    from scipy.spatial import ConvexHull

    hull = ConvexHull(v)
    faces = hull.simplices.copy()
    cen = v.mean(0)
    for i, f in enumerate(faces):  # outward orientation
        n = np.cross(v[f[1]] - v[f[0]], v[f[2]] - v[f[0]])
        if np.dot(n, v[f[0]] - cen) < 0:
            faces[i] = f[::-1]
    for legacy in (True, False):
        vol, com, I = C.mesh_volume_props(v, faces, legacy=legacy)
        assert vol == pytest.approx(0.2 * 0.4 * 0.6, rel=1e-12)
        np.testing.assert_allclose(com, 0, atol=1e-14)
        np.testing.assert_allclose(np.diag(I), [vol / 3 * (.04 + .09), vol / 3 * (.01 + .09), vol / 3 * (.01 + .04)], rtol=1e-12)


def test_static_lidar_table_sees_the_mast(blob_full):
    """The compiler ray-casts the lidar against the meshes welded to the laser.  Independent check: the mast
    (link_mast.obj: a 38.1 mm square tube with chamfered corners, stretch.xml:245-248) seen from the laser (stretch.xml:274) by a 2-D
    ray / rectangle intersection written here from the MJCF numbers."""
    m = B.loads(blob_full)
    L = m["sensor_lidar_static"]
    assert L.shape == (360,)
    cx, cy, hx, hy = -0.067, 0.135, 0.019063, 0.019063     # body pos; quat (1 1 0 0) maps mesh -z -> +y
    ox, oy = 0.004, 0.0                                     # laser body origin in base_link
    hits = 0
    for i in range(360):
        a = np.deg2rad(i) * (0.0174533 / np.deg2rad(1.0)) + np.pi   # laser body is yawed 180 deg
        dx, dy = np.cos(a), np.sin(a)
        t0, t1 = -np.inf, np.inf
        for o, d, lo, hi in ((ox, dx, cx - hx, cx + hx), (oy, dy, cy - hy, cy + hy)):
            if abs(d) < 1e-12:
                if not lo <= o <= hi:
                    t0, t1 = 1, 0
                continue
            a0, a1 = (lo - o) / d, (hi - o) / d
            t0, t1 = max(t0, min(a0, a1)), min(t1, max(a0, a1))
        if t1 >= max(t0, 0):
            px, py = ox + t0 * dx - cx, oy + t0 * dy - cy      # hit point on the bounding square
            on_flat_face = min(abs(abs(px) - hx), abs(abs(py) - hy)) < 1e-9 and max(min(abs(px), abs(py)), 0) < 0.012
            if on_flat_face:
                hits += 1
                assert L[i] == pytest.approx(t0, abs=1e-4), i
            elif L[i] >= 0:
                assert L[i] >= t0 - 1e-6   # chamfered corner: at or behind the bounding square
        elif L[i] >= 0:
            assert L[i] > 0.02   # some other welded geom; must not be the laser's own body
    assert hits >= 8 and (L >= 0).sum() >= hits


def test_simulator_compiles_a_scene_xml(tmp_path):
    """`scene="…xml"`: the reference lets the user pass any scene that includes stretch.xml (scene_xml_path); here it goes
    through the build's MJCF compiler.  Needs the reference's assets, so it only runs where /root/reference is mounted."""
    import os

    stretch = "/root/reference/stretch_mujoco/models/stretch.xml"
    if not os.path.exists(stretch):
        pytest.skip("reference assets not mounted")
    from stretch_mujoco_amd import StretchBatchSimulator, model_blob

    xml = tmp_path / "my_scene.xml"
    xml.write_text(f'<mujoco model="mine"><include file="{stretch}"/><worldbody><geom name="floor" type="plane" size="0 0 0.05"/>'
                   '<geom name="crate" type="box" pos="1.2 0 0.2" size="0.2 0.3 0.2"/></worldbody></mujoco>')
    sim = StretchBatchSimulator(num_envs=2, device="cpu", scene=str(xml))
    m = sim.model
    assert int(m["dims"][0]) == 27 and int(m["dims"][5]) == 127 and int(m["k_nconvpair"][0]) == 848 + 51
    mm = model_blob.loads(sim._blob)
    assert sum(1 for g in range(127) if mm["geom_type"][g] == 6 and mm["geom_bodyid"][g] == 0) == 1    # the crate, on the world body
    with pytest.raises(Exception):
        sim.start()          # no CPU fallback for the physics path


def test_default_scene_as_shipped_today_compiles_from_the_file_itself_and_steps():
    """models/scene.xml (stretch_mujoco_simulator.py:47: the scene every reference user gets): stretch.xml + docking_station.xml + table
    + two objects, compiled from the reference's file (where the reference is mounted) -- 44 dofs, the docking station's 18 convex
    collision pieces and plate, its visual shell (one of the blobs missing from the checkout) skipped and recorded; the committed blob
    `stretch_scene_docking.smjb` is that compile; on the lane emulator 60 steps follow the oracle."""
    import json
    import os

    import numpy as np

    from conftest import HOME_CTRL, MODELS
    from oracle.oracle import Oracle
    from stretch_mujoco_amd import mjcf_compiler, model_blob, model_fuse

    with open(os.path.join(MODELS, "stretch_scene_docking.smjb"), "rb") as f:
        blob = f.read()
    m = model_blob.loads(blob)
    names = json.loads(model_blob.get_str(m, "names_json"))
    assert int(m["dims"][1]) == 44 and "link_docking_station" in names["body"] and "link_docking_base" in names["missing_meshes"]
    ref = "/root/reference/stretch_mujoco/models/scene.xml"
    if os.path.exists(ref):
        f2 = model_fuse.prepare_for_kernels(mjcf_compiler.compile_file(ref))
        for k in ("body_mass", "geom_size", "pair_geom1", "qpos0", "hull_vert"):
            assert np.array_equal(np.asarray(f2[k]), np.asarray(m[k])), k
    from emul.emul import Emul

    o = Oracle(blob); o.set_option("solver", 2)
    o.arr("ctrl")[:] = HOME_CTRL
    # 43 contacts / 156 rows at rest (the docking station's pieces on the floor): the 50-column build (160 rows) is the primary kernel on
    # the device and hands steps beyond it to the 64-column build (224 rows); the emulator has no hand-over and runs that one
    e = Emul(blob, dict(nq=o.dim("nq"), nv=o.dim("nv"), nu=o.dim("nu"), nlidar=360), num_envs=1, variant="big")
    e.set_option("solver", 2)
    e.ctrl[:, 0] = HOME_CTRL
    from conftest import home_qpos
    q = home_qpos(o.arr("qpos"))          # (past the wrist-in-base start, a chaotic transient)
    o.arr("qpos")[:] = q; e.qpos[:, 0] = q
    o.step(60); e.step(60)
    assert int(e.info[3, 0]) == 0 and (int(e.info[1, 0]), int(e.info[0, 0])) == (o.ncon, o.nefc)
    assert np.abs(e.qpos[:, 0] - o.arr("qpos")).max() < 1e-4
