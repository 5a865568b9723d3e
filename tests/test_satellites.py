"""Satellites and the static-geometry broadphase (csrc/smj_sat.h, model_fuse.find_satellites / _static_grid_tables): free objects
and single-joint fixture parts run as one lane each beside the robot's dense 32-column problem; the world body's collision geoms
stay out of the pair table the kernels scan.  Everything here is the kernel source against the fp64 oracle -- through the lane
emulator on the CPU, through libsmj.so on the GPU (`-m gpu`) -- on the reference's own scene.xml, the kitchen stand-in with four
objects, the small kitchen export and the generated kitchen at Robocasa scale (tests/kitchen_robocasa_fixture.py:
44 fixture bodies, 307 collision geoms, 8 articulated parts, 8 free objects).  TEST INFRASTRUCTURE (imports oracle/)."""
import ctypes
import os

import numpy as np
import pytest

import rollout_common as rc
import stretch_mujoco_amd.model_blob as mb
from conftest import MODELS

SAT_SCENES = ["stretch_scene_sat", "stretch_kitchen4_sat", "stretch_kitchen_export_sat", "stretch_kitchen_robocasa"]


def _blob(scene):
    with open(f"{MODELS}/{scene}.smjb", "rb") as f:
        b = f.read()
    return b, mb.loads(b)


def test_kitchen_fixture_is_at_robocasa_scale_and_goes_through_the_import():
    """What the verdict of round 3 asked of the document: >= 40 fixture bodies, >= 300 collision geoms with convex mesh pieces
    among them, >= 8 articulated fixture joints, >= 8 free objects, marker geoms -- and that robocasa_import's clean-up
    (robocasa_gen.py:242-280) applies: robot body, actuators, sensors and the option section gone, markers invisible."""
    import xml.etree.ElementTree as ET
    from kitchen_robocasa_fixture import kitchen_xml
    from stretch_mujoco_amd.robocasa_import import convert_kitchen_xml

    xml, st = kitchen_xml()
    assert st["fixture_bodies"] >= 40 and st["collision_geoms"] >= 300 and st["mesh_collision_geoms"] >= 30
    assert st["articulated"] >= 8 and st["free_objects"] >= 8
    out, pose = convert_kitchen_xml(xml, "stretch.xml")
    root = ET.fromstring(out)
    assert root[0].tag == "include" and root[0].get("file") == "stretch.xml"
    assert not root.findall("actuator") and not root.findall("sensor") and not root.findall("option")
    assert all(b.get("name") != "robot0_base" for b in root.iter("body"))
    assert pose["pos"] == [0.3, -0.8, 0.0]
    assert not any(g.get("rgba") in ("0.5 0 0 0.5", "0.5 0 0 1") for g in root.iter("geom"))   # collision / marker colours -> alpha 0
    joints = [j.get("type") for j in root.iter("joint")]
    assert joints.count("hinge") + joints.count("slide") == st["articulated"] and len(list(root.iter("freejoint"))) == st["free_objects"]


def test_xml_scenes_pick_the_satellite_build_when_they_outgrow_the_dense_ones():
    """A scene handed over as an .xml path (the reference's `scene_xml_path`) is prepared with satellites = "auto": scene.xml
    (38 dofs, 20-odd geoms) fits the dense builds and stays there; the kitchen at Robocasa scale (82 dofs, 307 collision geoms)
    does not and comes out with 16 satellites and the static-geometry tables -- a blob equal to the committed one."""
    import os

    from kitchen_robocasa_fixture import kitchen_xml
    from stretch_mujoco_amd import mjcf_compiler as C, model_fuse as F
    from stretch_mujoco_amd.robocasa_import import convert_kitchen_xml

    ref = "/root/reference/stretch_mujoco/models"
    if not os.path.isdir(ref):
        pytest.skip("the reference's MJCF is not on this box")
    small = F.prepare_for_kernels(C.compile_file(os.path.join(ref, "scene.xml")), satellites="auto")
    assert int(small["k_nsat"][0]) == 0
    rx, _ = kitchen_xml()
    rkx, pose = convert_kitchen_xml(rx, os.path.join(ref, "stretch.xml"))
    rk = C.compile_string(rkx)
    rk["qpos0"][0:3] = pose["pos"]; rk["qpos0"][3:7] = pose["quat"]
    big = F.prepare_for_kernels(rk, satellites="auto")
    _, committed = _blob("stretch_kitchen_robocasa")
    assert int(big["k_nsat"][0]) == 16 == int(committed["k_nsat"][0])
    for k in ("k_sat_i", "k_main_dims", "k_nstatpair", "dims"):
        assert np.array_equal(np.asarray(big[k]), np.asarray(committed[k])), k


def test_compiled_kitchen_tables():
    """The committed blob: 16 satellites behind the robot's 26 dofs, the static world split off the scanned pair table, and the
    (moving geom, static geom) -> pair look-up consistent with the pair table MuJoCo's filters leave."""
    _, m = _blob("stretch_kitchen_robocasa")
    assert int(m["dims"][1]) == 82 and int(m["dims"][0]) == 91
    assert int(m["k_nsat"][0]) == 16 and list(m["k_main_dims"]) == [27, 26, 20, 21]
    si = np.asarray(m["k_sat_i"]).reshape(16, -1)
    assert sorted(si[:, 4].tolist()) == [1] * 8 + [6] * 8                       # 8 single-joint parts, 8 free objects
    assert (np.diff(si[:, 3]) > 0).all() and si[0, 3] == 26                     # dofs in order, right behind the robot's
    nsg, ncg = int(m["k_nsgeom"][0]), int(m["k_ncgeom"][0])
    assert nsg >= 250 and ncg <= 128 and int(m["k_nstatpair"][0]) > 20000 and int(m["k_nconvpair"][0]) < 4000
    sp, tab = np.asarray(m["k_statpair"])[:-1], np.asarray(m["k_spair"]).reshape(ncg, nsg)
    gb, cg, sg = np.asarray(m["geom_bodyid"]), np.asarray(m["k_cgeom"])[:ncg], np.asarray(m["k_sgeom"])[:nsg]
    assert (gb[sg] == 0).all() and (gb[cg] > 0).all()
    for i in np.random.default_rng(0).choice(len(sp), 300, replace=False):      # every static pair is reachable through the look-up
        g1, g2 = int(m["pair_geom1"][sp[i]]), int(m["pair_geom2"][sp[i]])
        s_, d_ = (g1, g2) if gb[g1] == 0 else (g2, g1)
        assert tab[list(cg).index(d_), list(sg).index(s_)] == i
    assert (tab >= 0).sum() == len(sp)
    # the uniform grid lists every static geom in every cell of its range
    dims, adr, lst, rng = np.asarray(m["k_grid"]), np.asarray(m["k_grid_adr"]), np.asarray(m["k_grid_list"]), np.asarray(m["k_sg_cell"]).reshape(-1, 6)
    assert len(adr) == int(np.prod(dims)) + 1
    for g in (0, nsg // 2, nsg - 1):
        a, b = rng[g, :3], rng[g, 3:]
        for cell in ((a[2] * dims[1] + a[1]) * dims[0] + a[0], (b[2] * dims[1] + b[1]) * dims[0] + b[0]):
            assert g in lst[adr[cell]:adr[cell + 1]]


def test_satellites_of_the_reference_scene():
    """scene.xml (the reference's default scene): the two free objects become satellites, the robot is the main tree; the small
    kitchen export: door (hinge), drawer (slide) and three objects."""
    _, m = _blob("stretch_scene_sat")
    assert int(m["k_nsat"][0]) == 2 and list(m["k_main_dims"]) == [27, 26, 20, 21]
    _, e = _blob("stretch_kitchen_export_sat")
    si = np.asarray(e["k_sat_i"]).reshape(int(e["k_nsat"][0]), -1)
    assert si[:, 1].tolist() == [3, 2, 0, 0, 0] and si[:, 4].tolist() == [1, 1, 6, 6, 6]


def _sync(scene, B, W, seed=5, variant=None):
    blob, model = _blob(scene)
    be = rc.EmulBackend(blob, B, variant=variant)
    rel, events = rc.state_synchronised(be, blob, model, B, W, seed=seed, twin=True)   # (the kernel keeps manifolds by default: against the oracle's twin of that rule)
    return be, rel, events


@pytest.mark.parametrize("scene", SAT_SCENES)
def test_emul_satellite_build_state_synchronised(scene):
    """The bench workload, state re-synchronised with the oracle before every step: one-step accelerations of EVERY dof (robot and
    satellites), contact lists pair by pair.  The kitchen runs on the 32-satellite build here (the emulator has no escalation;
    the 16-satellite build hands steps beyond its rows to that one on the device)."""
    be, rel, events = _sync(scene, 2, 3, variant="sat32" if scene == "stretch_kitchen_robocasa" else None)
    c = rc.state_synchronised.contacts
    ro, ob = rc.state_synchronised.rel_robot, rc.state_synchronised.rel_obj
    print(f"\n[{scene}] {len(rel)} env-steps: rel qacc, robot dofs p50 {np.percentile(ro, 50):.1e} p99 {np.percentile(ro, 99):.1e} max {ro.max():.1e}; satellite dofs (own scale) "
          f"p50 {np.percentile(ob, 50):.1e} p99 {np.percentile(ob, 99):.1e} max {ob.max():.1e}; events {len(events)}; contacts {c['n']}, steps with differing pair lists {c['mismatched_steps']}")
    assert np.percentile(ro, 99) < rc.TYPICAL_TOL * 4 and all(ev["explained"] and ev["flags"] == 0 for ev in events)
    rc.assert_object_dofs()
    assert len(rc.gross_events(events)) <= 0.01 * len(rel) + 2 and c["mismatched_steps"] <= 0.01 * len(rel) + 1
    assert c["n"] > 1000
    assert int(be.e.info[3].max()) == 0


def test_emul_satellite_build_equals_the_dense_build():
    """scene.xml on the satellite build and on the 38-column dense build (big38), same inputs, 150 free-running steps with the
    arm moving: the same algorithm, so the same states to fp32 rounding."""
    from emul.emul import Emul
    from oracle.oracle import Oracle

    outs = []
    for scene in ("stretch_scene_sat", "stretch_scene"):
        blob, _ = _blob(scene)
        o = Oracle(blob)
        e = Emul(blob, dict(nq=o.dim("nq"), nv=o.dim("nv"), nu=o.dim("nu"), nlidar=360), num_envs=1)
        e.set_option("solver", 2)
        e.qpos[:, 0] = o.arr("qpos"); e.qpos[9, 0] = 0.6; e.qpos[10:14, 0] = 0.025
        e.ctrl[:, 0] = [1.0, -0.5, 0.6, 0.1, 0.5, -0.3, 0.2, 0.01, 0.3, -0.2]
        e.step(150)
        outs.append((e.variant, e.qpos[:, 0].copy(), e.qvel[:, 0].copy(), int(e.info[3, 0])))
    assert outs[0][0] == "sat" and outs[1][0] == "big38"
    assert outs[0][3] == 0 and outs[1][3] == 0
    assert np.abs(outs[0][1] - outs[1][1]).max() < 2e-5 and np.abs(outs[0][2] - outs[1][2]).max() < 2e-3


def test_emul_dense_extension_and_manifold_cache_are_exercised():
    """The two mechanisms that only some steps use: the dense extension (a satellite coupled to the robot: Schur complement of its
    block) and the contact-manifold cache (a pair whose two bodies have not moved reuses its contacts).  With the cache switched off
    the one-step accelerations move by no more than the pose tolerance allows."""
    from emul.emul import lib

    L = lib("sat")
    L.emul_ext_steps.restype = ctypes.c_long; L.emul_mc_hits.restype = ctypes.c_long
    e0, h0 = L.emul_ext_steps(), L.emul_mc_hits()
    be, rel, events = _sync("stretch_scene_sat", 4, 4)
    assert L.emul_ext_steps() - e0 > 20, "no step coupled a satellite to the robot"
    assert L.emul_mc_hits() - h0 > 500, "the manifold cache was never hit"
    assert all(ev["explained"] for ev in events)
    # cache on / off on identical states
    blob, model = _blob("stretch_kitchen4_sat")
    outs = []
    for on in (1, 0):
        b = rc.EmulBackend(blob, 1)
        b.e.set_option("manifold_cache", on)
        orc = rc.settled_oracles(blob, 1)
        acc = []
        for _ in range(30):
            b.upload(*rc.state_of(orc))
            b.step(1)
            acc.append(b.download()["qacc"][:, 0].copy())
            orc[0].step(1)
        outs.append(np.array(acc))
    d = np.abs(outs[0] - outs[1]).max(1) / np.maximum(1.0, np.abs(outs[1]).max(1))
    assert d.max() < 2e-3, d.max()


def test_emul_kept_manifolds_ride_with_the_bodies_to_first_order():
    """Round 5: a cached manifold is carried along by the two bodies' motion since it was built (smj_mc_carry: the depth changes by
    the normal component of the relative displacement at the contact point), which is what lets the pose tolerance be 2e-5 instead of
    3e-7 -- MPR's own 1e-6 m tolerance made resting bodies jitter by 1e-6 a step, so the round-4 cache missed the pair with the most
    expensive manifold (five multiccd points) on every step of a settled kitchen.  Settled kitchen at Robocasa scale, state
    re-synchronised with the oracle before every step: 7 of the 8 object-on-fixture pairs come from the cache (the eighth never
    touches); the DEPTHS of all contacts are the oracle's, recomputed from scratch in fp64, to 2e-6.  Contact POSITIONS: the oracle's
    own manifold of one resting pair flips between two solutions 1.1 cm apart from one step to the next (MPR on a face contact); the
    kept manifold is one of them, so positions are asserted against the oracle's own step-to-step scatter, and a step whose
    accelerations differ is the oracle's on the kernel's contact list."""
    from emul.emul import lib

    L = lib("sat")
    L.emul_mc_hits.restype = ctypes.c_long
    blob, model = _blob("stretch_kitchen_robocasa")
    be = rc.EmulBackend(blob, 1, variant="sat")
    orc = rc.settled_oracles(blob, 1, settle=300)
    nu = orc[0].dim("nu")
    ctrl = np.array(rc.HOME[:nu], np.float32)
    be.set_ctrl(ctrl[:, None])
    cstat = dict(n=0, depth=[], pos=[], cosn=[], mismatched_steps=0)
    own, prev, unexplained, gross = 0.0, None, 0, 0
    for k in range(45):
        st = rc.state_of(orc)
        be.upload(*st)
        if k == 5:
            h0 = L.emul_mc_hits()
        be.step(1)
        out = be.download()
        orc[0].step(1)
        n = orc[0].ncon
        co = orc[0].arr("contact").reshape(n, -1).copy()
        pairs = [tuple(int(v) for v in co[i, -2:].copy().view(np.int32)[1:3]) for i in range(n)]
        if prev is not None and prev[0] == pairs:
            own = max(own, float(np.abs(co[:, 1:4] - prev[1][:, 1:4]).max()))
        prev = (pairs, co)
        qa = orc[0].arr("qacc")
        qk = out["qacc"][:len(qa), 0]
        if k >= 5:
            rc._compare_contacts(cstat, out["contacts"][:, 0], int(out["info"][1, 0]), orc[0])
            if rc.step_error(qk, qa) > rc.EVENT_TOL:
                gross += 1
                ok, err = rc._same_contacts_same_dynamics(blob, 2, (st[0][:, 0], st[1][:, 0], st[2][:, 0]), ctrl, qk, out["contacts"][:, 0], int(out["info"][1, 0]))
                unexplained += not ok
    hits = (L.emul_mc_hits() - h0) / 40
    print(f"cache hits per step {hits:.2f}; contacts compared {cstat['n']}, |ddist| max {max(cstat['depth']):.1e}, |dpos| p90 {np.percentile(cstat['pos'], 90):.1e} "
          f"max {max(cstat['pos']):.1e} (the oracle's own step-to-step scatter: {own:.1e}); steps beyond the event tolerance {gross}, unexplained {unexplained}")
    assert hits > 6.5
    assert cstat["mismatched_steps"] == 0 and cstat["n"] > 1200
    assert max(cstat["depth"]) < 2e-6
    assert np.percentile(cstat["pos"], 90) < 2e-4 and max(cstat["pos"]) < 1.05 * own + 2e-4
    assert unexplained == 0 and int(be.e.info[3].max()) == 0


def _pgs_backends(scenes, B):
    out = []
    for sc in scenes:
        blob, model = _blob(sc)
        be = rc.EmulBackend(blob, B, solver=0)
        be.e.set_option("qcqp_exact", 1)   # MuJoCo's own QCQP iteration (the oracle's), so that the sweeps are the only difference
        be.e.set_option("pgs_dual_warmstart", 0)   # MuJoCo's warm start on both sides (the dense builds have no other)
        be.e.set_option("manifold_cache", 0)       # (the two builds file kept manifolds under different slots -- different evictions, a different manifold kept now and then; these tests are about the sweeps)
        out.append((blob, model, be))
    return out


def test_emul_pgs_islands_equal_the_serial_sweep():
    """PGS on the satellite build (smj_sat_pgs.h): the dense system -- the robot's rows and those of satellites coupled to it -- is
    swept lane = row, every other satellite sweeps its own rows on its own lane, all in the same iteration, one improvement sum,
    one termination test.  Islands do not see each other's rows, so the result must be the serial sweep's: on identical states
    (the oracle's, uploaded before every step) the satellite build and the 50-column dense build (every row in one sweep, big50)
    give the same accelerations to fp32 rounding in 9 steps of 10, and both stay as close to the fp64 PGS oracle as each other.
    (100 sweeps do not converge PGS in a kitchen: the remainder carries the rounding of the sweeps, hence bounds of 1e-2, not 1e-4.)"""
    B = 1
    (blob, model, sat), (_, _, dense) = _pgs_backends(["stretch_kitchen4_sat", "stretch_kitchen4"], B)
    assert sat.e.variant == "sat" and dense.e.variant == "big50"
    orc = rc.settled_oracles(blob, B, 0, settle=300)
    nu, nv = orc[0].dim("nu"), orc[0].dim("nv")
    sched = rc.ctrl_schedule(model, nu, B, 2, 5)
    d_sd, d_so, d_do, iters = [], [], [], []
    for w in range(2):
        for be in (sat, dense):
            be.set_ctrl(sched[w])
        orc[0].arr("ctrl")[:nu] = sched[w][:, 0]
        for _ in range(40):
            st = rc.state_of(orc)
            q = []
            for be in (sat, dense):
                be.upload(*st); be.step(1); q.append(be.download()["qacc"][:nv, 0].copy())
            iters.append((int(sat.e.info[2, 0]), int(dense.e.info[2, 0])))
            orc[0].step(1)
            qa = orc[0].arr("qacc")
            sc = max(1.0, np.abs(qa).max())
            d_sd.append(np.abs(q[0] - q[1]).max() / sc); d_so.append(np.abs(q[0] - qa).max() / sc); d_do.append(np.abs(q[1] - qa).max() / sc)
    d_sd, d_so, d_do = np.array(d_sd), np.array(d_so), np.array(d_do)
    print(f"\nsatellite vs dense build: p50 {np.percentile(d_sd, 50):.1e} p90 {np.percentile(d_sd, 90):.1e} max {d_sd.max():.1e}; "
          f"vs oracle: satellite p50 {np.percentile(d_so, 50):.1e} max {d_so.max():.1e}, dense p50 {np.percentile(d_do, 50):.1e} max {d_do.max():.1e}")
    assert int(sat.e.info[3, 0]) == 0
    # (the two builds' CONTACT LISTS are bit-identical on every one of these steps -- checked, tools/objdof_probe.py's sibling in round 6 --; what
    # differs is the rounding of 100 unconverged sweeps in a different row order, amplified in the violent phases: 12 of 80 steps above 1e-4 with
    # round 6's exact face normals (smj_step_impl.h mpr_penetration), 7 of 80 with the fp32-noisy ones of rounds 1-5)
    assert np.percentile(d_sd, 75) < 1e-4 and np.percentile(d_sd, 90) < 1e-3 and d_sd.max() < 2e-2
    assert np.percentile(d_so, 50) < 5e-4 and d_so.max() < 2e-2 and d_so.max() < 2 * d_do.max() + 1e-3
    assert all(a == b for a, b in iters[:5]), iters[:5]   # one iteration count for the whole system, as the serial sweep's


@pytest.mark.parametrize("dual", [0, 1])
@pytest.mark.parametrize("scene,variant", [("stretch_kitchen_export_sat", None), ("stretch_kitchen_robocasa", "sat32")])
def test_emul_pgs_satellite_build_state_synchronised(scene, variant, dual):
    """PGS with hinged / sliding fixture parts among the satellites (friction-loss and limit rows on satellite lanes) and, in the
    kitchen at Robocasa scale, objects the robot pushes (coupled satellites in the dense system): one-step accelerations of every
    dof against the fp64 PGS oracle, contact lists pair by pair.  dual = 0: MuJoCo's warm start on both sides -- 100 sweeps on every
    step, neither side converged, the remainder carries the sweeps' rounding.  dual = 1 (the kernels' default, round 5): both sides may
    also start from the previous step's forces (option pgs_dual_warmstart, same fixed point): a quarter of the sweeps, and the two
    CONVERGED answers agree to p99 < 5e-3 (VERDICT r4 item 2's bound; observed 2.3e-3, max 2.8e-3 where dual = 0 gives 5.7e-3 / 1.2e-2)."""
    blob, model = _blob(scene)
    be = rc.EmulBackend(blob, 2, solver=0, variant=variant)
    be.e.set_option("qcqp_exact", 1)
    be.e.set_option("pgs_dual_warmstart", dual)
    sweeps = []
    step = be.step
    be.step = lambda n: (step(n), sweeps.append(float(be.e.info[2].mean())))[0]
    _, events = rc.state_synchronised(be, blob, model, 2, 2, seed=5, solver=0, oracle_options={"pgs_dual_warmstart": dual}, twin=True)
    rel, ob, so = rc.state_synchronised.rel_robot, rc.state_synchronised.rel_obj, rc.state_synchronised.same_obj   # (the bounds below: the robot's dofs; the satellites' dofs on their own scale are printed and bounded separately)
    print(f"   satellite dofs on their own scale: own narrowphase p50 {np.percentile(ob, 50):.1e} p99 {np.percentile(ob, 99):.1e} max {ob.max():.1e}; on the kernel's contact list p50 {np.percentile(so, 50):.1e} p99 {np.percentile(so, 99):.1e} max {so.max():.1e}")
    # (round 6: p99 was 4.1 / 0.3 on the objects' own scale until the ray update of the block solvers ended exactly at the cone's apex -- an fp32
    # residue of 2^-31 of the old force left a contact switched off for the rest of the solve, smj_sat_pgs.h pgs_block_lane; observed now <= 1e-2)
    assert np.percentile(so, 50) < 5e-3 and np.percentile(ob, 99) < 3e-2 and np.percentile(so, 99) < 3e-2
    c = rc.state_synchronised.contacts
    print(f"\n[{scene}, PGS, dual warm start {dual}] {len(rel)} env-steps: rel qacc p50 {np.percentile(rel, 50):.1e} p90 {np.percentile(rel, 90):.1e} p99 {np.percentile(rel, 99):.1e} "
          f"max {rel.max():.1e}; sweeps per step {np.mean(sweeps):.1f}; events {len(events)}; contacts {c['n']}, steps with differing pair lists {c['mismatched_steps']}")
    assert int(be.e.info[3].max()) == 0
    assert np.percentile(rel, 50) < 5e-4 and np.percentile(rel, 90) < 3e-3 and rel.max() < 3e-2
    it = rc.state_synchronised.iters
    conv = (it[:, 0] < 100) & (it[:, 1] < 100)
    print(f"   steps on which both sides left the sweeps before the cap of 100: {conv.mean():.2f}" + (f"; there: rel qacc p99 {np.percentile(rel[conv], 99):.1e} max {rel[conv].max():.1e}" if conv.any() else ""))
    if dual:
        assert np.percentile(rel, 99) < 5e-3 and rel.max() < 1e-2 and np.mean(sweeps) < 60
        assert conv.mean() > 0.6 and np.percentile(rel[conv], 99) < 5e-3
    else:
        assert np.mean(sweeps) > 90
    assert c["mismatched_steps"] <= 0.01 * len(rel) + 1 and c["n"] > 1000


def test_emul_more_sliding_contacts_than_cone_hessian_blocks():
    """A state the long soak of round 4 ran into (tests/golden/toppled_robot_state.npz: stretch_kitchen4 on the satellite build,
    the robot on its side, 37 contacts, most of them sliding): more contacts in the middle zone of their cone than the 16-block
    cone-Hessian pool holds.  The first version left the surplus out of H; Newton crept to its cap of 100 iterations and the robot
    rose at 2 m/s where the oracle's stays down.  With the diagonal stand-in the iteration count is the oracle's and 30 free-running
    steps stay on the oracle's."""
    import os
    from emul.emul import Emul
    from oracle.oracle import Oracle

    blob, _ = _blob("stretch_kitchen4_sat")
    st = np.load(os.path.join(os.path.dirname(__file__), "golden", "toppled_robot_state.npz"))
    o = Oracle(blob); o.set_option("solver", 2)
    nu = o.dim("nu")
    o.arr("qpos")[:] = st["qpos"]; o.arr("qvel")[:] = st["qvel"]; o.arr("qacc_warmstart")[:] = st["warm"]; o.arr("ctrl")[:nu] = st["ctrl"]
    e = Emul(blob, dict(nq=o.dim("nq"), nv=o.dim("nv"), nu=nu, nlidar=360), num_envs=1, variant="sat"); e.set_option("solver", 2)
    e.ctrl[:, 0] = st["ctrl"]; e.qpos[:, 0] = st["qpos"]; e.qvel[:, 0] = st["qvel"]; e.warm[:, 0] = st["warm"]
    for k in range(30):
        e.step(1); o.step(1)
        assert int(e.info[2, 0]) < 20 and int(e.info[3, 0]) == 0, (k, e.info[:, 0])
        if k == 0:
            assert (int(e.info[0, 0]), int(e.info[1, 0])) == (o.nefc, o.ncon) == (138, 37)
    assert abs(float(e.qpos[2, 0]) - o.arr("qpos")[2]) < 1e-4 and np.abs(e.qpos[:27, 0] - o.arr("qpos")[:27]).max() < 2e-3


def test_oracle_manifold_keep_rule_against_the_unmodified_oracle():
    """What the kernels' manifold cache CHANGES, measured without a kernel: the fp64 oracle with the twin of the rule (option
    manifold_keep: a convex pair whose bodies stand within 2e-5 of the poses its manifold was built at keeps it, carried to first order)
    against the unmodified oracle, free-running from the same settled state under the bench's random actions, the kitchen at Robocasa
    scale.  The rule does take effect (thousands of kept manifolds) and moves nothing outside the drift band: the resting objects' qpos
    stay within 1e-4 of the unmodified run (measured 3.2e-5: the unmodified oracle's bowl jitters on a manifold that flips between two
    solutions from step to step, the kept one sits still), and so do the robot's (1e-9 until it touches an object; 7.5e-5 in the env where it does), over 500 steps."""
    blob, model = _blob("stretch_kitchen_robocasa")
    B, W = 3, 10
    a = rc.settled_oracles(blob, B, 2)
    b = rc.settled_oracles(blob, B, 2, late_options=rc.TWIN)
    nu = a[0].dim("nu")
    sched = rc.ctrl_schedule(model, nu, B, W, 7)
    worst_robot = worst_obj = 0.0
    for w in range(W):
        for e in range(B):
            for o in (a[e], b[e]):
                o.arr("ctrl")[:nu] = sched[w][:, e]
                o.step(rc.HOLD)
        d = np.abs(np.stack([x.arr("qpos") for x in a], 1) - np.stack([x.arr("qpos") for x in b], 1))
        worst_robot, worst_obj = max(worst_robot, d[:27].max()), max(worst_obj, d[27:].max())
    hits = [int(x.iarr("mc_hits")[0]) for x in b]
    print(f"\nkeep rule vs unmodified oracle, {B} envs x {W * rc.HOLD} steps: max |dqpos| robot {worst_robot:.1e}, objects {worst_obj:.1e}; kept manifolds used {hits}")
    assert min(hits) > 1000 and all(int(x.iarr("mc_hits")[0]) == 0 for x in a)
    assert worst_robot < 1e-4 and worst_obj < 1e-4


# ------------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("scene", SAT_SCENES)
def test_gpu_satellite_build_state_synchronised(scene):
    """8 envs x 300 steps on the device, the oracle's state uploaded before every step: accelerations of all 38 / 50 / 46 / 82 dofs
    and the contact lists, with the hand-over from the 16-satellite build to the 32-satellite one where a step needs it."""
    blob, model = _blob(scene)
    # Twice.  With the manifold cache OFF every manifold is built on the step's own poses, as the unmodified oracle's are: the bound on the
    # dynamics and on the narrowphase.  With it ON (the default) a resting pair keeps the manifold it got up to 2e-5 of pose ago, carried to
    # first order -- NOT what MuJoCo does -- and the comparison is with the oracle's TWIN of that rule (option manifold_keep: same slots,
    # same keep test, same carry, fp64): the bound on the implementation.  What the rule itself changes is bounded oracle against oracle
    # (test_oracle_manifold_keep_rule_against_the_unmodified_oracle).  Robot dofs and satellite dofs are bounded EACH ON THEIR OWN SCALE
    # (VERDICT r5 "weak" 2: against the unmodified oracle the kept manifolds left object-dof errors of 0.1 median / 1.85 max that the
    # whole-vector metric did not see -- the oracle's own manifold of a resting bowl flips between two solutions from step to step).
    for cache, twin in ((0, False), (1, True)):
        be = rc.HipBackend(scene, 8)
        assert be.sim.nsat_max == 16
        be.sim.set_option("manifold_cache", cache)
        rel, events = rc.state_synchronised(be, blob, model, 8, 6, seed=3, twin=twin)
        flags = int(be.sim.info[3].max())
        be.close()
        c = rc.state_synchronised.contacts
        clean, cr, co = rc.state_synchronised.clean, rc.state_synchronised.clean_robot, rc.state_synchronised.clean_obj
        ob = rc.state_synchronised.rel_obj
        print(f"\n[{scene}, manifold cache {cache}] {len(rel)} env-steps: rel qacc p50 {np.percentile(rel, 50):.1e} p99 {np.percentile(rel, 99):.1e} max {rel.max():.1e}; "
              f"satellite dofs on their own scale p50 {np.percentile(ob, 50):.1e} p99 {np.percentile(ob, 99):.1e} max {ob.max():.1e}; on steps with agreeing contact lists: "
              f"robot p99 {np.percentile(cr, 99):.1e}, satellites p99 {np.percentile(co, 99):.1e}; events {len(events)}; contacts {c['n']}, steps with differing pair lists {c['mismatched_steps']}")
        assert flags == 0
        assert len(clean) > 0.9 * len(rel) and np.percentile(cr, 99) < rc.TYPICAL_TOL
        rc.assert_object_dofs(f"[cache {cache}] ")      # (all steps, on identical contacts and on each side's own narrowphase: p99 < 5e-3 on the satellites' own scale)
        assert all(ev["explained"] and ev["flags"] == 0 for ev in events) and len(rc.gross_events(events)) <= 0.005 * len(rel) + 2
        assert c["mismatched_steps"] <= 0.005 * len(rel) + 1 and np.percentile(np.array(c["depth"]), 99) < 5e-5


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["stretch_kitchen4_sat", "stretch_kitchen_robocasa"])
def test_gpu_pgs_satellite_build_state_synchronised(scene):
    """PGS with constraint islands on the device (smj_sat_pgs.h), 4 envs x 200 steps, the oracle's state uploaded before every step,
    MuJoCo's QCQP iteration on both sides: accelerations of every dof against the fp64 PGS oracle.  Bounds as for the dense builds
    under PGS (100 sweeps leave a remainder that carries the sweeps' rounding)."""
    blob, model = _blob(scene)
    for dual in (0, 1):
        be = rc.HipBackend(scene, 4, solver=0)
        be.sim.set_option("qcqp_exact", 1)
        be.sim.set_option("pgs_dual_warmstart", dual)
        assert be.sim.nsat_max == 16
        sweeps = []
        step = be.step
        be.step = lambda n: (step(n), sweeps.append(float(be.sim.info[2].float().mean())))[0]
        _, events = rc.state_synchronised(be, blob, model, 4, 4, seed=3, solver=0, oracle_options={"pgs_dual_warmstart": dual}, twin=True)
        rel, ob, so = rc.state_synchronised.rel_robot, rc.state_synchronised.rel_obj, rc.state_synchronised.same_obj   # (the bounds below: the robot's dofs)
        print(f"   satellite dofs on their own scale: own narrowphase p50 {np.percentile(ob, 50):.1e} p99 {np.percentile(ob, 99):.1e} max {ob.max():.1e}; on the kernel's contact list p50 {np.percentile(so, 50):.1e} p99 {np.percentile(so, 99):.1e} max {so.max():.1e}")
        assert np.percentile(so, 50) < 5e-3 and np.percentile(ob, 99) < 5e-2 and np.percentile(so, 99) < 5e-2   # (as in the emulator's test above; observed on the device: p99 2.1e-3 .. 8.8e-3 own narrowphase, 2.8e-3 .. 2.5e-2 on the kernel's list -- where the robot's island is at the cap and pushes an object)
        flags = int(be.sim.info[3].max())
        be.close()
        c = rc.state_synchronised.contacts
        print(f"\n[{scene}, PGS, dual warm start {dual}] {len(rel)} env-steps: rel qacc p50 {np.percentile(rel, 50):.1e} p90 {np.percentile(rel, 90):.1e} p99 {np.percentile(rel, 99):.1e} "
              f"max {rel.max():.1e}; sweeps per step {np.mean(sweeps):.1f}; events {len(events)}; contacts {c['n']}, steps with differing pair lists {c['mismatched_steps']}")
        assert flags == 0
        # dual = 0 (MuJoCo's warm start on both sides; measured, emulator = device: p50 1.5e-4, p90 1.5e-3, p99 2.1e-2): the tail is PGS's
        # own -- e.g. a gripper finger driven into its stop at 4e4 rad/s^2, where the fp64 oracle's sweeps end on a point the costChange
        # guard will not leave while the fp32 sweeps reach Newton's answer; DESIGN.md section 5.  dual = 1 (the default): both sides
        # converge on most steps and agree there to VERDICT r4 item 2's p99 < 5e-3; where both run into the cap their remainders
        # differ MORE than with MuJoCo's start (there the two sides repeat the same 100 sweeps from the same point): overall p99 5e-2.
        it = rc.state_synchronised.iters
        conv = (it[:, 0] < 100) & (it[:, 1] < 100)
        print(f"   steps on which both sides left the sweeps before the cap of 100: {conv.mean():.2f}" + (f"; there: rel qacc p50 {np.percentile(rel[conv], 50):.1e} p99 {np.percentile(rel[conv], 99):.1e} max {rel[conv].max():.1e}" if conv.any() else ""))
        assert np.percentile(rel, 50) < 5e-4 and np.percentile(rel, 90) < 3e-3 and np.percentile(rel, 99) < 5e-2
        if dual:
            # (measured on the device: kitchen4 51 sweeps per step, both sides below the cap on 77 % of the steps, there p50 1.6e-4 / p90 5e-4 /
            # p99 1.6e-2; Robocasa-scale kitchen 66 sweeps, 53 %, 1.4e-4 / 2e-3 / 3.9e-3.  Leaving the sweeps on MuJoCo's improvement test is not
            # the same as being at the optimum -- a slowly converging island passes it early on either side --, hence a bound on p90, not p99)
            assert np.mean(sweeps) < 75 and conv.mean() > 0.5 and np.percentile(rel[conv], 50) < 3e-4 and np.percentile(rel[conv], 90) < 3e-3
        assert c["mismatched_steps"] <= 0.005 * len(rel) + 1


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["stretch_kitchen4_sat", "stretch_kitchen_robocasa"])
def test_gpu_pgs_two_wavefronts_per_env_agree_with_one(scene):
    """The PGS kernel of the 16-satellite build runs TWO wavefronts per env (smj_kernels_satp.hip): the second one sweeps the
    satellite islands beside the first one's sweeps of the dense system, one workgroup barrier per sweep.  Same algorithm, same
    order of operations as the one-wavefront kernel (option pgs_two_waves = 0) in another translation unit: 256 envs, random
    actions, the one-wavefront sim re-synchronised to the two-wavefront one before every compared step -- same row / contact /
    sweep counts, one-step velocities to fp32 rounding (the compiler contracts multiply-adds differently in the two builds; 100
    sweeps carry that to ~1e-8 relative, 1e-6 on a step whose contact manifold comes from the cache in one sim only)."""
    import torch
    from stretch_mujoco_amd import StretchBatchSimulator

    B = 256
    sims = []
    for two in (1, 0):
        sim = StretchBatchSimulator(num_envs=B, device="cuda:0", scene=scene, solver="pgs")
        sim.start(home=False)
        sim.set_option("pgs_two_waves", two)
        sim.set_option("manifold_cache", 0)   # (sim a has 50 steps of history when sim b starts: their caches hold manifolds built at different poses -- valid either way, 1e-5 apart in the velocities -- and this test is about the sweeps)
        sim.home(settle=False)
        sims.append(sim)
    a, b = sims
    a.step(50)
    b.step(1)   # (the first step after home() applies the pending keyframe: out of the way before states are copied in)
    cr = torch.tensor(np.asarray(a.model["actuator_ctrlrange"]), dtype=torch.float32, device=a.device)
    g = torch.Generator(device=a.device); g.manual_seed(11)
    rel, same_counts = [], []
    for w in range(3):
        a.ctrl[:] = cr[:, :1] + (cr[:, 1:] - cr[:, :1]) * torch.rand(a.nu, B, generator=g, device=a.device)
        b.ctrl[:] = a.ctrl
        for k in range(12):
            for sub in range(4):   # lockstep: the second start of a step (pgs_dual_warmstart) is the previous step's forces -- both sims must have taken that step from the same state
                b.qpos[:] = a.qpos; b.qvel[:] = a.qvel; b.qacc_warmstart[:] = a.qacc_warmstart; b.ctrl[:] = a.ctrl   # (the base controller writes the wheels' ctrl)
                a.step(1); b.step(1)
            torch.cuda.synchronize()
            d = (a.qvel - b.qvel).abs().amax(0) / (1.0 + a.qvel.abs().amax(0))
            rel.append(d.cpu().numpy())
            same_counts.append(float(((a.info[:2] == b.info[:2]).all(0) & ((a.info[2] - b.info[2]).abs() <= 2)).float().mean()))
    rel = np.concatenate(rel)
    print(f"\n[{scene}] two wavefronts vs one, {len(rel)} env-steps: rel dqvel p50 {np.percentile(rel, 50):.1e} p99 {np.percentile(rel, 99):.1e} max {rel.max():.1e}; "
          f"same nefc / ncon and sweeps within 2 on {100 * np.mean(same_counts):.2f} % of the env-steps")
    assert bool(torch.isfinite(a.qpos).all())
    # (round 5, with the second start the sweeps END before the cap on most steps, and the two builds' rounding decides on which sweep:
    # same row / contact counts and sweep counts within 2 on > 90 % of the env-steps; observed p50 2.7e-9, p99 3e-4, max 0.16 -- a step on
    # which one build stopped a few sweeps before the other on a slowly converging island)
    # (round 6, exact face normals from the narrowphase: p99 1.2e-3 in the Robocasa-scale kitchen -- more steps on which the two builds leave the
    # sweeps a few sweeps apart; the bound is on two roundings of an unconverged iteration, not on an error)
    assert np.percentile(rel, 50) < 1e-6 and np.percentile(rel, 99) < 5e-3 and rel.max() < 0.5 and np.mean(same_counts) > 0.9
    for sim in sims:
        sim.stop()


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["stretch_kitchen4_sat", "stretch_kitchen_robocasa"])
def test_gpu_newton_two_wavefronts_per_env_equal_one_bit_for_bit(scene):
    """The Newton kernel of the 16-satellite build runs TWO wavefronts per env (smj_kernels_sat2.hip): in the collision stage the
    second one works the moving-moving pairs while the first one works the pairs with the static world, and the first one appends
    the second one's contacts behind its own -- the one-wavefront kernel's list (option newton_two_waves = 0), contact for contact.
    So the two kernels must agree BIT FOR BIT, free-running, manifold cache and all: 256 envs, 300 steps under random actions."""
    import torch
    from stretch_mujoco_amd import StretchBatchSimulator

    B = 256
    sims = []
    for two in (1, 0):
        sim = StretchBatchSimulator(num_envs=B, device="cuda:0", scene=scene)
        sim.start(home=False)
        sim.set_option("newton_two_waves", two)
        sim.home(settle=False)
        sims.append(sim)
    a, b = sims
    cr = torch.tensor(np.asarray(a.model["actuator_ctrlrange"]), dtype=torch.float32, device=a.device)
    g = torch.Generator(device=a.device); g.manual_seed(5)
    ncon = []
    for w in range(6):
        if w:
            a.ctrl[:] = cr[:, :1] + (cr[:, 1:] - cr[:, :1]) * torch.rand(a.nu, B, generator=g, device=a.device)
            b.ctrl[:] = a.ctrl
        a.step(50); b.step(50)
        torch.cuda.synchronize()
        assert torch.equal(a.qpos, b.qpos) and torch.equal(a.qvel, b.qvel) and torch.equal(a.qacc_warmstart, b.qacc_warmstart), f"launch {w}"
        assert torch.equal(a.info[:3], b.info[:3])   # rows, contacts, solver iterations of the last step
        ncon.append(float(a.info[1].float().mean()))
    print(f"\n[{scene}] two wavefronts vs one under Newton: identical states over 300 steps x {B} envs; contacts per env {np.mean(ncon):.1f}")
    assert bool(torch.isfinite(a.qpos).all()) and np.mean(ncon) > 4
    for sim in sims:
        sim.stop()


@pytest.mark.gpu
@pytest.mark.parametrize("escalate", [1, 0])
def test_gpu_newton_two_wavefronts_at_scale_with_contact_overflow(escalate):
    """The same comparison where contact lists OVERFLOW: 4096 kitchens at Robocasa scale under random actions park a few envs per
    launch (more than 56 contacts: the step is redone by the 32-satellite build; with option escalate = 0 it is flagged and goes on
    with the contacts it has).  The two wavefronts claim their contact slots from one counter; a step whose claims do not fit must
    leave a list without holes (the first version did not: stale pair indices reached the row builder -- a memory fault, found at
    this scale only).  Every env that neither kernel flagged has the same state bit for bit; the flagged ones stay finite."""
    import torch
    from stretch_mujoco_amd import StretchBatchSimulator

    B = 4096
    sims = []
    for two in (1, 0):
        sim = StretchBatchSimulator(num_envs=B, device="cuda:0", scene="stretch_kitchen_robocasa")
        sim.start(home=False)
        sim.set_option("newton_two_waves", two)
        sim.set_option("escalate", escalate)
        sim.set_option("pollers", 0)   # (parked chunks go to the sweep after the launch: the order the large build works them in does not depend on timing)
        sim.home(settle=False)
        sims.append(sim)
    a, b = sims
    cr = torch.tensor(np.asarray(a.model["actuator_ctrlrange"]), dtype=torch.float32, device=a.device)
    g = torch.Generator(device=a.device); g.manual_seed(99)
    a.step(300); b.step(300)
    flagged = torch.zeros(B, dtype=torch.bool, device=a.device)
    for w in range(5):
        a.ctrl[:] = cr[:, :1] + (cr[:, 1:] - cr[:, :1]) * torch.rand(a.nu, B, generator=g, device=a.device)
        b.ctrl[:] = a.ctrl
        a.step(50); b.step(50)
        torch.cuda.synchronize()
        flagged |= a.info[3].ne(0) | b.info[3].ne(0)
    clean = ~flagged
    same = (a.qpos == b.qpos).all(0) & (a.qvel == b.qvel).all(0)
    print(f"\n[escalate {escalate}] 4096 kitchens x 550 steps: {int(flagged.sum())} envs flagged at some launch; of the {int(clean.sum())} others {int((same & clean).sum())} identical")
    assert bool(torch.isfinite(a.qpos).all()) and bool(torch.isfinite(b.qpos).all())
    assert int(clean.sum()) > 0.9 * B
    # (escalate = 1: a parked step is redone from the same state by the same large build in both sims -- those envs are not flagged and
    # are among the compared ones)
    assert bool(same[clean].all()), int((~same & clean).sum())
    # Round 6: a step whose contact list overflows is redone by one wavefront in table order in BOTH builds (run(), smj_step_impl.h), so the
    # flagged envs stay equal too -- the truncated list no longer depends on the two wavefronts' timing.  With escalate = 0
    # nothing is handed over and every env must agree; with escalate = 1 the envs flagged for contacts (more than 64: nobody to hand to).
    nfl = int(flagged.sum())
    print(f"   flagged envs with identical states in the two builds: {int((same & flagged).sum())} of {nfl}")
    for e in torch.nonzero(flagged & ~same).flatten().tolist()[:8]:
        print(f"      env {e}: flags two / one wavefront {hex(int(a.info[3, e]))} / {hex(int(b.info[3, e]))}, max |dqpos| {float((a.qpos[:, e] - b.qpos[:, e]).abs().max()):.1e}, contacts {int(a.info[1, e])} / {int(b.info[1, e])}")
    # (an env that ran out of ROWS / dense rows under escalate = 0 goes on with a truncated constraint set, which the two builds do not
    # truncate bit for bit alike -- never claimed; the claim is about the contact list)
    only_contacts = flagged & ((a.info[3] | b.info[3]) & ~(0x4000 | 2)).eq(0)
    print(f"   of them flagged for contacts only: {int(only_contacts.sum())}, identical: {int((same & only_contacts).sum())}")
    assert bool(same[only_contacts].all()), int((~same & only_contacts).sum())
    for sim in sims:
        sim.stop()


@pytest.mark.gpu
def test_gpu_kitchen_soak_keeps_every_robot_on_the_floor_and_in_the_room():
    """The regression guard of round 6's narrowphase fixes: 4096 kitchens at Robocasa scale x 4000 steps of
    heterogeneous random actions.  With the fp32 contact normal of rounds 1-5 (a 2e-3 rad error on every resting contact) a soak of
    this kind threw robots up to 3.9 m into the air and 10 m out of the 5 m room; now no base rises above 0.35 m, every robot stays
    within 1.5 m of the origin in x and y, states are finite, there is no bad-state reset, and at most 1.5 % of the envs carry the
    contact-capacity flag (64 contacts: a lane count)."""
    import torch
    from stretch_mujoco_amd import StretchBatchSimulator

    B = 4096
    sim = StretchBatchSimulator(num_envs=B, device="cuda:0", scene="stretch_kitchen_robocasa")
    sim.start(home=True)
    cr = torch.tensor(np.asarray(sim.model["actuator_ctrlrange"]), dtype=torch.float32, device=sim.device)
    g = torch.Generator(device=sim.device); g.manual_seed(2024)
    zmax = xymax = 0.0
    for _ in range(80):
        sim.ctrl[:] = cr[:, :1] + (cr[:, 1:] - cr[:, :1]) * torch.rand(sim.nu, B, generator=g, device=sim.device)
        sim.step(50)
        zmax = max(zmax, float(sim.qpos[2].max()))
        xymax = max(xymax, float(sim.qpos[0:2].abs().max()))
    torch.cuda.synchronize()
    fl = sim.info[3]
    print(f"\n4096 kitchens x 4000 steps: highest base {zmax:.3f} m, |x|,|y| max {xymax:.2f} m, contact flag on {float(((fl & 2) != 0).float().mean()):.4f} of the envs, "
          f"bad-state resets {int(((fl & 4) != 0).sum())}, steps {int(sim.nstep.min())}..{int(sim.nstep.max())}")
    assert bool(torch.isfinite(sim.qpos).all()) and bool(torch.isfinite(sim.qvel).all())
    assert int(sim.nstep.min()) == int(sim.nstep.max())
    assert zmax < 0.35 and xymax < 1.5
    assert int(((fl & 4) != 0).sum()) == 0 and float(((fl & 2) != 0).float().mean()) < 0.015
    sim.stop()


@pytest.mark.gpu
def test_gpu_pgs_kitchen_at_robocasa_scale_steps_every_env():
    """PGS, 1024 envs of the kitchen at Robocasa scale, 200 steps of random actions: every env steps, dense systems beyond the
    16-satellite build's 96 rows go to the 32-satellite build (160), states stay finite, at most 1 % of the envs carry a flag."""
    import torch
    from stretch_mujoco_amd import StretchBatchSimulator

    B = 1024
    sim = StretchBatchSimulator(num_envs=B, device="cuda:0", scene="stretch_kitchen_robocasa", solver="pgs")
    sim.start(home=False)
    sim.home(settle=False)
    sim.step(100)
    cr = torch.tensor(np.asarray(sim.model["actuator_ctrlrange"]), dtype=torch.float32, device=sim.device)
    g = torch.Generator(device=sim.device); g.manual_seed(5)
    for _ in range(4):
        sim.ctrl[:] = cr[:, :1] + (cr[:, 1:] - cr[:, :1]) * torch.rand(sim.nu, B, generator=g, device=sim.device)
        sim.step(50)
    torch.cuda.synchronize()
    fl = sim.info[3]
    assert int(sim.nstep.min()) == int(sim.nstep.max()) == 300
    assert bool(torch.isfinite(sim.qpos).all()) and bool(torch.isfinite(sim.qvel).all())
    assert float((fl != 0).float().mean()) <= 0.01, (int((fl != 0).sum()), hex(int(fl.max())))
    sim.stop()


@pytest.mark.gpu
def test_gpu_config4_as_worded_4096_kitchens_under_pgs():
    """BASELINE.json config 4 at its full size and with the solver its wording names: 4096 kitchens at Robocasa scale under PGS, 150
    steps of heterogeneous random actions.  Size-independent properties: every env steps exactly as often as asked; states finite;
    unit quaternions of the base and of all 8 free objects; the equality constraints of the arm hold; at most 1 % of the envs carry a
    capacity flag; and the envs are independent -- the first 256 envs of the batch, run again ALONE with the same inputs, end in the
    same states bit for bit in >= 95 % of the unflagged envs (the rest: envs handed to the 32-satellite build for the rest of a chunk)."""
    import torch
    from stretch_mujoco_amd import StretchBatchSimulator

    B, sub = 4096, 256
    res = []
    for nb in (B, sub):
        sim = StretchBatchSimulator(num_envs=nb, device="cuda:0", scene="stretch_kitchen_robocasa", solver="pgs")
        sim.start(home=False)
        sim.home(settle=False)
        sim.step(50)
        cr = torch.tensor(np.asarray(sim.model["actuator_ctrlrange"]), dtype=torch.float32, device=sim.device)
        g = torch.Generator(device=sim.device); g.manual_seed(11)
        for _ in range(2):
            u = torch.rand(sim.nu, B, generator=g, device=sim.device)[:, :nb]   # the sub-batch gets the first columns of the same draws
            sim.ctrl[:] = cr[:, :1] + (cr[:, 1:] - cr[:, :1]) * u
            sim.step(50)
        torch.cuda.synchronize()
        assert int(sim.nstep.min()) == int(sim.nstep.max()) == 150
        res.append((sim.qpos.clone(), sim.qvel.clone(), sim.info[3].clone(), sim.info[2].clone()))
        model = sim.model
        sim.stop()
    (q, v, fl, it), (qs, vs, fls, _) = res
    assert bool(torch.isfinite(q).all()) and bool(torch.isfinite(v).all())
    assert float((fl != 0).float().mean()) <= 0.01, (int((fl != 0).sum()), hex(int(fl.max())))
    ok = fl == 0
    assert float((q[3:7, ok].norm(dim=0) - 1).abs().max()) < 1e-5
    qa = np.asarray(model["jnt_qposadr"]); jt = np.asarray(model["jnt_type"])
    for j in np.nonzero(jt == 0)[0][1:]:      # the free objects' quaternions
        assert float((q[qa[j] + 3: qa[j] + 7, ok].norm(dim=0) - 1).abs().max()) < 1e-5, j
    assert float((q[10:14, ok] - q[10:11, ok]).abs().max()) < 3e-2      # telescoping arm segments equal (soft equality)
    both = ok[:sub] & (fls == 0)
    same = (q[:, :sub] == qs).all(0) & (v[:, :sub] == vs).all(0)
    print(f"\n4096 kitchens under PGS, 150 steps: {int((~ok).sum())} envs flagged, sweeps of the last step mean {float(it.float().mean()):.1f}; "
          f"first {sub} envs run alone: {int((same & both).sum())} of {int(both.sum())} unflagged envs identical")
    # (bit for bit for the envs whose steps all ran on the 16-satellite build; an env that was handed to the 32-satellite build finishes its
    # CHUNK there, chunk boundaries move with the batch size, and the two builds' unconverged sweeps differ in the last bits -- observed 3 of 255)
    dq = float((q[:, :sub] - qs).abs().amax(0)[both & ~same].max()) if bool((both & ~same).any()) else 0.0
    print(f"   largest |dqpos| among the {int((both & ~same).sum())} envs that differ: {dq:.1e}")
    assert int(both.sum()) > 0.9 * sub and int((same & both).sum()) >= 0.95 * int(both.sum())


@pytest.mark.gpu
def test_gpu_lidar_in_the_kitchen_at_robocasa_scale():
    """Config 3's readout in config 4's scene: the 360-ray lidar among 300 fixture geoms (boxes, cylinders, convex mesh pieces),
    doors and drawers at their settled angles, objects where they came to rest -- against the oracle's ray caster on the state the
    device reached after 60 steps (1e-3 m, at most 3 rays at silhouettes), IMU alongside."""
    import torch
    from oracle.oracle import Oracle
    from stretch_mujoco_amd import StretchBatchSimulator, StretchSensors

    sim = StretchBatchSimulator(num_envs=2, device="cuda:0", scene="stretch_kitchen_robocasa", sensors_to_use=StretchSensors.all())
    sim.start(home=False)
    sim.step(60)
    q = sim.qpos[:, 0].cpu().numpy().astype(np.float64)
    v = sim.qvel[:, 0].cpu().numpy().astype(np.float64)
    sim.step(1)
    torch.cuda.synchronize()
    L = sim.pull_sensor_data().lidar.cpu().numpy()[0]
    o = Oracle(sim._blob)
    o.arr("qpos")[:] = q; o.arr("qvel")[:] = v
    o.forward(); o.sensors(True)
    ref = o.arr("lidar")
    bad = np.abs(L - ref) > 1e-3
    print(f"\nkitchen lidar: {int((ref > 0).sum())} of 360 rays return, {int(bad.sum())} differ, nearest {ref[ref > 0].min():.3f} m")
    assert bad.sum() <= 3, (int(bad.sum()), L[bad], ref[bad])
    assert (ref > 0).sum() > 200 and ref[ref > 0].min() < 1.5   # fixtures all around the robot
    sim.stop()


@pytest.mark.gpu
def test_gpu_kitchen_at_robocasa_scale_runs_the_bench_workload_without_flags():
    """4096 envs of the generated kitchen, 600 steps of full-range random actions: every env steps every step; rows, dense rows,
    coupled satellites beyond the 16-satellite build go to the 32-satellite one (320 rows) and are NOT flagged.  What can still be
    flagged is the one capacity that is a lane count -- more than 64 contacts in one env (the robot driven into a shelf full of
    objects) -- and the bad-state reset: at most 0.2 % of the envs (measured: 3 of 4096).  Quaternions stay normalised, the
    free objects nobody touched are where they were put."""
    import torch
    from stretch_mujoco_amd import StretchBatchSimulator

    B = 4096
    sim = StretchBatchSimulator(num_envs=B, device="cuda:0", scene="stretch_kitchen_robocasa")
    sim.start(home=False)
    sim.home(settle=False)
    sim.step(300)
    q0 = sim.qpos.clone()
    cr = torch.tensor(np.asarray(sim.model["actuator_ctrlrange"]), dtype=torch.float32, device=sim.device)
    g = torch.Generator(device=sim.device); g.manual_seed(5)
    for _ in range(12):
        sim.ctrl[:] = cr[:, :1] + (cr[:, 1:] - cr[:, :1]) * torch.rand(sim.nu, B, generator=g, device=sim.device)
        sim.step(50)
    torch.cuda.synchronize()
    fl = sim.info[3]
    # rows, dense rows, coupled satellites, pools, pipeline: never.  Row items of one satellite (0x100) and the broadphase candidate
    # list (0x800) have the same size in the 32-satellite build's hand-over path for the list and a 1-in-5000 hit rate without it
    # (profiles/r04_sat_caps.txt): at most two envs of 4096 (which envs get there depends on the order the pollers take parked chunks in)
    assert int(((fl & (0x200 | 0x400 | 0x1000 | 0x2000 | 8)) != 0).sum()) == 0, hex(int(fl.max()))
    assert int(((fl & (0x100 | 0x800)) != 0).sum()) <= 2, int(((fl & (0x100 | 0x800)) != 0).sum())
    assert float((fl != 0).float().mean()) <= 0.002, int((fl != 0).sum())
    assert int(sim.nstep.min()) == int(sim.nstep.max()) == 900
    q = sim.qpos
    assert bool(torch.isfinite(q).all())
    si = np.asarray(sim.model["k_sat_i"]).reshape(16, -1)
    for s in si[si[:, 4] == 6]:
        qa = int(s[2])
        assert float((q[qa + 3:qa + 7].norm(dim=0) - 1).abs().max()) < 1e-5
    sponge = int(si[-1, 2])   # the sponge on the table: out of the robot's reach
    assert float((q[sponge:sponge + 3] - q0[sponge:sponge + 3]).abs().max()) < 2e-3
    sim.stop()


@pytest.mark.gpu
def test_gpu_satellite_build_hands_over_to_the_large_build():
    """`primary_rows` lowered to 100: the settled kitchen (138 rows) cannot stay in the 16-satellite build, every step is parked and
    finished by the 32-satellite build -- same states as the oracle."""
    import torch
    from oracle.oracle import Oracle
    from stretch_mujoco_amd import StretchBatchSimulator

    blob, _ = _blob("stretch_kitchen_robocasa")
    o = Oracle(blob); o.set_option("solver", 2)
    o.arr("ctrl")[:10] = [0, 0, 0.6, 0.1, 0, 0, 0, 0, 0, 0]
    o.step(400)
    sim = StretchBatchSimulator(num_envs=4, device="cuda:0", scene="stretch_kitchen_robocasa")
    sim.start(home=False)
    sim.set_option("primary_rows", 100)
    sim.ctrl[:] = torch.tensor(o.arr("ctrl")[:10], dtype=torch.float32, device=sim.device).unsqueeze(1)
    for name, t in (("qpos", sim.qpos), ("qvel", sim.qvel), ("qacc_warmstart", sim.qacc_warmstart)):
        t[:] = torch.tensor(o.arr(name), dtype=torch.float32, device=sim.device).unsqueeze(1)
    sim.step(40)
    o.step(40)
    torch.cuda.synchronize()
    assert int(sim.info[3].max()) == 0 and int(sim.info[0, 0]) > 100
    assert float(np.abs(sim.qpos[:, 0].cpu().numpy() - o.arr("qpos")).max()) < 1e-4
    sim.stop()
