"""Degenerate resting configurations of primitive pairs -- yaw angles that are multiples of 45 / 90 / 180 degrees, equal sizes,
edges and corners exactly in line: where fp32 and fp64 take different branches of a narrowphase if they can.  The step kernel's
source (lane emulator) against the fp64 oracle, one step from rest, 1 mm of penetration.  (Round 4: this sweep found the bogus
edge-edge point of box-box on boxes lying flat, the two-point manifold of equal boxes stacked in line, and the manifold size that
depended on ties in the cap's selection rule.)  TEST INFRASTRUCTURE (imports oracle/)."""
import math

import numpy as np
import pytest

from oracle.oracle import Oracle
from stretch_mujoco_amd import mjcf_compiler as C
from stretch_mujoco_amd import model_blob as B
from stretch_mujoco_amd import model_fuse as F

OPT = '<option integrator="implicitfast" cone="elliptic" impratio="20"/>'
SHAPES = {"box": ("box", ".2 .15 .1", 0.1), "sphere": ("sphere", ".1", 0.1), "capsule": ("capsule", ".06 .12", 0.18),
          "cylinder": ("cylinder", ".08 .1", 0.1), "ellipsoid": ("ellipsoid", ".12 .08 .1", 0.1)}
YAWS = [0, math.pi / 4, math.pi / 2, math.pi, -math.pi / 2, 0.3, 0.1, 1.0, 2.0]


def _run(a, b, ya, yb, off, static):
    from emul.emul import Emul
    from stretch_mujoco_amd.lib import debug_layout

    ta, sa, ha = SHAPES[a]; tb, sb, hb = SHAPES[b]
    za = 0.3
    zb = za + ha + hb - 0.001
    A = (f'<geom type="{ta}" size="{sa}" pos="0 0 {za}" euler="0 0 {ya}"/>' if static else
         f'<body pos="0 0 {za}" euler="0 0 {ya}"><freejoint/><geom type="{ta}" size="{sa}" mass="1"/></body>')
    scene = ('<mujoco><compiler angle="radian"/>' + OPT + '<worldbody>' + A +
             f'<body pos="{off[0]} {off[1]} {zb}" euler="0 0 {yb}"><freejoint/><geom type="{tb}" size="{sb}" mass="1"/></body></worldbody></mujoco>')
    blob = B.dumps(F.prepare_for_kernels(C.compile_string(scene)))
    o = Oracle(blob); o.set_option("solver", 2)
    e = Emul(blob, dict(nq=o.dim("nq"), nv=o.dim("nv"), nu=0, nlidar=0), num_envs=1, variant="standard", debug=True); e.set_option("solver", 2)
    D = debug_layout(e.nvp, e.ncon_max, 0)
    e.qpos[:, 0] = o.arr("qpos")
    o.step(1); e.step(1)
    nk = int(e.info[1, 0])
    dk = np.sort(e.debug[D["con"]:D["con"] + 8 * nk, 0].reshape(nk, 8)[:, 0])
    do = np.sort(o.arr("contact").reshape(o.ncon, -1)[:, 0]) if o.ncon else np.zeros(0)
    same = nk == o.ncon and (nk == 0 or np.abs(dk - do).max() < 2e-5)
    return same, nk, o.ncon, int(e.info[3, 0])


def test_box_on_box_resting_configurations_agree_exactly():
    """Every one of 400 box-on-box configurations: same number of contacts, same depths, no flag; a face contact never comes out
    as a single point (the round-4 bug) and boxes in line get their four corners."""
    rng = np.random.default_rng(7)
    for trial in range(400):
        ya, yb = rng.choice(YAWS), rng.choice(YAWS)
        off = [rng.choice([0, 0.05, -0.1, 0.13]), rng.choice([0, 0.05, -0.08]), 0]
        same, nk, no, fl = _run("box", "box", ya, yb, off, rng.random() < 0.5)
        assert same and fl == 0 and nk >= 3, (trial, ya, yb, off, nk, no)


def test_primitive_pairs_resting_configurations_mostly_agree():
    """All pairs of box / sphere / capsule / cylinder / ellipsoid: at least 95 % of 400 configurations agree exactly.  The rest are
    ties of multiccd's duplicate test (a new point 1e-3 x the smaller bounding radius from an earlier one, exactly on the
    threshold for a cylinder standing on a capsule's axis): one point more or less of the same manifold, in fp64 and fp32 alike."""
    rng = np.random.default_rng(11)
    names = list(SHAPES)
    agree = 0
    for trial in range(400):
        a, b = rng.choice(names), rng.choice(names)
        ya, yb = rng.choice(YAWS), rng.choice(YAWS)
        off = [rng.choice([0, 0.05, -0.1, 0.13]), rng.choice([0, 0.05, -0.08]), 0]
        same, nk, no, fl = _run(a, b, ya, yb, off, rng.random() < 0.5)
        assert fl == 0 and abs(nk - no) <= 2, (trial, a, b, ya, yb, off, nk, no)
        agree += same
    assert agree >= 0.95 * 400, agree


def _rest_height(a, e):
    """Distance from the centre of shape `a` at xyz Euler angles e to its lowest point."""
    cx, cy, cz = np.cos(e); sx, sy, sz = np.sin(e)
    R2 = np.array([sx * sz - cx * sy * cz, sx * cz + cx * sy * sz, cx * cy])   # third row of the rotation
    if a == "box":
        return float(np.abs(R2) @ np.array([.2, .15, .1]))
    if a == "sphere":
        return 0.1
    if a == "capsule":
        return 0.06 + abs(R2[2]) * 0.12
    if a == "cylinder":
        return abs(R2[2]) * 0.1 + math.sqrt(max(0.0, 1 - R2[2] ** 2)) * 0.08
    return math.sqrt((R2[0] * .12) ** 2 + (R2[1] * .08) ** 2 + (R2[2] * .1) ** 2)


def _rest_on(support, a, e, yawb=0.0):
    from emul.emul import Emul
    from stretch_mujoco_amd.lib import debug_layout

    ta, sa, _ = SHAPES[a]
    h = _rest_height(a, e)
    ground = ('<geom type="plane" size="0 0 1"/>' if support == "plane" else f'<geom type="box" size=".5 .4 .2" pos="0 0 -0.2" euler="0 0 {yawb}"/>')
    scene = ('<mujoco><compiler angle="radian"/>' + OPT + '<worldbody>' + ground +
             f'<body pos="0.1 -0.2 {h - 0.001}" euler="{e[0]} {e[1]} {e[2]}"><freejoint/><geom type="{ta}" size="{sa}" mass="1"/></body></worldbody></mujoco>')
    blob = B.dumps(F.prepare_for_kernels(C.compile_string(scene)))
    o = Oracle(blob); o.set_option("solver", 2)
    em = Emul(blob, dict(nq=o.dim("nq"), nv=o.dim("nv"), nu=0, nlidar=0), num_envs=1, variant="standard", debug=True); em.set_option("solver", 2)
    D = debug_layout(em.nvp, em.ncon_max, 0)
    em.qpos[:, 0] = o.arr("qpos")
    o.step(1); em.step(1)
    nk = int(em.info[1, 0])
    dk = np.sort(em.debug[D["con"]:D["con"] + 8 * nk, 0].reshape(nk, 8)[:, 0])
    do = np.sort(o.arr("contact").reshape(o.ncon, -1)[:, 0]) if o.ncon else np.zeros(0)
    return nk == o.ncon and (nk == 0 or np.abs(dk - do).max() < 2e-5), nk, o.ncon


def test_primitives_resting_on_the_plane_at_right_angles():
    """MuJoCo's plane rules at orientations made of quarter turns (and 45 degrees, 0.3 rad): every one of 300 configurations agrees.
    (Round 4: a cylinder STANDING on the plane after a yaw lost its contacts -- the "downhill" direction of mjc_PlaneCylinder was
    1e-7 of rounding, normalised; the upright branch now takes over below a tilt of 3e-5 rad.)"""
    rng = np.random.default_rng(3)
    ang = [0, math.pi / 2, math.pi, -math.pi / 2, math.pi / 4, 0.3]
    for trial in range(300):
        a = rng.choice(list(SHAPES))
        e = [rng.choice(ang), rng.choice(ang), rng.choice(ang)] if rng.random() < 0.6 else [0, 0, rng.choice(ang)]
        same, nk, no = _rest_on("plane", a, e)
        assert same and nk >= 1, (trial, a, e, nk, no)


def test_primitives_resting_on_a_box_at_right_angles():
    """The same bodies rolled by quarter turns onto a large box (closed forms, box-box, MPR + multiccd): everything but a cylinder
    lying on its side agrees exactly; that one is a LINE contact, where MPR's first point is anywhere on the line and the four
    counter-rotated queries of multiccd land 0.1-0.4 mm apart in depth in fp32 and fp64 -- same manifold, other samples of it."""
    rng = np.random.default_rng(5)
    ang = [0, math.pi / 2, math.pi, -math.pi / 2]
    for trial in range(300):
        a = rng.choice(list(SHAPES))
        e = [rng.choice(ang), rng.choice(ang), rng.choice(ang)] if rng.random() < 0.6 else [0, 0, rng.choice(ang)]
        same, nk, no = _rest_on("box", a, e, rng.choice([0, math.pi / 4, math.pi / 2, 0.3]))
        lying_cylinder = a == "cylinder" and abs(math.cos(e[0]) * math.cos(e[1])) < 0.5
        assert nk >= 1 and (same or (lying_cylinder and abs(nk - no) <= 1)), (trial, a, e, nk, no)


def test_tower_of_equal_boxes_stands():
    """Three equal boxes stacked exactly in line on the plane (the configuration that came out with two diagonal contacts per
    face before the candidates on coincident edges were handled): 1500 steps, in the oracle and in the kernel's source, free running
    -- the tower stands (top box within 0.2 mm of where it settles, upright to 1e-4), four contacts per face, and the two agree."""
    from emul.emul import Emul

    scene = ('<mujoco><compiler angle="radian"/>' + OPT + '<option timestep="0.002"/><worldbody><geom type="plane" size="0 0 1"/>' +
             "".join(f'<body pos="0 0 {0.1 + 0.2 * k - 0.0005 * (k + 1)}"><freejoint/><geom type="box" size=".2 .15 .1" mass="{2 - 0.5 * k}"/></body>' for k in range(3)) +
             '</worldbody></mujoco>')
    blob = B.dumps(F.prepare_for_kernels(C.compile_string(scene)))
    o = Oracle(blob); o.set_option("solver", 2)
    e = Emul(blob, dict(nq=o.dim("nq"), nv=o.dim("nv"), nu=0, nlidar=0), num_envs=1, variant="standard"); e.set_option("solver", 2)
    e.qpos[:, 0] = o.arr("qpos")
    o.step(300); e.step(300)
    z300 = o.arr("qpos")[16]
    assert o.ncon == 12 and int(e.info[1, 0]) == 12, (o.ncon, int(e.info[1, 0]))
    o.step(1200); e.step(1200)
    q = o.arr("qpos")
    assert abs(q[16] - z300) < 2e-4 and abs(q[14]) < 1e-3 and abs(q[15]) < 1e-3          # top box: height, x, y
    assert all(abs(abs(q[7 * k + 3]) - 1) < 1e-8 for k in range(3))                      # every quaternion still the identity: upright, no yaw creep
    assert np.abs(e.qpos[:, 0] - q).max() < 1e-4 and int(e.info[3, 0]) == 0
    assert np.abs(o.arr("qvel")).max() < 1e-3 and np.abs(e.qvel[:, 0]).max() < 1e-3


# ------------------------------------------------------------------------------------------------------------------- GPU
def _device_contacts(scene_xml):
    """One step of the scene on the device (libsmj.so through StretchBatchSimulator on a robot-less blob) and in the oracle:
    (same, device count, oracle count)."""
    import torch
    from stretch_mujoco_amd import StretchBatchSimulator

    blob = B.dumps(F.prepare_for_kernels(C.compile_string(scene_xml)))
    o = Oracle(blob); o.set_option("solver", 2)
    sim = StretchBatchSimulator(num_envs=1, device="cuda:0", model_blob_bytes=blob, debug=True)
    sim.start(home=False)
    sim.qpos[:, 0] = torch.tensor(o.arr("qpos"), dtype=torch.float32, device=sim.device)
    o.step(1); sim.step(1)
    torch.cuda.synchronize()
    D = sim.debug_layout
    nk = int(sim.info[1, 0])
    dk = np.sort(sim.debug[D["con"]:D["con"] + 8 * nk, 0].cpu().numpy().reshape(nk, 8)[:, 0])
    do = np.sort(o.arr("contact").reshape(o.ncon, -1)[:, 0]) if o.ncon else np.zeros(0)
    fl = int(sim.info[3, 0])
    sim.stop()
    return nk == o.ncon and (nk == 0 or np.abs(dk - do).max() < 2e-5), nk, o.ncon, fl


@pytest.mark.gpu
def test_gpu_degenerate_resting_configurations():
    """The sweeps above through the compiled kernels (v_rcp / v_sqrt sequences instead of IEEE divides, fused multiply-adds where the
    compiler chose them): 120 box-on-box pairs, 100 bodies on the plane and 100 on a box top, every one as in the emulator runs --
    box pairs and plane contacts exact, a cylinder lying on a box within one point."""
    rng = np.random.default_rng(17)
    for trial in range(120):
        ya, yb = rng.choice(YAWS), rng.choice(YAWS)
        off = [rng.choice([0, 0.05, -0.1, 0.13]), rng.choice([0, 0.05, -0.08]), 0]
        static = rng.random() < 0.5
        A = (f'<geom type="box" size=".2 .15 .1" pos="0 0 0.3" euler="0 0 {ya}"/>' if static else
             f'<body pos="0 0 0.3" euler="0 0 {ya}"><freejoint/><geom type="box" size=".2 .15 .1" mass="1"/></body>')
        scene = ('<mujoco><compiler angle="radian"/>' + OPT + '<worldbody>' + A +
                 f'<body pos="{off[0]} {off[1]} 0.499" euler="0 0 {yb}"><freejoint/><geom type="box" size=".2 .15 .1" mass="1"/></body></worldbody></mujoco>')
        same, nk, no, fl = _device_contacts(scene)
        assert same and fl == 0 and nk >= 3, ("box-box", trial, ya, yb, off, static, nk, no)
    ang = [0, math.pi / 2, math.pi, -math.pi / 2]
    for support in ("plane", "box"):
        for trial in range(100):
            a = rng.choice(list(SHAPES))
            e = [rng.choice(ang), rng.choice(ang), rng.choice(ang)] if rng.random() < 0.6 else [0, 0, rng.choice(ang + [math.pi / 4, 0.3])]
            yawb = rng.choice([0, math.pi / 4, math.pi / 2, 0.3])
            ta, sa, _ = SHAPES[a]
            h = _rest_height(a, e)
            ground = ('<geom type="plane" size="0 0 1"/>' if support == "plane" else f'<geom type="box" size=".5 .4 .2" pos="0 0 -0.2" euler="0 0 {yawb}"/>')
            scene = ('<mujoco><compiler angle="radian"/>' + OPT + '<worldbody>' + ground +
                     f'<body pos="0.1 -0.2 {h - 0.001}" euler="{e[0]} {e[1]} {e[2]}"><freejoint/><geom type="{ta}" size="{sa}" mass="1"/></body></worldbody></mujoco>')
            same, nk, no, fl = _device_contacts(scene)
            lying_cylinder = support == "box" and a == "cylinder" and abs(math.cos(e[0]) * math.cos(e[1])) < 0.5
            assert fl == 0 and nk >= 1 and (same or (lying_cylinder and abs(nk - no) <= 1)), (support, trial, a, e, nk, no)


@pytest.mark.gpu
@pytest.mark.parametrize("nbodies,solver", [(5, "newton"), (8, "newton"), (5, "pgs")])
def test_gpu_piles_of_primitives_state_synchronised(nbodies, solver):
    """Six random piles -- five (30 dofs: the standard kernel) or eight (48 dofs: the 50-column build) bodies of mixed shapes dropped
    on a box and the plane, orientations from the quarter-turn set -- 200 steps each on the device with the oracle's state uploaded
    before every step: contact / row counts equal on >= 99 % of the steps (the rest: multiccd ties, one point more or less), one-step
    velocities p90 < 1e-4 relative on the steps that agree (observed 1198 of 1200, p90 9e-6), no flag."""
    import torch
    from stretch_mujoco_amd import StretchBatchSimulator

    rng = np.random.default_rng(23)
    ang = [0, math.pi / 2, math.pi / 4, 0.3, -math.pi / 2]
    agree, total, errs = 0, 0, []
    for pile in range(6):
        bodies = ""
        for k in range(nbodies):
            a = rng.choice(list(SHAPES)); ta, sa, _ = SHAPES[a]
            e = [rng.choice(ang), rng.choice(ang), rng.choice(ang)]
            bodies += (f'<body pos="{rng.uniform(-0.25, 0.25):.3f} {rng.uniform(-0.2, 0.2):.3f} {0.35 + 0.28 * k:.3f}" euler="{e[0]} {e[1]} {e[2]}"><freejoint/>'
                       f'<geom type="{ta}" size="{sa}" mass="{rng.uniform(0.2, 1.5):.2f}"/></body>')
        scene = ('<mujoco><compiler angle="radian"/>' + OPT + '<option timestep="0.002"/><worldbody><geom type="plane" size="0 0 1"/>'
                 '<geom type="box" size=".35 .3 .1" pos="0 0 0.1"/>' + bodies + '</worldbody></mujoco>')
        blob = B.dumps(F.prepare_for_kernels(C.compile_string(scene)))
        o = Oracle(blob); o.set_option("solver", 2 if solver == "newton" else 0)
        sim = StretchBatchSimulator(num_envs=1, device="cuda:0", model_blob_bytes=blob, solver=solver)
        sim.start(home=False)
        if solver == "pgs":
            sim.set_option("qcqp_exact", 1)   # MuJoCo's own QCQP iteration on both sides
            o.set_option("pgs_dual_warmstart", 1)   # the kernels' default: a second start from the previous step's forces, on both sides
        for step in range(200):
            for name, t in (("qpos", sim.qpos), ("qvel", sim.qvel), ("qacc_warmstart", sim.qacc_warmstart)):
                t[:, 0] = torch.tensor(o.arr(name), dtype=torch.float32, device=sim.device)
            o.step(1); sim.step(1)
            torch.cuda.synchronize()
            total += 1
            if int(sim.info[1, 0]) == o.ncon and int(sim.info[0, 0]) == o.nefc:
                agree += 1
                v = o.arr("qvel")
                errs.append(np.abs(sim.qvel[:, 0].cpu().numpy() - v).max() / max(1.0, np.abs(v).max()))
        assert int(sim.info[3, 0]) == 0, (pile, hex(int(sim.info[3, 0])))
        sim.stop()
    print(f"\npiles of {nbodies} [{solver}]: {agree} of {total} steps with equal contact / row counts; rel dqvel p50 {np.percentile(errs, 50):.1e} p90 {np.percentile(errs, 90):.1e} max {max(errs):.1e}")
    # (PGS observed: 1199 of 1200, p90 4e-6, max 1.3e-3)
    assert agree >= 0.99 * total and np.percentile(errs, 90) < 1e-4


@pytest.mark.parametrize("shape,kind", [("cylinder", "slide"), ("box", "slide"), ("cylinder", "spin")])
def test_kept_manifolds_on_a_slowly_sliding_or_turning_body(shape, kind):
    """the manifold cache keeps a resting pair's manifold while both bodies stay within 2e-5 of the poses it was
    built at and carries it to first order with their motion.  A body that creeps -- sliding down a 4 degree ramp with little
    friction, or turning slowly about the vertical on a flat top -- is the case that exercises the carry: the kernel source (lane
    emulator, cache on) against (i) the oracle's twin of the rule, state by state, and (ii) the UNMODIFIED oracle free-running, 300
    steps.  The cache must actually be used, the twin must be followed to fp32 rounding, and the rule must not move the body by more
    than 1e-5 against MuJoCo's rebuild-every-step."""
    from emul.emul import Emul, lib

    t, sz, h = SHAPES[shape]
    tilt = 0.07 if kind == "slide" else 0.0
    fr = "1 0.005 0.0001"      # (sticking contact: what moves the body down the ramp is the creep of the soft friction constraint, ~1e-6 m per step)
    scene = ('<mujoco><compiler angle="radian"/>' + OPT + '<option timestep="0.002"/><worldbody>'
             f'<geom type="box" size=".6 .5 .1" pos="0 0 0.1" euler="0 {tilt} 0" friction="{fr}"/>'
             f'<body pos="0 0 {0.2 + h + 0.0005}" euler="0 {tilt} 0"><freejoint/><geom type="{t}" size="{sz}" mass="0.4" friction="{fr}"/></body>'
             '</worldbody></mujoco>')
    blob = B.dumps(F.prepare_for_kernels(C.compile_string(scene)))
    plain, twin = Oracle(blob), Oracle(blob)
    for o in (plain, twin):
        o.set_option("solver", 2)
    nq, nv = plain.dim("nq"), plain.dim("nv")
    if kind == "spin":
        for o in (plain, twin):
            o.arr("qvel")[5] = 0.02          # 0.02 rad/s about z: 4e-5 rad per step -- every second step leaves the keep tolerance
    for o in (plain, twin):
        o.step(150)                           # come to rest on the top first (both without the rule: the same state)
    twin.set_option("manifold_keep", 1)
    e = Emul(blob, dict(nq=nq, nv=nv, nu=0, nlidar=0), num_envs=1, variant="standard", debug=True)
    e.set_option("solver", 2)
    e.set_option("manifold_cache", 1)         # (off by default for models of <= 32 dofs: a robot alone has no resting convex pair)
    e.qpos[:, 0] = plain.arr("qpos"); e.qvel[:, 0] = plain.arr("qvel"); e.warm[:, 0] = plain.arr("qacc_warmstart")
    hits0 = lib("standard").emul_mc_hits()
    worst_twin = 0.0
    for k in range(300):
        twin.step(1); plain.step(1); e.step(1)
        worst_twin = max(worst_twin, float(np.abs(e.qpos[:, 0] - twin.arr("qpos")).max()))
    hits = lib("standard").emul_mc_hits() - hits0
    d_plain = float(np.abs(e.qpos[:, 0] - plain.arr("qpos")).max())
    d_rule = float(np.abs(twin.arr("qpos") - plain.arr("qpos")).max())
    print(f"\n[{shape}, {kind}] kept manifolds used on {hits} of 300 steps (twin: {int(twin.iarr('mc_hits')[0])}); kernel vs twin {worst_twin:.1e}; "
          f"kernel vs unmodified oracle {d_plain:.1e}; twin vs unmodified {d_rule:.1e}; body at {plain.arr('qpos')[:3]}")
    assert int(e.info[3, 0]) == 0
    assert hits > 20 and int(twin.iarr("mc_hits")[0]) > 20
    assert worst_twin < 2e-5 and d_rule < 1e-5 and d_plain < 3e-5
