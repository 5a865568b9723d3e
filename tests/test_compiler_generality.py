"""MJCF features beyond stretch.xml that a kitchen export uses (robocasa_gen.py:129-280 emits robosuite
models): angle="degree" (MuJoCo's default), <inertial>, fromto geoms, capsule and ellipsoid collision / ray geoms.  The fp64
oracle against closed forms; the kernel logic (lane emulator) against the oracle."""
import math

import numpy as np
import pytest

from oracle.oracle import Oracle
from stretch_mujoco_amd import mjcf_compiler as C
from stretch_mujoco_amd import model_blob as B

OPT = '<option integrator="implicitfast" cone="elliptic" impratio="20"/>'


def _compile(xml):
    return C.compile_string(xml)


def test_degrees_are_the_default_and_equal_the_radian_document():
    body = ('<worldbody><body pos="0 0 1" euler="{e}"><joint type="hinge" axis="0 1 0" range="{r}" ref="{f}" springref="{f}" stiffness="1"/>'
            '<geom type="box" size=".1 .2 .3" axisangle="0 0 1 {a}"/></body></worldbody>')
    deg = _compile("<mujoco>" + OPT + body.format(e="30 45 60", r="-90 45", f="10", a="90") + "</mujoco>")   # no <compiler>: degree
    rad = _compile('<mujoco><compiler angle="radian"/>' + OPT
                   + body.format(e=f"{math.radians(30)} {math.radians(45)} {math.radians(60)}", r=f"{-math.pi / 2} {math.pi / 4}",
                                 f=f"{math.radians(10)}", a=f"{math.pi / 2}") + "</mujoco>")
    for k in ("body_quat", "jnt_range", "qpos0", "qpos_spring", "geom_quat", "body_inertia", "body_iquat"):
        assert np.allclose(deg[k], rad[k], atol=1e-12), k
    assert np.allclose(deg["jnt_range"][0], [-math.pi / 2, math.pi / 4])


def test_xyaxes_and_eulerseq_against_scipy():
    """`xyaxes` (x axis, y axis made orthogonal to it) and `<compiler eulerseq>`: lower case = rotations about the axes of the
    ROTATING frame, upper case = about the FIXED axes ([MJ] XML reference; scipy spells it the other way round)."""
    from scipy.spatial.transform import Rotation as R

    def quat_of(compiler, attr):
        m = _compile(f"<mujoco>{compiler}{OPT}<worldbody><body pos='0 0 1' {attr}><joint type='hinge' axis='0 1 0'/>"
                     "<geom type='box' size='.1 .2 .3'/></body></worldbody></mujoco>")
        return np.asarray(m["body_quat"][1])

    def same(q, r):   # MuJoCo (w, x, y, z) against a scipy rotation, up to sign
        x = r.as_quat()
        ref = np.array([x[3], x[0], x[1], x[2]])
        return min(np.abs(q - ref).max(), np.abs(q + ref).max()) < 1e-12

    assert same(quat_of("", "xyaxes='0 1 0 -1 0 0'"), R.from_euler("z", 90, degrees=True))
    assert same(quat_of("", "xyaxes='1 1 0 0 2 0'"), R.from_euler("z", 45, degrees=True))   # y is made orthogonal to x
    e = [20.0, -35.0, 50.0]
    for seq in ("xyz", "zyx", "yzx", "XYZ", "ZYX", "zxZ"):
        scipy_seq = seq.swapcase() if seq.islower() or seq.isupper() else None
        q = quat_of(f"<compiler eulerseq='{seq}'/>", "euler='%g %g %g'" % tuple(e))
        if scipy_seq is None:   # mixed case: compose by hand, one factor at a time
            r = R.identity()
            for ch, a in zip(seq, e):
                step = R.from_euler(ch.lower(), a, degrees=True)
                r = r * step if ch.islower() else step * r
        else:
            r = R.from_euler(scipy_seq, e, degrees=True)
        assert same(q, r), seq
    assert same(quat_of("<compiler angle='radian' eulerseq='ZYX'/>", "euler='%r %r %r'" % tuple(math.radians(a) for a in e)),
                R.from_euler("zyx", e, degrees=True))


def test_compiler_settings_do_not_leak_between_documents():
    """angle unit and eulerseq belong to the compiler instance of ONE document (they were module globals)."""
    body = "<worldbody><body pos='0 0 1' euler='{e}'><joint type='hinge' axis='0 1 0'/><geom type='box' size='.1 .2 .3'/></body></worldbody>"
    deg = "<mujoco>" + OPT + body.format(e="30 45 60") + "</mujoco>"
    rad = "<mujoco><compiler angle='radian' eulerseq='ZYX'/>" + OPT + body.format(e="0.3 0.2 0.1") + "</mujoco>"
    a1 = _compile(deg); b1 = _compile(rad); a2 = _compile(deg); b2 = _compile(rad)
    assert np.array_equal(a1["body_quat"], a2["body_quat"]) and np.array_equal(b1["body_quat"], b2["body_quat"])
    assert not hasattr(C, "_ANGLE_SCALE") and not hasattr(C, "_EULER_SEQ")
    assert np.allclose(C._orient({"euler": "90 0 0"}, math.pi / 180, "xyz"), [math.sqrt(0.5), math.sqrt(0.5), 0, 0])


def test_fromto_box_needs_both_cross_section_sizes():
    doc = "<mujoco>" + OPT + "<worldbody><body pos='0 0 1'><joint type='hinge' axis='0 1 0'/><geom name='bar' type='box' fromto='0 0 0 0 0 1' size='{s}'/></body></worldbody></mujoco>"
    m = _compile(doc.format(s=".05 .02"))
    assert np.allclose(m["geom_size"][0], [.05, .02, .5])
    with pytest.raises(ValueError, match="two size values"):
        _compile(doc.format(s=".05"))


def test_inertial_overrides_the_geoms():
    m = _compile('<mujoco><compiler angle="radian"/>' + OPT + '<worldbody><body><freejoint/><inertial pos="0.1 0 0" mass="2.5" '
                 'diaginertia="0.3 0.2 0.1"/><geom type="box" size=".1 .1 .1"/></body>'
                 '<body pos="1 0 0"><freejoint/><inertial pos="0 0 0" mass="1" fullinertia="0.2 0.3 0.1 0.01 0 0"/>'
                 '<geom type="sphere" size=".1"/></body></worldbody></mujoco>')
    assert m["body_mass"][1] == 2.5 and np.allclose(m["body_ipos"][1], [0.1, 0, 0]) and np.allclose(m["body_inertia"][1], [0.3, 0.2, 0.1])
    w = np.linalg.eigvalsh(np.array([[0.2, 0.01, 0], [0.01, 0.3, 0], [0, 0, 0.1]]))
    assert np.allclose(np.sort(m["body_inertia"][2]), np.sort(w))


def test_fromto_capsule_equals_the_explicit_geom():
    a = _compile('<mujoco><compiler angle="radian"/>' + OPT + '<worldbody><body><freejoint/>'
                 '<geom type="capsule" size="0.05" fromto="0 0 0.1 0.3 0 0.1"/></body></worldbody></mujoco>')
    b = _compile('<mujoco><compiler angle="radian"/>' + OPT + '<worldbody><body><freejoint/>'
                 '<geom type="capsule" size="0.05 0.15" pos="0.15 0 0.1" zaxis="-1 0 0"/></body></worldbody></mujoco>')
    for k in ("geom_pos", "geom_size", "body_mass", "body_ipos"):
        assert np.allclose(a[k], b[k], atol=1e-12), k
    assert np.allclose(np.sort(a["body_inertia"][1]), np.sort(b["body_inertia"][1]))
    assert abs(abs(np.dot(a["geom_quat"][0], b["geom_quat"][0])) - 1) < 1e-12


def _make(xml):
    return Oracle(B.dumps(_compile(xml)))


def test_capsule_rests_on_the_plane_on_two_contacts():
    """[MJ] mjc_PlaneCapsule: a lying capsule touches with its two end spheres; weight = the two normal forces."""
    o = _make('<mujoco><compiler angle="radian"/>' + OPT + '<worldbody><geom type="plane" size="0 0 1"/>'
              '<body pos="0 0 0.0502"><freejoint/><geom type="capsule" size="0.05 0.2" euler="0 1.5707963267948966 0" mass="2"/></body></worldbody></mujoco>')
    o.set_option("solver", 2)
    o.step(800)
    o.forward()
    c = o.arr("contact").reshape(o.ncon, -1)
    assert o.ncon == 2 and abs(abs(c[0][1] - c[1][1]) - 0.4) < 1e-3 and np.allclose(c[:, 4:7], [0, 0, 1], atol=1e-9)
    assert abs(o.arr("qpos")[2] - 0.05) < 1e-3 and np.abs(o.arr("qvel")).max() < 1e-4
    f = o.arr("efc_force")
    assert abs(f[0] + f[3] - 2 * 9.81) < 1e-3 * 2 * 9.81      # first rows of the two condim-3 contacts


def test_ellipsoid_plane_depth_and_support():
    """[MJ] mjc_PlaneEllipsoid: the contact point is the surface point whose normal is the plane's; tilted 30 degrees about y the
    lowest point of an ellipsoid with semi-axes (a, b, c) is sqrt(a^2 sin^2 + c^2 cos^2) below its centre."""
    a_, b_, c_, th = 0.3, 0.2, 0.1, math.radians(30)
    h = math.sqrt(a_ * a_ * math.sin(th) ** 2 + c_ * c_ * math.cos(th) ** 2)
    o = _make('<mujoco><compiler angle="radian"/>' + OPT + f'<worldbody><geom type="plane" size="0 0 1"/>'
              f'<body pos="0 0 {h - 0.01}"><freejoint/><geom type="ellipsoid" size="{a_} {b_} {c_}" euler="0 {th} 0"/></body></worldbody></mujoco>')
    o.forward()
    c = o.arr("contact").reshape(o.ncon, -1)
    assert o.ncon == 1 and abs(c[0][0] + 0.01) < 1e-12
    # ellipsoid against a box through MPR: a sphere-like ellipsoid (all semi-axes equal) must give the sphere's answer
    o = _make('<mujoco><compiler angle="radian"/>' + OPT + '<worldbody><body><freejoint/><geom type="box" size=".2 .2 .1"/></body>'
              '<body pos="0.03 0.02 0.195"><freejoint/><geom type="ellipsoid" size=".1 .1 .1"/></body></worldbody></mujoco>')
    o.forward()
    c = o.arr("contact").reshape(o.ncon, -1)
    assert o.ncon >= 1 and abs(c[0][0] + 0.005) < 1e-5 and np.allclose(np.abs(c[0][4:7]), [0, 0, 1], atol=1e-4)


def test_sphere_capsule_and_capsule_capsule_closed_forms():
    """[MJ] mjc_SphereCapsule / mjc_CapsuleCapsule (these pairs went through MPR + multiccd): the sphere against the
    nearest point of the capsule's segment; two capsules through the nearest points of their segments -- ONE contact when the axes
    cross, TWO (the overlapping ends) when they are parallel; depth and normal exact, no MPR tolerance."""
    two = lambda a, b: ('<mujoco><compiler angle="radian"/>' + OPT + f'<worldbody><body>{a}</body><body>{b}</body></worldbody></mujoco>')
    # sphere (r = 0.1) beside the cylindrical part of a capsule along x (r = 0.05, half length 0.3): distance 0.14 - 0.15 = -0.01
    o = _make(two('<freejoint/><geom type="sphere" size="0.1" pos="0.1 0.14 0"/>', '<freejoint/><geom type="capsule" size="0.05 0.3" euler="0 1.5707963267948966 0"/>'))
    o.forward()
    c = o.arr("contact").reshape(o.ncon, -1)
    assert o.ncon == 1 and abs(c[0][0] + 0.01) < 1e-12 and np.allclose(c[0][4:7], [0, -1, 0], atol=1e-12)
    assert np.allclose(c[0][1:4], [0.1, 0.14 - 0.1 + 0.005, 0], atol=1e-12)
    # ... and past the capsule's end: the end cap's sphere at x = 0.3
    o = _make(two('<freejoint/><geom type="sphere" size="0.1" pos="0.42 0 0"/>', '<freejoint/><geom type="capsule" size="0.05 0.3" euler="0 1.5707963267948966 0"/>'))
    o.forward()
    c = o.arr("contact").reshape(o.ncon, -1)
    assert o.ncon == 1 and abs(c[0][0] + 0.03) < 1e-12 and np.allclose(c[0][4:7], [-1, 0, 0], atol=1e-12)
    # capsule's body first, sphere's second: the compiled pair is ordered by geom type as MuJoCo's (sphere = geom1), same contact
    o = _make(two('<freejoint/><geom type="capsule" size="0.05 0.3" euler="0 1.5707963267948966 0"/>', '<freejoint/><geom type="sphere" size="0.1" pos="0.1 0.14 0"/>'))
    o.forward()
    c = o.arr("contact").reshape(o.ncon, -1)
    assert o.ncon == 1 and abs(c[0][0] + 0.01) < 1e-12 and np.allclose(c[0][4:7], [0, -1, 0], atol=1e-12)
    # crossed capsules (x axis and y axis, 0.09 apart in z, radii 0.05): one contact, depth 0.01, normal z
    o = _make(two('<freejoint/><geom type="capsule" size="0.05 0.3" euler="0 1.5707963267948966 0"/>',
                  '<freejoint/><geom type="capsule" size="0.05 0.3" pos="0.1 0.05 0.09" euler="1.5707963267948966 0 0"/>'))
    o.forward()
    c = o.arr("contact").reshape(o.ncon, -1)
    assert o.ncon == 1 and abs(c[0][0] + 0.01) < 1e-12 and np.allclose(c[0][4:7], [0, 0, 1], atol=1e-12)
    assert np.allclose(c[0][1:4], [0.1, 0, 0.045], atol=1e-12)
    # parallel capsules, the upper one shifted by 0.2 along the axis: two contacts, at the ends of the overlap (x = -0.1 and x = 0.3)
    o = _make(two('<freejoint/><geom type="capsule" size="0.05 0.3" euler="0 1.5707963267948966 0"/>',
                  '<freejoint/><geom type="capsule" size="0.05 0.3" pos="0.2 0 0.09" euler="0 1.5707963267948966 0"/>'))
    o.forward()
    c = o.arr("contact").reshape(o.ncon, -1)
    assert o.ncon == 2 and np.allclose(c[:, 0], -0.01, atol=1e-12) and np.allclose(c[:, 4:7], [0, 0, 1], atol=1e-12)
    assert sorted(np.round(c[:, 1], 9)) == [-0.1, 0.3]


def test_kernel_logic_capsule_closed_forms_vs_oracle():
    """The same closed forms in the step kernel's source (lane emulator): two capsules and a sphere dropped on each other and on a
    lying capsule, state-synchronised against the oracle -- contact counts equal on every step, velocities to fp32 rounding."""
    from emul.emul import Emul
    from stretch_mujoco_amd import model_fuse as F

    scene = ('<mujoco><compiler angle="radian"/>' + OPT + '<option timestep="0.002"/><worldbody><geom type="plane" size="0 0 1"/>'
             '<body pos="0 0 0.0495"><freejoint/><geom type="capsule" size="0.05 0.3" euler="0 1.5707963267948966 0" mass="2"/></body>'
             '<body pos="0.05 0.02 0.16"><freejoint/><geom type="capsule" size="0.04 0.15" euler="1.5707963267948966 0 0.2" mass="0.4"/></body>'
             '<body pos="-0.2 0.0 0.155"><freejoint/><geom type="capsule" size="0.05 0.1" euler="0 1.5707963267948966 0.6" mass="0.3"/></body>'
             '<body pos="0.22 0.01 0.2"><freejoint/><geom type="sphere" size="0.06" mass="0.3"/></body>'
             '</worldbody></mujoco>')
    m = F.prepare_for_kernels(_compile(scene))
    blob = B.dumps(m)
    o = Oracle(blob); o.set_option("solver", 2)
    e = Emul(blob, dict(nq=o.dim("nq"), nv=o.dim("nv"), nu=0, nlidar=0), num_envs=1, variant="standard"); e.set_option("solver", 2)
    errs, cc = [], 0
    for k in range(250):
        e.qpos[:, 0] = o.arr("qpos"); e.qvel[:, 0] = o.arr("qvel"); e.warm[:, 0] = o.arr("qacc_warmstart")
        o.step(1); e.step(1)
        assert int(e.info[3, 0]) == 0
        assert int(e.info[1, 0]) == o.ncon, (k, int(e.info[1, 0]), o.ncon)
        g = o.arr("contact").reshape(o.ncon, -1) if o.ncon else np.zeros((0, 8))
        cc += o.ncon > 2   # (more than the lying capsule's two plane contacts: a capsule / sphere pair is touching)
        errs.append(np.abs(e.qvel[:, 0] - o.arr("qvel")).max() / max(1.0, np.abs(o.arr("qvel")).max()))
    assert cc > 100 and np.percentile(errs, 90) < 1e-4 and max(errs) < 1e-2, (cc, np.percentile(errs, 90), max(errs))
    # exactly parallel capsules (the two-contact branch), one step from rest: same two contacts as the oracle
    par = ('<mujoco><compiler angle="radian"/>' + OPT + '<worldbody><body><freejoint/><geom type="capsule" size="0.05 0.3" euler="0 1.5707963267948966 0"/></body>'
           '<body pos="0.2 0 0.09"><freejoint/><geom type="capsule" size="0.05 0.3" euler="0 1.5707963267948966 0"/></body></worldbody></mujoco>')
    blob = B.dumps(F.prepare_for_kernels(_compile(par)))
    o = Oracle(blob); o.set_option("solver", 2)
    e = Emul(blob, dict(nq=o.dim("nq"), nv=o.dim("nv"), nu=0, nlidar=0), num_envs=1, variant="standard"); e.set_option("solver", 2)
    e.qpos[:, 0] = o.arr("qpos")
    o.step(1); e.step(1)
    assert int(e.info[1, 0]) == o.ncon == 2 and np.abs(e.qvel[:, 0] - o.arr("qvel")).max() < 1e-4 * max(1.0, np.abs(o.arr("qvel")).max())


def test_box_box_face_polygon_of_eight_points_with_the_option():
    """[MJ] mjc_BoxBox keeps up to 8 points of a face contact; oracle and kernels keep max_contacts_per_pair (default 4: the extreme
    points along the two axes of the reference face -- DESIGN.md section 7).  Two equal squares, one turned by 45 degrees, overlap in
    an octagon: with the option at 8 both sides give its eight corners, with the default both give four, same depths."""
    from emul.emul import Emul
    from stretch_mujoco_amd import model_fuse as F

    scene = ('<mujoco><compiler angle="radian"/>' + OPT + '<worldbody><geom type="box" size=".2 .2 .1" pos="0 0 0.1"/>'
             '<body pos="0 0 0.299" euler="0 0 0.7853981633974483"><freejoint/><geom type="box" size=".2 .2 .1" mass="1"/></body></worldbody></mujoco>')
    blob = B.dumps(F.prepare_for_kernels(_compile(scene)))
    for cap, want in ((8, 8), (4, 4)):
        o = Oracle(blob); o.set_option("solver", 2); o.set_option("max_contacts_per_pair", cap)
        e = Emul(blob, dict(nq=o.dim("nq"), nv=o.dim("nv"), nu=0, nlidar=0), num_envs=1, variant="standard", debug=True)
        e.set_option("solver", 2); e.set_option("max_contacts_per_pair", cap)
        e.qpos[:, 0] = o.arr("qpos")
        o.step(1); e.step(1)
        c = o.arr("contact").reshape(o.ncon, -1)
        assert o.ncon == want == int(e.info[1, 0]), (cap, o.ncon, int(e.info[1, 0]))
        assert np.allclose(c[:, 0], -0.001, atol=1e-9) and np.allclose(np.abs(c[:, 6]), 1, atol=1e-9)
        r = np.hypot(c[:, 1], c[:, 2])
        if cap == 8:   # the octagon's corners: |x| + |y| = 0.2 sqrt 2 on the turned square's edges and max(|x|, |y|) = 0.2 on the other's
            assert np.allclose(r, 0.2 / math.cos(math.pi / 8), atol=1e-6)
        assert np.abs(e.qvel[:, 0] - o.arr("qvel")).max() < 1e-4 * max(1.0, np.abs(o.arr("qvel")).max())


def test_capsule_box_penetration_through_mpr():
    """A vertical capsule pushed 4 mm into the top face of a box: depth and normal of the first MPR contact.  (KNOWN DEVIATION,
    DESIGN.md section 7: MuJoCo runs its closed form mjc_CapsuleBox here -- at most two contacts, no MPR tolerance; its ~300 lines
    are not restated from memory, capsule-box stays on MPR + multiccd in the oracle and the kernel.)"""
    o = _make('<mujoco><compiler angle="radian"/>' + OPT + '<worldbody><body><freejoint/><geom type="box" size=".2 .2 .1"/></body>'
              '<body pos="0.01 -0.02 0.346"><freejoint/><geom type="capsule" size=".05 .2"/></body></worldbody></mujoco>')
    o.forward()
    c = o.arr("contact").reshape(o.ncon, -1)
    assert o.ncon >= 1 and abs(c[0][0] + 0.004) < 2e-5 and np.allclose(np.abs(c[0][4:7]), [0, 0, 1], atol=1e-3)


SCENE = ('<mujoco><compiler angle="radian"/>' + OPT + '<option timestep="0.002"/><worldbody><geom type="plane" size="0 0 1"/>'
         '<geom type="box" size=".4 .4 .2" pos="0 0 0.2"/>'
         '<body pos="0.05 0.02 0.46"><freejoint/><geom type="capsule" size="0.04 0.12" euler="0 1.4 0.3" mass="0.4"/></body>'
         '<body pos="-0.15 -0.1 0.47"><freejoint/><geom type="ellipsoid" size="0.08 0.05 0.06" mass="0.3"/></body>'
         '<body pos="0.9 0.3 0.08"><freejoint/><geom type="capsule" size="0.05" fromto="0 0 0 0.2 0.1 0" mass="0.5"/></body>'
         '</worldbody></mujoco>')


def test_kernel_logic_with_capsules_and_ellipsoids_vs_oracle():
    """The step kernel's source (lane emulator) on a scene whose free bodies are capsules and an ellipsoid -- on a box (MPR +
    multiccd) and on the plane (closed forms): state-synchronised against the oracle while the bodies drop, roll and settle."""
    from emul.emul import Emul
    from stretch_mujoco_amd import model_fuse as F

    m = F.prepare_for_kernels(_compile(SCENE))
    blob = B.dumps(m)
    o = Oracle(blob); o.set_option("solver", 2)
    nq, nv = o.dim("nq"), o.dim("nv")
    e = Emul(blob, dict(nq=nq, nv=nv, nu=0, nlidar=0), num_envs=1, variant="standard"); e.set_option("solver", 2)
    errs, seen = [], 0
    for k in range(300):
        e.qpos[:, 0] = o.arr("qpos"); e.qvel[:, 0] = o.arr("qvel"); e.warm[:, 0] = o.arr("qacc_warmstart")
        o.step(1); e.step(1)
        assert int(e.info[3, 0]) == 0
        if (int(e.info[0, 0]), int(e.info[1, 0])) == (o.nefc, o.ncon):
            seen += o.ncon > 0
            errs.append(np.abs(e.qvel[:, 0] - o.arr("qvel")).max() / max(1.0, np.abs(o.arr("qvel")).max()))
    errs = np.sort(errs)
    assert seen > 150 and len(errs) > 250 and errs[int(0.9 * len(errs))] < 1e-3, (seen, len(errs), errs[-10:])


KX_SCRIPT = ((0, [0, 0, 1.05, 0.45, 0, -0.9, 0, 0.02, 0, 0]), (150, [0, 0, 0.9, 0.45, 0, -1.2, 0, -0.01, 0, 0]),
             (500, [0, 0, 0.9, 0.25, 0.6, -1.2, 0, -0.01, 0, 0]))   # reach over the counter, come down on the lemon, sweep it off


def kx_start(o):
    """Robot at the removed robosuite robot's spawn pose (what the converter returns), lift and arm near the first targets."""
    q = o.arr("qpos")
    q[0:3] = [0, -0.2, 0]; q[9] = 1.05; q[10:14] = 0.11


def test_kitchen_export_through_converter_compiler_and_kernel_logic():
    """tests/kitchen_export_fixture.py (robosuite-style export: articulated door and drawer, <inertial>, visual / collision
    geom groups, capsule / ellipsoid objects, markers, the robosuite robot) -> robocasa_import.convert_kitchen_xml ->
    mjcf_compiler -> model_fuse -> the committed blob stretch_kitchen_export.smjb (tools/build_models.py; 46 dofs, the 50-column
    big build).  The step kernel's source (lane emulator) against the fp64 oracle, state-synchronised, while the gripper comes
    down on the ellipsoid and sweeps it off the counter."""
    import json
    import os

    from conftest import MODELS
    from emul.emul import Emul

    blob = open(os.path.join(MODELS, "stretch_kitchen_export.smjb"), "rb").read()
    names = json.loads(B.get_str(B.loads(blob), "names_json"))
    assert {"door_hinge", "drawer_slide", "bottle_joint0", "lemon_joint0", "spatula_joint0"} <= set(names["joint"])
    assert not any(n.startswith("robot0") for n in names["joint"] + names["body"])
    o = Oracle(blob); o.set_option("solver", 2)
    kx_start(o)
    e = Emul(blob, dict(nq=o.dim("nq"), nv=o.dim("nv"), nu=10, nlidar=360), num_envs=1); e.set_option("solver", 2)
    assert e.variant == "big50"
    errs, same = [], 0
    for k in range(400):
        for k0, c in KX_SCRIPT:
            if k == k0:
                o.arr("ctrl")[:] = c; e.ctrl[:, 0] = np.asarray(c, np.float32)
        e.qpos[:, 0] = o.arr("qpos"); e.qvel[:, 0] = o.arr("qvel"); e.warm[:, 0] = o.arr("qacc_warmstart")
        o.step(1); e.step(1)
        assert int(e.info[3, 0]) == 0, k
        if (int(e.info[0, 0]), int(e.info[1, 0])) == (o.nefc, o.ncon):
            same += 1
            errs.append(np.abs(e.qvel[:, 0] - o.arr("qvel")).max() / max(1.0, np.abs(o.arr("qvel")).max()))
    errs = np.sort(errs)
    assert same >= 380 and errs[int(0.9 * len(errs))] < 2e-4 and o.arr("qpos")[38] < 0.9, (same, errs[-5:])   # (the lemon has left the counter top)


def test_inertiagrouprange_limits_which_geoms_carry_mass():
    m = _compile('<mujoco><compiler angle="radian" inertiagrouprange="0 0"/>' + OPT + '<worldbody><body><freejoint/>'
                 '<geom type="box" size=".1 .1 .1" group="0" mass="1"/><geom type="box" size=".2 .2 .2" group="1"/></body></worldbody></mujoco>')
    assert m["body_mass"][1] == 1.0
