"""Compiled model blob -> standalone MJCF, for the live-MuJoCo harness (tools/mujoco_harness.py).

The GPU box has neither /root/reference nor its mesh files, so the harness cannot hand MuJoCo the reference's own
stretch.xml.  What it can hand over is THIS build's compiled model (models/*.smjb), written back out as MJCF with nothing left
for MuJoCo's compiler to decide:

  * every body carries an explicit <inertial> (pos, quat, mass, diaginertia) -- the blob's values, incl. the mesh-derived ones;
  * collision meshes are inline <mesh vertex="..."> assets holding the blob's hull vertices (geom frame);
  * collisions are the blob's explicit pair table (<contact><pair .../>) with its mixed parameters; all geoms get
    contype = conaffinity = 0, so MuJoCo's own pair generation and filtering play no part;
  * actuators are <general> with the blob's gain / bias parameters, joints carry ref / springref / armature / damping /
    frictionloss / limits explicitly, options are spelled out (implicitfast, elliptic, impratio 20, Newton, multiccd).

A mismatch that MuJoCo then shows against the oracle / the HIP kernels is a difference in the ARITHMETIC (the [MJ] items of the
survey), not in how the model was put together.  `with_visual` adds the render meshes (geom groups 0-2), which the
rangefinder sensors see; they come from the product blob only.
"""
from __future__ import annotations

import json
from typing import Dict

import numpy as np

from . import model_blob

_GEOM_TYPES = {0: "plane", 2: "sphere", 3: "capsule", 4: "ellipsoid", 5: "cylinder", 6: "box", 7: "mesh"}
_JNT_TYPES = {0: "free", 1: "ball", 2: "slide", 3: "hinge"}


def _f(a) -> str:
    return " ".join(repr(float(x)) for x in np.asarray(a, np.float64).ravel())


def export_mjcf(m: Dict[str, np.ndarray], with_visual: bool = False, model_name: str = "smj_export") -> str:
    """MJCF text of a compiled (fused or body-for-body) model dict (model_blob.loads)."""
    names = json.loads(model_blob.get_str(m, "names_json"))
    nb, njnt, ngeom = int(m["dims"][3]), int(m["dims"][4]), int(m["dims"][5])
    fused_from = m.get("fused_from")

    def bname(b):
        src = int(fused_from[b]) if fused_from is not None else b
        return names["body"][src] if src < len(names["body"]) and names["body"][src] else f"body{b}"

    jname = lambda j: names["joint"][j] or f"joint{j}"
    gname = lambda g: f"g{g}"
    out = [f'<mujoco model="{model_name}">',
           '<compiler angle="radian" autolimits="false" inertiafromgeom="false" boundmass="0" boundinertia="0" balanceinertia="false"/>']
    integ = {0: "Euler", 1: "RK4", 2: "implicit", 3: "implicitfast"}.get(int(np.ravel(m["opt_integrator"])[0]), "implicitfast")
    cone = "elliptic" if int(np.ravel(m["opt_cone"])[0]) == 1 else "pyramidal"
    out.append(f'<option timestep="{_f(m["opt_timestep"])}" gravity="{_f(m["opt_gravity"])}" impratio="{_f(m["opt_impratio"])}" '
               f'tolerance="{_f(m["opt_tolerance"])}" iterations="{int(np.ravel(m["opt_iterations"])[0])}" cone="{cone}" integrator="{integ}" '
               f'solver="Newton"><flag multiccd="enable"/></option>')
    ext = np.ravel(m["vis_znear_zfar_extent"]) if "vis_znear_zfar_extent" in m else None
    if ext is not None:
        out.append(f'<statistic extent="{_f(ext[2])}"/><visual><map znear="{_f(ext[0])}" zfar="{_f(ext[1])}"/></visual>')
    # ---- assets: one inline hull mesh per collision mesh geom (+ render meshes)
    hv = np.asarray(m["hull_vert"], np.float64).reshape(-1, 3)
    out.append("<asset>")
    for g in range(ngeom):
        if int(m["geom_type"][g]) == 7 and int(m["geom_hullnum"][g]) > 0:
            a, n = int(m["geom_hulladr"][g]), int(m["geom_hullnum"][g])
            out.append(f'<mesh name="hull{g}" vertex="{_f(hv[a:a + n])}"/>')
    visual = []
    if with_visual and "rmesh_vert" in m and len(np.asarray(m["rmesh_vert"])):
        rv, rf = np.asarray(m["rmesh_vert"], np.float64).reshape(-1, 3), np.asarray(m["rmesh_face"]).reshape(-1, 3)
        for g in range(ngeom):
            r = int(m["geom_rmeshid"][g])
            if r >= 0 and int(m["geom_contype"][g]) == 0 and int(m["rmesh_vertnum"][r]) > 0:
                va, vn, fa, fn = int(m["rmesh_vertadr"][r]), int(m["rmesh_vertnum"][r]), int(m["rmesh_faceadr"][r]), int(m["rmesh_facenum"][r])
                out.append(f'<mesh name="vis{g}" vertex="{_f(rv[va:va + vn])}" face="{" ".join(str(int(x)) for x in rf[fa:fa + fn].ravel())}"/>')
                visual.append(g)
    out.append("</asset>")
    # ---- body tree
    children = {b: [] for b in range(nb)}
    for b in range(1, nb):
        children[int(m["body_parentid"][b])].append(b)
    geoms_of = {b: [] for b in range(nb)}
    for g in range(ngeom):
        geoms_of[int(m["geom_bodyid"][g])].append(g)
    sites_of = {b: [] for b in range(nb)}
    for s in range(len(m["site_bodyid"])):
        sites_of[int(m["site_bodyid"][s])].append(s)
    cams_of = {b: [] for b in range(nb)}
    for c in range(len(m["cam_bodyid"])):
        cams_of[int(m["cam_bodyid"][c])].append(c)
    collide = set(int(g) for g in m["pair_geom1"]) | set(int(g) for g in m["pair_geom2"])

    def emit_geom(g):
        t = int(m["geom_type"][g])
        if t == 7 and int(m["geom_hullnum"][g]) == 0 and g not in visual:
            return   # a visual mesh without render data: no part in the physics
        if g not in collide and g not in visual and t == 7:
            return
        mesh = ""
        if t == 7:
            mesh = f' mesh="{"hull" if g in collide else "vis"}{g}"'
        size = "" if t == 7 else f' size="{_f(m["geom_size"][g][: {0: 3, 2: 1, 3: 2, 4: 3, 5: 2, 6: 3}[t]])}"'
        out.append(f'<geom name="{gname(g)}" type="{_GEOM_TYPES[t]}"{mesh}{size} pos="{_f(m["geom_pos"][g])}" quat="{_f(m["geom_quat"][g])}" '
                   f'rgba="{_f(m["geom_rgba"][g])}" group="{int(m["geom_group"][g])}" contype="0" conaffinity="0" condim="{int(m["geom_condim"][g])}" '
                   f'friction="{_f(m["geom_friction"][g])}" priority="{int(m["geom_priority"][g])}" mass="0"/>')

    def emit_body(b):
        if b > 0:
            out.append(f'<body name="{bname(b)}" pos="{_f(m["body_pos"][b])}" quat="{_f(m["body_quat"][b])}" gravcomp="{_f(m["body_gravcomp"][b])}">')
            if float(m["body_mass"][b]) > 0:
                out.append(f'<inertial pos="{_f(m["body_ipos"][b])}" quat="{_f(m["body_iquat"][b])}" mass="{_f(m["body_mass"][b])}" '
                           f'diaginertia="{_f(m["body_inertia"][b])}"/>')
            for j in range(int(m["body_jntadr"][b]), int(m["body_jntadr"][b]) + int(m["body_jntnum"][b])):
                t = int(m["jnt_type"][j])
                if t == 0:
                    out.append(f'<freejoint name="{jname(j)}"/>')
                    continue
                d, q = int(m["jnt_dofadr"][j]), int(m["jnt_qposadr"][j])
                out.append(f'<joint name="{jname(j)}" type="{_JNT_TYPES[t]}" pos="{_f(m["jnt_pos"][j])}" axis="{_f(m["jnt_axis"][j])}" '
                           f'ref="{_f(m["qpos0"][q])}" springref="{_f(m["qpos_spring"][q])}" stiffness="{_f(m["jnt_stiffness"][j])}" '
                           f'limited="{"true" if int(m["jnt_limited"][j]) else "false"}" range="{_f(m["jnt_range"][j])}" margin="{_f(m["jnt_margin"][j])}" '
                           f'solreflimit="{_f(m["jnt_solref"][j])}" solimplimit="{_f(m["jnt_solimp"][j])}" armature="{_f(m["dof_armature"][d])}" '
                           f'damping="{_f(m["dof_damping"][d])}" frictionloss="{_f(m["dof_frictionloss"][d])}" '
                           f'solreffriction="{_f(m["dof_solref"][d])}" solimpfriction="{_f(m["dof_solimp"][d])}"/>')
        for g in geoms_of[b]:
            emit_geom(g)
        for s in sites_of[b]:
            out.append(f'<site name="{names["site"][s] or f"site{s}"}" pos="{_f(m["site_pos"][s])}" quat="{_f(m["site_quat"][s])}"/>')
        for c in cams_of[b]:
            out.append(f'<camera name="{names["camera"][c]}" pos="{_f(m["cam_pos"][c])}" quat="{_f(m["cam_quat"][c])}" fovy="{float(m["cam_fovy"][c])!r}"/>')   # fovy is always in degrees in MJCF
        for ch in children[b]:
            emit_body(ch)
        if b > 0:
            out.append("</body>")

    out.append("<worldbody>")
    emit_body(0)
    out.append("</worldbody>")
    # ---- explicit contact pairs with the blob's mixed parameters
    out.append("<contact>")
    for p in range(len(m["pair_geom1"])):
        out.append(f'<pair geom1="{gname(int(m["pair_geom1"][p]))}" geom2="{gname(int(m["pair_geom2"][p]))}" condim="{int(m["pair_condim"][p])}" '
                   f'friction="{_f(m["pair_friction"][p])}" solref="{_f(m["pair_solref"][p])}" solimp="{_f(m["pair_solimp"][p])}" '
                   f'margin="{_f(m["pair_margin"][p])}" gap="{_f(m["pair_gap"][p])}"/>')
    out.append("</contact>")
    # ---- fixed tendons
    if len(names.get("tendon", [])):
        out.append("<tendon>")
        for t, tn in enumerate(names["tendon"]):
            out.append(f'<fixed name="{tn}">')
            a, n = int(m["tendon_adr"][t]), int(m["tendon_num"][t])
            for w in range(a, a + n):
                out.append(f'<joint joint="{jname(int(m["wrap_objid"][w]))}" coef="{_f(m["wrap_prm"][w])}"/>')
            out.append("</fixed>")
        out.append("</tendon>")
    # ---- joint equalities
    if len(m["eq_obj1id"]):
        out.append("<equality>")
        for e in range(len(m["eq_obj1id"])):
            j2 = int(m["eq_obj2id"][e])
            second = f' joint2="{jname(j2)}"' if j2 >= 0 else ""
            out.append(f'<joint joint1="{jname(int(m["eq_obj1id"][e]))}"{second} polycoef="{_f(m["eq_data"][e][:5])}" solref="{_f(m["eq_solref"][e])}" '
                       f'solimp="{_f(m["eq_solimp"][e])}" active="{"true" if int(m["eq_active"][e]) else "false"}"/>')
        out.append("</equality>")
    # ---- actuators as <general>
    out.append("<actuator>")
    for a, an in enumerate(names["actuator"]):
        trn = int(m["actuator_trntype"][a])
        # the blob's transmission code: 0 joint, 1 fixed tendon (mjcf_compiler.py)
        target = f'tendon="{names["tendon"][int(m["actuator_trnid"][a])]}"' if trn == 1 else f'joint="{jname(int(m["actuator_trnid"][a]))}"'
        out.append(f'<general name="{an}" {target} gear="{_f(m["actuator_gear"][a])}" gaintype="fixed" gainprm="{_f(m["actuator_gainprm"][a])}" '
                   f'biastype="{"affine" if int(m["actuator_biastype"][a]) == 1 else "none"}" biasprm="{_f(m["actuator_biasprm"][a])}" '
                   f'ctrllimited="{"true" if int(m["actuator_ctrllimited"][a]) else "false"}" ctrlrange="{_f(m["actuator_ctrlrange"][a])}" '
                   f'forcelimited="{"true" if int(m["actuator_forcelimited"][a]) else "false"}" forcerange="{_f(m["actuator_forcerange"][a])}"/>')
    out.append("</actuator>")
    # ---- sensors: gyro + accelerometer at the IMU site, one rangefinder per lidar site
    out.append("<sensor>")
    imu = int(np.ravel(m["sensor_imu_site"])[0])
    if imu >= 0:
        sn = names["site"][imu]
        out.append(f'<gyro name="base_gyro" site="{sn}"/><accelerometer name="base_accel" site="{sn}"/>')
    cut = float(np.ravel(m["sensor_lidar_cutoff"])[0])
    for i, s in enumerate(np.ravel(m["sensor_lidar_site"])):
        out.append(f'<rangefinder name="base_lidar{i:03d}" site="{names["site"][int(s)]}" cutoff="{cut!r}"/>')
    out.append("</sensor>")
    out.append("<keyframe>")
    for k, kn in enumerate(names["key"]):
        out.append(f'<key name="{kn}" ctrl="{_f(m["key_ctrl"][k])}"/>')
    out.append("</keyframe></mujoco>")
    return "\n".join(out)


def export_blob_file(path: str, with_visual: bool = False) -> str:
    with open(path, "rb") as f:
        return export_mjcf(model_blob.loads(f.read()), with_visual=with_visual)
