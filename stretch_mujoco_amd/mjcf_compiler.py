"""MJCF -> compiled model arrays for the batched Stretch simulator.

Stands in for `MjModel.from_xml_path(scene_xml_path)` in the reference
(`stretch_mujoco/mujoco_server.py:252`): it resolves default classes,
`<include>`, `<replicate>`, orientation forms, mesh mass properties and convex
hulls, and produces the flat arrays consumed by the HIP kernels (through the
C-ABI blob, see model_blob.py) and by the fp64 CPU oracle.

Only the MJCF subset used by `stretch_mujoco/models/{stretch,scene,docking_station}.xml`
and by simple primitive scenes is supported.  Semantics follow MuJoCo 3.2.6's
documented compiler behaviour ([MJ] in SURVEY.md); MuJoCo itself is not
available to cross-check, so every derived constant here is "parity unpinned".

Host-side, offline, numpy only.  Nothing in here runs on the hot path.
"""
from __future__ import annotations

import copy
import json
import math
import os
import struct
import xml.etree.ElementTree as ET
from typing import Dict, List, Optional, Tuple

import numpy as np

# MuJoCo enum values kept so that dumps can be diffed against a real MjModel.
JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = 0, 1, 2, 3
GEOM_PLANE, GEOM_HFIELD, GEOM_SPHERE, GEOM_CAPSULE, GEOM_ELLIPSOID, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH = range(8)
GEOM_TYPES = {"plane": 0, "hfield": 1, "sphere": 2, "capsule": 3, "ellipsoid": 4, "cylinder": 5, "box": 6, "mesh": 7}
MINVAL = 1e-15


# ----------------------------------------------------------------------------- math
def quat_mul(a, b):
    a = np.asarray(a, float)
    b = np.asarray(b, float)
    return np.array([
        a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
        a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
        a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
        a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0],
    ])


def quat_norm(q):
    q = np.asarray(q, float)
    n = np.linalg.norm(q)
    return np.array([1.0, 0, 0, 0]) if n < MINVAL else q / n


def quat_conj(q):
    return np.array([q[0], -q[1], -q[2], -q[3]])


def quat2mat(q):
    w, x, y, z = q
    return np.array([
        [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z],
    ])


def mat2quat(R):
    # robust (Shepperd)
    R = np.asarray(R, float)
    t = np.trace(R)
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = math.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        q = [(R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s]
    elif R[1, 1] > R[2, 2]:
        s = math.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        q = [(R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s]
    else:
        s = math.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        q = [(R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s]
    return quat_norm(np.array(q))


def axisangle2quat(axis, angle):
    axis = np.asarray(axis, float)
    n = np.linalg.norm(axis)
    if n < MINVAL:
        return np.array([1.0, 0, 0, 0])
    s = math.sin(angle / 2)
    return np.concatenate([[math.cos(angle / 2)], axis / n * s])


def euler2quat(e, seq="xyz"):
    """MuJoCo `euler` attribute: lowercase = intrinsic (rotating axes), applied left to right."""
    q = np.array([1.0, 0, 0, 0])
    for ang, ax in zip(e, seq):
        axis = {"x": [1, 0, 0], "y": [0, 1, 0], "z": [0, 0, 1]}[ax.lower()]
        r = axisangle2quat(axis, ang)
        q = quat_mul(q, r) if ax.islower() else quat_mul(r, q)
    return quat_norm(q)


def zaxis2quat(z):
    """Minimal rotation that takes (0,0,1) to z (MuJoCo `zaxis` attribute)."""
    z = np.asarray(z, float)
    z = z / np.linalg.norm(z)
    c = np.cross([0, 0, 1.0], z)
    s = np.linalg.norm(c)
    ang = math.atan2(s, z[2])
    if s < MINVAL:
        return np.array([1.0, 0, 0, 0]) if z[2] > 0 else np.array([0.0, 1, 0, 0])
    return axisangle2quat(c / s, ang)


def _floats(s, n=None):
    v = [float(x) for x in s.split()]
    if n is not None and len(v) < n:
        v = v + [0.0] * (n - len(v))
    return v


# ----------------------------------------------------------------------------- meshes
def load_obj(path) -> Tuple[np.ndarray, np.ndarray]:
    verts, faces = [], []
    with open(path, "r", errors="ignore") as f:
        for line in f:
            if line.startswith("v "):
                p = line.split()
                verts.append((float(p[1]), float(p[2]), float(p[3])))
            elif line.startswith("f "):
                idx = [int(tok.split("/")[0]) for tok in line.split()[1:]]
                idx = [i - 1 if i > 0 else len(verts) + i for i in idx]
                for k in range(1, len(idx) - 1):
                    faces.append((idx[0], idx[k], idx[k + 1]))
    return np.array(verts, float).reshape(-1, 3), np.array(faces, np.int64).reshape(-1, 3)


def load_stl(path) -> Tuple[np.ndarray, np.ndarray]:
    with open(path, "rb") as f:
        data = f.read()
    n = struct.unpack_from("<I", data, 80)[0]
    if 84 + 50 * n != len(data):
        raise ValueError(f"{path}: only binary STL is supported")
    rec = np.frombuffer(data, dtype=np.dtype([("n", "<f4", 3), ("v", "<f4", (3, 3)), ("a", "<u2")]), count=n, offset=84)
    tri = rec["v"].astype(np.float64).reshape(-1, 3)
    # merge repeated vertices (MuJoCo does the same for STL)
    uniq, inv = np.unique(np.round(tri, 9), axis=0, return_inverse=True)
    return uniq, inv.reshape(-1, 3).astype(np.int64)


def load_mesh(path):
    ext = os.path.splitext(path)[1].lower()
    if ext == ".obj":
        return load_obj(path)
    if ext == ".stl":
        return load_stl(path)
    raise ValueError(f"unsupported mesh format: {path}")


def _tet_cov(a, b, c, d):
    """Covariance integral  int (x-0)(x-0)^T dV  of tetrahedra (a,b,c,d) stacked on axis 0, and their volumes."""
    M = np.stack([b - a, c - a, d - a], axis=-1)  # columns
    det = np.linalg.det(M)
    vol = det / 6.0
    s = a + b + c + d
    # int x x^T dV = vol/20 * (sum_i p_i p_i^T + s s^T)
    P = sum(np.einsum("ni,nj->nij", p, p) for p in (a, b, c, d)) + np.einsum("ni,nj->nij", s, s)
    return P * (vol / 20.0)[:, None, None], vol, s / 4.0


def mesh_volume_props(verts, faces, legacy=True):
    """(volume, com, inertia tensor about com) of a unit-density solid mesh.

    `legacy=True` reproduces MuJoCo <= 3.2 default (exactmeshinertia="false"): tetrahedra
    are built from the area-weighted surface centroid and their volumes are taken in
    absolute value.  `legacy=False` is the exact signed-volume integration.
    """
    a, b, c = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    area2 = np.linalg.norm(np.cross(b - a, c - a), axis=1)
    keep = area2 > 1e-18
    a, b, c, area2 = a[keep], b[keep], c[keep], area2[keep]
    cen0 = ((a + b + c) / 3.0 * area2[:, None]).sum(0) / area2.sum()
    o = np.broadcast_to(cen0, a.shape)
    cov, vol, tc = _tet_cov(o, a, b, c)
    if legacy:
        sgn = np.sign(vol)
        sgn[sgn == 0] = 1.0
        cov = cov * sgn[:, None, None]
        vol = np.abs(vol)
    V = vol.sum()
    if abs(V) < 1e-18:
        raise ValueError("mesh volume is zero")
    com = (tc * vol[:, None]).sum(0) / V
    C = cov.sum(0)  # second moment about origin
    C = C - V * np.outer(com, com)
    inertia = np.trace(C) * np.eye(3) - C
    return float(V), com, inertia


def mesh_shell_props(verts, faces):
    """(area, com, inertia about com) of a unit-surface-density shell."""
    a, b, c = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
    A = area.sum()
    com = ((a + b + c) / 3.0 * area[:, None]).sum(0) / A
    s = a + b + c
    P = sum(np.einsum("ni,nj->nij", p, p) for p in (a, b, c)) + np.einsum("ni,nj->nij", s, s)
    C = (P * (area / 12.0)[:, None, None]).sum(0) - A * np.outer(com, com)
    inertia = np.trace(C) * np.eye(3) - C
    return float(A), com, inertia


def convex_hull_vertices(verts):
    from scipy.spatial import ConvexHull, QhullError

    v = np.unique(np.round(verts, 9), axis=0)
    try:
        hull = ConvexHull(v)
        return v[np.sort(hull.vertices)]
    except QhullError:
        hull = ConvexHull(v, qhull_options="QJ")
        return v[np.sort(hull.vertices)]


def ray_triangles(orig, dirs, tri, chunk=20000):
    """Nearest hit distance of rays (orig [R,3], unit dirs [R,3]) with triangles tri [T,3,3]; -1 where none.
    Moeller-Trumbore, both faces, vectorised over rays x triangles in chunks."""
    R = len(orig)
    best = np.full(R, np.inf)
    for s0 in range(0, len(tri), chunk):
        t = tri[s0:s0 + chunk]
        v0, e1, e2 = t[:, 0], t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]
        for r in range(R):
            p = np.cross(dirs[r], e2)
            det = np.einsum("ij,ij->i", e1, p)
            ok = np.abs(det) > 1e-20
            inv = np.where(ok, 1.0 / np.where(ok, det, 1.0), 0.0)
            tv = orig[r] - v0
            u = np.einsum("ij,ij->i", tv, p) * inv
            q = np.cross(tv, e1)
            v = (q @ dirs[r]) * inv
            x = np.einsum("ij,ij->i", e2, q) * inv
            hit = ok & (u >= 0) & (v >= 0) & (u + v <= 1) & (x >= 0)
            if hit.any():
                best[r] = min(best[r], x[hit].min())
    return np.where(np.isfinite(best), best, -1.0)


def _ray_primitive(t, size, lp, lv):
    """[MJ] mj_rayGeom for plane/sphere/cylinder/box in the geom frame (same rules as the kernels); -1 if no hit."""
    def quad(a, b, c):
        det = b * b - a * c
        if det < MINVAL:
            return -1.0
        det = math.sqrt(det)
        x0, x1 = (-b - det) / a, (-b + det) / a
        return x0 if x0 >= 0 else (x1 if x1 >= 0 else -1.0)

    if t == GEOM_SPHERE:
        return quad(lv @ lv, lv @ lp, lp @ lp - size[0] ** 2)
    if t == GEOM_CYLINDER:
        best = -1.0
        a, b, c = lv[0] ** 2 + lv[1] ** 2, lv[0] * lp[0] + lv[1] * lp[1], lp[0] ** 2 + lp[1] ** 2 - size[0] ** 2
        if a > MINVAL:
            x = quad(a, b, c)
            if x >= 0 and abs(lp[2] + x * lv[2]) <= size[1]:
                best = x
        if abs(lv[2]) > MINVAL:
            for sg in (-1, 1):
                x = (sg * size[1] - lp[2]) / lv[2]
                if x >= 0 and (lp[0] + x * lv[0]) ** 2 + (lp[1] + x * lv[1]) ** 2 <= size[0] ** 2 and (best < 0 or x < best):
                    best = x
        return best
    if t == GEOM_BOX:
        best = -1.0
        for ax in range(3):
            if abs(lv[ax]) < MINVAL:
                continue
            for sg in (-1, 1):
                x = (sg * size[ax] - lp[ax]) / lv[ax]
                a1, a2 = (ax + 1) % 3, (ax + 2) % 3
                if x >= 0 and abs(lp[a1] + x * lv[a1]) <= size[a1] and abs(lp[a2] + x * lv[a2]) <= size[a2] and (best < 0 or x < best):
                    best = x
        return best
    return -1.0


# ----------------------------------------------------------------------------- defaults
_GEOM_DEF = dict(type="sphere", size="0 0 0", pos="0 0 0", contype="1", conaffinity="1", condim="3", group="0",
                 priority="0", friction="1 0.005 0.0001", solmix="1", solref="0.02 1",
                 solimp="0.9 0.95 0.001 0.5 2", margin="0", gap="0", density="1000", rgba="0.5 0.5 0.5 1")
_JOINT_DEF = dict(type="hinge", pos="0 0 0", axis="0 0 1", stiffness="0", damping="0", armature="0",
                  frictionloss="0", springref="0", ref="0", margin="0", solreflimit="0.02 1",
                  solimplimit="0.9 0.95 0.001 0.5 2", solreffriction="0.02 1",
                  solimpfriction="0.9 0.95 0.001 0.5 2")
_SITE_DEF = dict(pos="0 0 0", size="0.005")
_CAM_DEF = dict(pos="0 0 0", fovy="45")
_EQ_DEF = dict(solref="0.02 1", solimp="0.9 0.95 0.001 0.5 2", active="true")
_ACT_DEF = dict(gainprm=[1.0, 0, 0], biasprm=[0.0, 0, 0], gear=[1.0, 0, 0, 0, 0, 0], ctrlrange=None, forcerange=None,
                ctrllimited="auto", forcelimited="auto", gaintype="fixed", biastype="none")
_ACT_TAGS = ("general", "motor", "position", "velocity")


def _apply_actuator(tag: str, attrib: Dict[str, str], st: dict) -> dict:
    """Fold one <general|motor|position|velocity> element (or default) into actuator state `st`.

    Shortcut semantics [MJ]: <position> sets gain=kp and ALWAYS biasprm[1]=-gainprm[0] (also when
    kp is inherited), biasprm[2]=-kv only when kv is given; <velocity> sets gain=kv,
    bias=(0,0,-kv).
    """
    st = copy.deepcopy(st)
    if "gainprm" in attrib:
        v = _floats(attrib["gainprm"])
        st["gainprm"][: len(v)] = v[:3]
    if "biasprm" in attrib:
        v = _floats(attrib["biasprm"])
        st["biasprm"][: len(v)] = v[:3]
    if "gear" in attrib:
        v = _floats(attrib["gear"])
        st["gear"][: len(v)] = v
    for k in ("ctrlrange", "forcerange"):
        if k in attrib:
            st[k] = _floats(attrib[k], 2)
    for k in ("ctrllimited", "forcelimited", "gaintype", "biastype"):
        if k in attrib:
            st[k] = attrib[k]
    if tag == "motor":
        st["gainprm"] = [1.0, 0, 0]
        st["biasprm"] = [0.0, 0, 0]
        st["gaintype"], st["biastype"] = "fixed", "none"
    elif tag == "position":
        if "kp" in attrib:
            st["gainprm"][0] = float(attrib["kp"])
        st["biasprm"][1] = -st["gainprm"][0]
        if "kv" in attrib:
            st["biasprm"][2] = -float(attrib["kv"])
        st["gaintype"], st["biastype"] = "fixed", "affine"
    elif tag == "velocity":
        if "kv" in attrib:
            st["gainprm"][0] = float(attrib["kv"])
        st["biasprm"] = [0.0, 0.0, -st["gainprm"][0]]
        st["gaintype"], st["biastype"] = "fixed", "affine"
    return st


class _Defaults:
    def __init__(self):
        self.cls: Dict[str, dict] = {"main": dict(geom={}, joint={}, site={}, camera={}, equality={}, mesh={},
                                                  actuator=copy.deepcopy(_ACT_DEF))}

    def parse(self, elem: ET.Element, parent: str = "main", top=True):
        name = elem.get("class", "main" if top else None)
        if name is None:
            raise ValueError("nested <default> needs a class")
        if name not in self.cls:
            self.cls[name] = copy.deepcopy(self.cls[parent])
        cur = self.cls[name]
        for ch in elem:
            if ch.tag == "default":
                continue
            if ch.tag in _ACT_TAGS:
                cur["actuator"] = _apply_actuator(ch.tag, ch.attrib, cur["actuator"])
            elif ch.tag in cur:
                cur[ch.tag].update(ch.attrib)
        for ch in elem:
            if ch.tag == "default":
                self.parse(ch, name, top=False)

    def get(self, cls: Optional[str], tag: str) -> dict:
        return self.cls[cls or "main"][tag]


# ----------------------------------------------------------------------------- spec tree
class _Body:
    def __init__(self, name, parent, pos, quat, gravcomp=0.0):
        self.name, self.parent, self.pos, self.quat, self.gravcomp = name, parent, np.asarray(pos, float), quat, gravcomp
        self.joints: List[dict] = []
        self.geoms: List[dict] = []
        self.sites: List[dict] = []
        self.cams: List[dict] = []
        self.inertial: Optional[dict] = None
        self.id = -1


def _orient(attrib: Dict[str, str], angle_scale: float, eulerseq: str) -> np.ndarray:
    """Orientation attributes of a body / geom / site / camera / inertial.  `angle_scale` = radians per angle unit of the document
    ([MJ] <compiler angle>, default degree), `eulerseq` its <compiler eulerseq>: the compiler instance's, passed in (no module state)."""
    if "quat" in attrib:
        return quat_norm(_floats(attrib["quat"]))
    if "euler" in attrib:
        return euler2quat([a * angle_scale for a in _floats(attrib["euler"])], eulerseq)
    if "zaxis" in attrib:
        return zaxis2quat(_floats(attrib["zaxis"]))
    if "axisangle" in attrib:
        v = _floats(attrib["axisangle"])
        return axisangle2quat(v[:3], v[3] * angle_scale)
    if "xyaxes" in attrib:
        v = np.array(_floats(attrib["xyaxes"]))
        x = v[:3] / np.linalg.norm(v[:3])
        y = v[3:] - np.dot(v[3:], x) * x
        y = y / np.linalg.norm(y)
        return mat2quat(np.stack([x, y, np.cross(x, y)], axis=1))
    return np.array([1.0, 0, 0, 0])


def _expand_includes(root: ET.Element, base_dir: str) -> List[Tuple[ET.Element, str]]:
    """Return [(mujoco-root, directory)] in document order with includes expanded depth-first."""
    out = []
    own = ET.Element("mujoco", root.attrib)
    pending = []
    for ch in root:
        if ch.tag == "include":
            if len(own):
                pending.append((own, base_dir))
                own = ET.Element("mujoco", root.attrib)
            p = ch.get("file")
            p = p if os.path.isabs(p) else os.path.join(base_dir, p)
            sub = ET.parse(p).getroot()
            pending.extend(_expand_includes(sub, os.path.dirname(os.path.abspath(p))))
        else:
            own.append(ch)
    if len(own):
        pending.append((own, base_dir))
    out.extend(pending)
    return out


class MjcfCompiler:
    def __init__(self):
        self.defaults = _Defaults()
        self.meshes: Dict[str, dict] = {}
        self.materials: Dict[str, np.ndarray] = {}
        self.bodies: List[_Body] = [_Body("world", None, [0, 0, 0], np.array([1.0, 0, 0, 0]))]
        self.excludes: List[Tuple[str, str]] = []
        self.tendons: List[dict] = []
        self.equalities: List[dict] = []
        self.actuators: List[dict] = []
        self.sensors: List[dict] = []
        self.keys: List[dict] = []
        self.option = dict(timestep=0.002, gravity=[0, 0, -9.81], integrator="Euler", impratio=1.0, cone="pyramidal",
                           solver="Newton", iterations=100, tolerance=1e-8, multiccd=False)
        self.stat_extent: Optional[float] = None
        self.znear, self.zfar = 0.01, 50.0
        self.meshes_missing: List[str] = []
        self.angle_scale = math.pi / 180.0   # [MJ] <compiler angle> defaults to degree
        self.eulerseq = "xyz"
        self.inertia_groups = (0, 5)
        self._geom_mesh_names: List[Optional[str]] = []

    # ---- parsing
    def parse_file(self, path: str):
        root = ET.parse(path).getroot()
        self.parse_root(root, os.path.dirname(os.path.abspath(path)))

    def parse_string(self, xml: str, base_dir: str):
        self.parse_root(ET.fromstring(xml), base_dir)

    def parse_root(self, root: ET.Element, base_dir: str):
        parts = _expand_includes(root, base_dir)
        # pass 1: compiler/defaults/assets of every part (MuJoCo parses defaults before bodies of the same file;
        # class names are global so doing all of them first is equivalent for non-conflicting files)
        for part, d in parts:
            assetdir = d
            for ch in part:
                if ch.tag == "compiler":
                    # [MJ] compiler settings are global (includes are textual): the last `angle` / `eulerseq` given wins
                    if "angle" in ch.attrib:
                        self.angle_scale = 1.0 if ch.get("angle") == "radian" else math.pi / 180.0
                    if "eulerseq" in ch.attrib:
                        self.eulerseq = ch.get("eulerseq")
                    if "inertiagrouprange" in ch.attrib:   # [MJ] only geoms of these groups give their bodies mass (robosuite: "0 0")
                        self.inertia_groups = tuple(int(v) for v in ch.get("inertiagrouprange").split())
                    assetdir = os.path.join(d, ch.get("assetdir", ch.get("meshdir", "")))
            part.set("_assetdir", assetdir)
            for ch in part:
                if ch.tag == "default":
                    self.defaults.parse(ch)
        for part, d in parts:
            for ch in part:
                if ch.tag == "asset":
                    self._parse_assets(ch, part.get("_assetdir"))
                elif ch.tag == "option":
                    for k in ("timestep", "impratio", "tolerance"):
                        if k in ch.attrib:
                            self.option[k] = float(ch.get(k))
                    if "iterations" in ch.attrib:
                        self.option["iterations"] = int(ch.get("iterations"))
                    for k in ("integrator", "cone", "solver"):
                        if k in ch.attrib:
                            self.option[k] = ch.get(k)
                    if "gravity" in ch.attrib:
                        self.option["gravity"] = _floats(ch.get("gravity"))
                    for fl in ch.findall("flag"):
                        if fl.get("multiccd") == "enable":
                            self.option["multiccd"] = True
                elif ch.tag == "statistic":
                    if "extent" in ch.attrib:
                        self.stat_extent = float(ch.get("extent"))
                elif ch.tag == "visual":
                    for mp in ch.findall("map"):
                        self.znear = float(mp.get("znear", self.znear))
                        self.zfar = float(mp.get("zfar", self.zfar))
        for part, d in parts:
            for ch in part:
                if ch.tag == "worldbody":
                    self._parse_body_children(ch, self.bodies[0], None)
                elif ch.tag == "contact":
                    for ex in ch.findall("exclude"):
                        self.excludes.append((ex.get("body1"), ex.get("body2")))
                elif ch.tag == "tendon":
                    for t in ch.findall("fixed"):
                        self.tendons.append(dict(name=t.get("name"), joints=[(j.get("joint"), float(j.get("coef", "1")))
                                                                             for j in t.findall("joint")]))
                elif ch.tag == "equality":
                    for e in ch:
                        if e.tag != "joint":
                            raise ValueError(f"equality <{e.tag}> is not supported: this build restates <joint> equalities only "
                                             f"(connect / weld / tendon / flex rows are not on the Stretch path)")
                        a = dict(_EQ_DEF)
                        a.update(self.defaults.get(e.get("class"), "equality"))
                        a.update(e.attrib)
                        self.equalities.append(a)
                elif ch.tag == "actuator":
                    for a in ch:
                        if a.tag not in _ACT_TAGS:
                            raise ValueError(f"unsupported actuator {a.tag}")
                        st = _apply_actuator(a.tag, a.attrib, self.defaults.get(a.get("class"), "actuator"))
                        st.update(name=a.get("name"), joint=a.get("joint"), tendon=a.get("tendon"))
                        self.actuators.append(st)
                elif ch.tag == "sensor":
                    for s in ch:
                        self.sensors.append(dict(type=s.tag, **s.attrib))
                elif ch.tag == "keyframe":
                    for k in ch.findall("key"):
                        self.keys.append(dict(k.attrib))

    def _parse_assets(self, elem, assetdir):
        for ch in elem:
            if ch.tag == "mesh":
                f = ch.get("file")
                if f is None:
                    # [MJ] inline mesh: vertex="x y z ..." (+ optional face="i j k ..."); without faces MuJoCo takes the convex hull
                    if ch.get("vertex") is None or ch.get("name") is None:
                        raise ValueError("<mesh> needs a file, or a name and inline vertex data")
                    v = np.array(_floats(ch.get("vertex")), float).reshape(-1, 3)
                    fc = np.array([int(x) for x in ch.get("face").split()], np.int64).reshape(-1, 3) if ch.get("face") else None
                    self.meshes[ch.get("name")] = dict(path=None, scale=_floats(ch.get("scale", "1 1 1")), inline=(v, fc))
                    continue
                name = ch.get("name") or os.path.splitext(os.path.basename(f))[0]
                self.meshes[name] = dict(path=os.path.join(assetdir, f), scale=_floats(ch.get("scale", "1 1 1")))
            elif ch.tag == "material":
                self.materials[ch.get("name")] = np.array(_floats(ch.get("rgba", "1 1 1 1")))

    def _parse_body_children(self, elem, body: _Body, childclass, frame=None):
        """frame = (pos, quat) of an enclosing <replicate>/<frame>, applied to direct children."""
        fpos, fquat = frame if frame is not None else (np.zeros(3), np.array([1.0, 0, 0, 0]))

        def place(attrib):
            p = np.array(_floats(attrib.get("pos", "0 0 0")))
            q = _orient(attrib, self.angle_scale, self.eulerseq)
            return fpos + quat2mat(fquat) @ p, quat_norm(quat_mul(fquat, q))

        for ch in elem:
            cls = ch.get("class", childclass)
            if ch.tag == "body":
                p, q = place(ch.attrib)
                b = _Body(ch.get("name"), body, p, q, float(ch.get("gravcomp", "0")))
                self.bodies.append(b)
                self._parse_body_children(ch, b, ch.get("childclass", childclass))
            elif ch.tag in ("joint", "freejoint"):
                a = dict(_JOINT_DEF)
                if ch.tag == "freejoint":
                    a["type"] = "free"
                else:
                    a.update(self.defaults.get(cls, "joint"))
                a.update(ch.attrib)
                body.joints.append(a)
            elif ch.tag == "geom":
                a = dict(_GEOM_DEF)
                a.update(self.defaults.get(cls, "geom"))
                a.update(ch.attrib)
                if "fromto" in a:   # [MJ] capsule / cylinder / box / ellipsoid between two points: centre, z axis and half length from them
                    ft = np.array(_floats(a["fromto"]))
                    vec = ft[0:3] - ft[3:6]
                    sz = _floats(a["size"])
                    half = 0.5 * float(np.linalg.norm(vec))
                    if a["type"] in ("capsule", "cylinder"):
                        a["size"] = f"{sz[0]} {half} 0"
                    else:
                        if len(sz) < 2:   # [MJ] a box / ellipsoid given by fromto needs its two cross-section half sizes
                            raise ValueError(f"geom '{a.get('name', '?')}': fromto {a['type']} needs two size values (x and y half sizes), got {len(sz)}")
                        a["size"] = f"{sz[0]} {sz[1]} {half}"
                    a = {k: v for k, v in a.items() if k not in ("quat", "euler", "axisangle", "xyaxes")}
                    a["pos"] = " ".join(repr(float(x)) for x in 0.5 * (ft[0:3] + ft[3:6]))
                    a["zaxis"] = " ".join(repr(float(x)) for x in vec)
                a["_pos"], a["_quat"] = place({k: a[k] for k in ("pos", "quat", "euler", "zaxis", "axisangle", "xyaxes") if k in a})
                body.geoms.append(a)
            elif ch.tag == "site":
                a = dict(_SITE_DEF)
                a.update(self.defaults.get(cls, "site"))
                a.update(ch.attrib)
                a["_pos"], a["_quat"] = place(a)
                if frame is not None and "_suffix" in elem.attrib and "name" in a:
                    a["name"] = a["name"] + elem.get("_suffix")
                body.sites.append(a)
            elif ch.tag == "camera":
                a = dict(_CAM_DEF)
                a.update(self.defaults.get(cls, "camera"))
                a.update(ch.attrib)
                a["_pos"], a["_quat"] = place(a)
                body.cams.append(a)
            elif ch.tag == "replicate":
                count = int(ch.get("count"))
                dq = _orient(ch.attrib, self.angle_scale, self.eulerseq)
                dp = np.array(_floats(ch.get("offset", "0 0 0")))
                width = len(str(count - 1))
                p, q = fpos.copy(), fquat.copy()
                for i in range(count):
                    ch.set("_suffix", str(i).zfill(width))
                    self._parse_body_children(ch, body, childclass, frame=(p.copy(), q.copy()))
                    p = p + quat2mat(q) @ dp
                    q = quat_norm(quat_mul(q, dq))
            elif ch.tag == "inertial":
                # [MJ] explicit body inertia: replaces what the geoms would give (compiler inertiafromgeom="auto")
                body.inertial = dict(pos=np.array(_floats(ch.get("pos", "0 0 0"))), quat=_orient(ch.attrib, self.angle_scale, self.eulerseq), mass=float(ch.get("mass")),
                                     diag=_floats(ch.get("diaginertia")) if "diaginertia" in ch.attrib else None,
                                     full=_floats(ch.get("fullinertia")) if "fullinertia" in ch.attrib else None)
            elif ch.tag == "light":
                pass
            else:
                raise ValueError(f"unsupported element <{ch.tag}> in body")

    def _lidar_static(self, m, mesh_cache, G):
        sites = list(m["sensor_lidar_site"])
        return _lidar_static_impl(m, mesh_cache, self._geom_mesh_names, sites)

    # ---- compile
    def compile(self) -> Dict[str, np.ndarray]:
        bodies = self.bodies  # already depth-first document order, parents before children
        for i, b in enumerate(bodies):
            b.id = i
        nbody = len(bodies)
        bid = {b.name: b.id for b in bodies if b.name}

        mesh_cache: Dict[str, dict] = {}

        def mesh_data(name):
            if name in mesh_cache:
                return mesh_cache[name]
            m = self.meshes[name]
            if m.get("inline") is not None:
                v, f = m["inline"]
                if f is None:   # hull faces, wound outwards
                    from scipy.spatial import ConvexHull

                    h = ConvexHull(v)
                    f = h.simplices.copy()
                    cen = v[h.vertices].mean(0)
                    for k in range(len(f)):
                        a, b, c = v[f[k]]
                        if np.dot(np.cross(b - a, c - a), a - cen) < 0:
                            f[k] = f[k][::-1]
            elif not os.path.exists(m["path"]):
                mesh_cache[name] = None
                self.meshes_missing.append(name)
                return None
            else:
                v, f = load_mesh(m["path"])
            v = v * np.array(m["scale"])
            if np.prod(m["scale"]) < 0:
                f = f[:, ::-1]
            mesh_cache[name] = dict(v=v, f=f)
            return mesh_cache[name]

        # ---------------- joints / dofs
        jnt_type, jnt_qposadr, jnt_dofadr, jnt_bodyid, jnt_pos, jnt_axis = [], [], [], [], [], []
        jnt_stiffness, jnt_range, jnt_limited, jnt_margin, jnt_solref, jnt_solimp, jnt_names = [], [], [], [], [], [], []
        dof_bodyid, dof_jntid, dof_parentid, dof_armature, dof_damping, dof_frictionloss = [], [], [], [], [], []
        dof_solref, dof_solimp = [], []
        qpos0, qpos_spring = [], []
        body_jntadr, body_jntnum, body_dofadr, body_dofnum = [], [], [], []
        body_lastdof = [-1] * nbody
        for b in bodies:
            body_jntadr.append(len(jnt_type) if b.joints else -1)
            body_jntnum.append(len(b.joints))
            body_dofadr.append(len(dof_bodyid) if b.joints else -1)
            last = body_lastdof[b.parent.id] if b.parent is not None else -1
            nd0 = len(dof_bodyid)
            for j in b.joints:
                t = {"free": JNT_FREE, "ball": JNT_BALL, "slide": JNT_SLIDE, "hinge": JNT_HINGE}[j["type"]]
                if t == JNT_BALL:
                    raise ValueError("ball joints not supported")
                jid = len(jnt_type)
                jnt_names.append(j.get("name", f"joint{jid}"))
                jnt_type.append(t)
                jnt_qposadr.append(len(qpos0))
                jnt_dofadr.append(len(dof_bodyid))
                jnt_bodyid.append(b.id)
                jnt_pos.append(_floats(j["pos"], 3))
                ax = np.array(_floats(j["axis"], 3))
                jnt_axis.append(ax / max(np.linalg.norm(ax), MINVAL))
                jnt_stiffness.append(float(j["stiffness"]))
                asc = self.angle_scale if t == JNT_HINGE else 1.0   # [MJ] hinge range / ref / springref are in the compiler's angle unit
                rng = [asc * x for x in _floats(j["range"], 2)] if "range" in j else [0.0, 0.0]
                lim = j.get("limited", "auto")
                jnt_limited.append(int((lim == "true") or (lim == "auto" and "range" in j)))
                jnt_range.append(rng)
                jnt_margin.append(float(j["margin"]))
                jnt_solref.append(_floats(j["solreflimit"], 2))
                jnt_solimp.append(_floats(j["solimplimit"], 5))
                nd = 6 if t == JNT_FREE else 1
                for k in range(nd):
                    dof_bodyid.append(b.id)
                    dof_jntid.append(jid)
                    dof_parentid.append(last)
                    last = len(dof_bodyid) - 1
                    dof_armature.append(float(j["armature"]))
                    dof_damping.append(float(j["damping"]))
                    dof_frictionloss.append(float(j["frictionloss"]))
                    dof_solref.append(_floats(j["solreffriction"], 2))
                    dof_solimp.append(_floats(j["solimpfriction"], 5))
                if t == JNT_FREE:
                    qpos0.extend(list(b.pos) + list(b.quat))
                    qpos_spring.extend(list(b.pos) + list(b.quat))
                else:
                    qpos0.append(asc * float(j["ref"]))
                    qpos_spring.append(asc * float(j["springref"]))
            body_dofnum.append(len(dof_bodyid) - nd0)
            body_lastdof[b.id] = last
        nq, nv, njnt = len(qpos0), len(dof_bodyid), len(jnt_type)
        jid_by_name = {n: i for i, n in enumerate(jnt_names)}

        body_parentid = [0 if b.parent is None else b.parent.id for b in bodies]
        body_weldid = [0] * nbody
        body_rootid = [0] * nbody
        for b in bodies[1:]:
            body_weldid[b.id] = b.id if b.joints else body_weldid[b.parent.id]
            body_rootid[b.id] = b.id if b.parent.id == 0 else body_rootid[b.parent.id]

        # ---------------- geoms + inertia
        G = dict(type=[], bodyid=[], pos=[], quat=[], size=[], contype=[], conaffinity=[], condim=[], group=[],
                 priority=[], friction=[], solmix=[], solref=[], solimp=[], margin=[], gap=[], rgba=[], rbound=[],
                 center=[], aabb=[], hulladr=[], hullnum=[], meshid=[], name=[], ccenter=[])
        hull_verts: List[np.ndarray] = []
        hull_cache: Dict[str, Tuple[int, int]] = {}
        mesh_names: List[str] = []
        body_mass = np.zeros(nbody)
        body_ipos = np.zeros((nbody, 3))
        body_iquat = np.tile([1.0, 0, 0, 0], (nbody, 1))
        body_inertia = np.zeros((nbody, 3))
        nhv = 0
        for b in bodies:
            parts = []  # (mass, com_in_body, inertia_about_com_in_body_axes)
            for g in b.geoms:
                t = GEOM_TYPES[g["type"]]
                size = _floats(g["size"], 3)
                gpos, gquat = g["_pos"], g["_quat"]
                R = quat2mat(gquat)
                md = None
                if t == GEOM_MESH:
                    md = mesh_data(g["mesh"])
                    if md is None:
                        # missing visual-only asset (.MISSING_LARGE_BLOBS); must not carry mass or collide
                        if float(g.get("mass", "-1")) > 0 or int(g["contype"]) or int(g["conaffinity"]):
                            raise FileNotFoundError(self.meshes[g["mesh"]]["path"])
                        continue
                # mass properties
                mass_attr = float(g["mass"]) if "mass" in g else None
                dens = float(g["density"])
                shell = g.get("shellinertia", "false") == "true"
                vol, com, I = 0.0, np.zeros(3), np.zeros((3, 3))
                if t == GEOM_MESH:
                    if (mass_attr is None and dens > 0) or (mass_attr is not None and mass_attr > 0):
                        vol, com, I = mesh_shell_props(md["v"], md["f"]) if shell else mesh_volume_props(md["v"], md["f"])
                elif t == GEOM_SPHERE:
                    r = size[0]
                    vol = 4.0 / 3 * math.pi * r ** 3
                    I = np.eye(3) * (0.4 * vol * r * r)
                elif t == GEOM_BOX:
                    vol = 8 * size[0] * size[1] * size[2]
                    I = np.diag([vol / 3 * (size[1] ** 2 + size[2] ** 2), vol / 3 * (size[0] ** 2 + size[2] ** 2),
                                 vol / 3 * (size[0] ** 2 + size[1] ** 2)])
                elif t == GEOM_CYLINDER:
                    r, h = size[0], size[1]
                    vol = math.pi * r * r * 2 * h
                    ixx = vol * (3 * r * r + (2 * h) ** 2) / 12
                    I = np.diag([ixx, ixx, vol * r * r / 2])
                elif t == GEOM_CAPSULE:
                    r, h = size[0], size[1]
                    vc, vs = math.pi * r * r * 2 * h, 4.0 / 3 * math.pi * r ** 3
                    vol = vc + vs
                    izz = vc * r * r / 2 + vs * 0.4 * r * r
                    ixx = vc * (3 * r * r + (2 * h) ** 2) / 12 + vs * (0.4 * r * r + h * h + 0.75 * h * r)
                    I = np.diag([ixx, ixx, izz])
                elif t == GEOM_ELLIPSOID:
                    vol = 4.0 / 3 * math.pi * size[0] * size[1] * size[2]
                    I = np.diag([vol / 5 * (size[1] ** 2 + size[2] ** 2), vol / 5 * (size[0] ** 2 + size[2] ** 2),
                                 vol / 5 * (size[0] ** 2 + size[1] ** 2)])
                if t != GEOM_PLANE and vol > 0 and self.inertia_groups[0] <= int(g["group"]) <= self.inertia_groups[1]:
                    m = mass_attr if mass_attr is not None else dens * vol
                    if m > 0:
                        sc = m / vol
                        parts.append((m, gpos + R @ com, R @ (I * sc) @ R.T))
                # collision / ray shape data
                collides = bool(int(g["contype"]) or int(g["conaffinity"]))
                if t == GEOM_MESH and not collides:
                    adr, num, meshid = -1, 0, -1
                    lo, hi = md["v"].min(0), md["v"].max(0)
                    cen = 0.5 * (lo + hi)
                    rb = float(np.linalg.norm(md["v"] - cen, axis=1).max())
                    aabb = np.concatenate([cen, 0.5 * (hi - lo)])
                elif t == GEOM_MESH:
                    key = g["mesh"]
                    if key not in hull_cache:
                        hv = convex_hull_vertices(md["v"])
                        hull_cache[key] = (nhv, len(hv))
                        hull_verts.append(hv)
                        nhv += len(hv)
                        mesh_names.append(key)
                    adr, num = hull_cache[key]
                    hv = hull_verts[mesh_names.index(key)]
                    lo, hi = md["v"].min(0), md["v"].max(0)
                    cen = 0.5 * (lo + hi)
                    rb = float(np.linalg.norm(md["v"] - cen, axis=1).max())
                    aabb = np.concatenate([cen, 0.5 * (hi - lo)])
                    meshid = mesh_names.index(key)
                else:
                    adr, num, cen, meshid = -1, 0, np.zeros(3), -1
                    if t == GEOM_PLANE:
                        rb, aabb = 0.0, np.zeros(6)
                    elif t == GEOM_SPHERE:
                        rb, aabb = size[0], np.array([0, 0, 0, size[0], size[0], size[0]])
                    elif t == GEOM_BOX:
                        rb, aabb = float(np.linalg.norm(size)), np.array([0, 0, 0] + size)
                    elif t == GEOM_CYLINDER:
                        rb, aabb = math.hypot(size[0], size[1]), np.array([0, 0, 0, size[0], size[0], size[1]])
                    elif t == GEOM_CAPSULE:
                        rb, aabb = size[0] + size[1], np.array([0, 0, 0, size[0], size[0], size[0] + size[1]])
                    elif t == GEOM_ELLIPSOID:
                        rb, aabb = max(size), np.array([0, 0, 0] + size)
                    else:
                        raise ValueError("unsupported geom type")
                rgba = np.array(_floats(g["rgba"], 4))
                if "material" in g and g["rgba"] == _GEOM_DEF["rgba"]:
                    rgba = self.materials.get(g["material"], rgba)
                fr = _floats(g["friction"], 3)
                G["type"].append(t); G["bodyid"].append(b.id); G["pos"].append(gpos); G["quat"].append(gquat)
                G["size"].append(size); G["contype"].append(int(g["contype"])); G["conaffinity"].append(int(g["conaffinity"]))
                G["condim"].append(int(g["condim"])); G["group"].append(int(g["group"])); G["priority"].append(int(g["priority"]))
                G["friction"].append(fr); G["solmix"].append(float(g["solmix"])); G["solref"].append(_floats(g["solref"], 2))
                G["solimp"].append(_floats(g["solimp"], 5)); G["margin"].append(float(g["margin"])); G["gap"].append(float(g["gap"]))
                G["rgba"].append(rgba); G["rbound"].append(rb); G["center"].append(cen); G["aabb"].append(aabb)
                G["hulladr"].append(adr); G["hullnum"].append(num); G["meshid"].append(meshid)
                # an interior point of the convex shape (MPR portal centre): hull-vertex mean, primitives: frame origin
                G["ccenter"].append(hull_verts[mesh_names.index(g["mesh"])].mean(0) if (t == GEOM_MESH and num > 0) else np.zeros(3))
                G["name"].append(g.get("name", ""))
                self._geom_mesh_names.append(g.get("mesh") if t == GEOM_MESH else None)
            if b.inertial is not None:
                it = b.inertial
                if it["full"] is not None:   # xx yy zz xy xz yz in the body frame: principal axes by eigen-decomposition
                    f = it["full"]
                    I = np.array([[f[0], f[3], f[4]], [f[3], f[1], f[5]], [f[4], f[5], f[2]]])
                    w, V = np.linalg.eigh(I)
                    order = np.argsort(-w)
                    w, V = w[order], V[:, order]
                    if np.linalg.det(V) < 0:
                        V[:, 2] = -V[:, 2]
                    iq = mat2quat(V)
                else:
                    w, iq = np.array(it["diag"], float), it["quat"]
                body_mass[b.id], body_ipos[b.id], body_iquat[b.id], body_inertia[b.id] = it["mass"], it["pos"], iq, w
            elif parts:
                M = sum(p[0] for p in parts)
                c = sum(p[0] * p[1] for p in parts) / M
                I = np.zeros((3, 3))
                for m, pc, Ic in parts:
                    d = pc - c
                    I += Ic + m * (np.dot(d, d) * np.eye(3) - np.outer(d, d))
                w, V = np.linalg.eigh(I)
                order = np.argsort(-w)
                w, V = w[order], V[:, order]
                if np.linalg.det(V) < 0:
                    V[:, 2] = -V[:, 2]
                body_mass[b.id], body_ipos[b.id], body_iquat[b.id], body_inertia[b.id] = M, c, mat2quat(V), w
            elif b.joints:
                raise ValueError(f"moving body {b.name} has no mass")
        ngeom = len(G["type"])

        # ---------------- sites, cameras
        site_bodyid, site_pos, site_quat, site_names = [], [], [], []
        cam_bodyid, cam_pos, cam_quat, cam_fovy, cam_names = [], [], [], [], []
        for b in bodies:
            for s in b.sites:
                site_bodyid.append(b.id); site_pos.append(s["_pos"]); site_quat.append(s["_quat"]); site_names.append(s.get("name", ""))
            for c in b.cams:
                cam_bodyid.append(b.id); cam_pos.append(c["_pos"]); cam_quat.append(c["_quat"])
                cam_fovy.append(float(c["fovy"])); cam_names.append(c.get("name", ""))

        # ---------------- tendons (fixed), equalities, actuators, keys
        ten_names = [t["name"] for t in self.tendons]
        tendon_adr, tendon_num, wrap_objid, wrap_prm = [], [], [], []
        for t in self.tendons:
            tendon_adr.append(len(wrap_objid)); tendon_num.append(len(t["joints"]))
            for jn, coef in t["joints"]:
                wrap_objid.append(jid_by_name[jn]); wrap_prm.append(coef)
        eq_obj1, eq_obj2, eq_data, eq_solref, eq_solimp, eq_active = [], [], [], [], [], []
        for e in self.equalities:
            eq_obj1.append(jid_by_name[e["joint1"]])
            eq_obj2.append(jid_by_name[e["joint2"]] if "joint2" in e else -1)
            eq_data.append(_floats(e.get("polycoef", "0 1 0 0 0"), 5))
            eq_solref.append(_floats(e["solref"], 2)); eq_solimp.append(_floats(e["solimp"], 5))
            eq_active.append(int(e.get("active", "true") == "true"))
        nu = len(self.actuators)
        A = dict(trntype=[], trnid=[], gear=[], gainprm=[], biasprm=[], ctrllimited=[], ctrlrange=[], forcelimited=[],
                 forcerange=[], biastype=[], name=[])
        for a in self.actuators:
            if a["joint"] is not None:
                A["trntype"].append(0); A["trnid"].append(jid_by_name[a["joint"]])
                if jnt_type[jid_by_name[a["joint"]]] not in (JNT_SLIDE, JNT_HINGE):
                    raise ValueError("actuators on free/ball joints are not supported")
            elif a["tendon"] is not None:
                A["trntype"].append(1); A["trnid"].append(ten_names.index(a["tendon"]))
            else:
                raise ValueError("actuator needs joint or tendon transmission")
            A["gear"].append(a["gear"][0]); A["gainprm"].append(a["gainprm"]); A["biasprm"].append(a["biasprm"])
            cl = a["ctrllimited"]; fl = a["forcelimited"]
            A["ctrllimited"].append(int(cl == "true" or (cl == "auto" and a["ctrlrange"] is not None)))
            A["forcelimited"].append(int(fl == "true" or (fl == "auto" and a["forcerange"] is not None)))
            A["ctrlrange"].append(a["ctrlrange"] or [0.0, 0.0]); A["forcerange"].append(a["forcerange"] or [0.0, 0.0])
            A["biastype"].append({"none": 0, "affine": 1}[a["biastype"]]); A["name"].append(a["name"])
        key_ctrl = np.zeros((len(self.keys), max(nu, 1)))
        for i, k in enumerate(self.keys):
            if "ctrl" in k:
                key_ctrl[i, :nu] = _floats(k["ctrl"], nu)[:nu]
            if "qpos" in k:
                raise ValueError("keyframe qpos not supported")

        # ---------------- static collision pair table (body-pair major, then geom order; MuJoCo filters [MJ] B.3)
        excl = set()
        for b1, b2 in self.excludes:
            excl.add((min(bid[b1], bid[b2]), max(bid[b1], bid[b2])))
        pairs = []
        gb = G["bodyid"]
        for g1 in range(ngeom):
            for g2 in range(g1 + 1, ngeom):
                b1, b2 = gb[g1], gb[g2]
                if b1 == b2:
                    continue
                if not ((G["contype"][g1] & G["conaffinity"][g2]) or (G["contype"][g2] & G["conaffinity"][g1])):
                    continue
                w1, w2 = body_weldid[b1], body_weldid[b2]
                if w1 == w2:
                    continue
                p1, p2 = body_weldid[body_parentid[w1]], body_weldid[body_parentid[w2]]
                if w1 and w2 and (w1 == p2 or w2 == p1):
                    continue
                if (min(b1, b2), max(b1, b2)) in excl:
                    continue
                if G["type"][g1] == GEOM_PLANE and G["type"][g2] == GEOM_PLANE:
                    continue
                a, b_ = (g1, g2) if G["type"][g1] <= G["type"][g2] else (g2, g1)
                pairs.append((min(b1, b2), max(b1, b2), g1, g2, a, b_))
        pairs.sort(key=lambda p: (p[0], p[1], p[2], p[3]))
        P = dict(geom1=[], geom2=[], condim=[], friction=[], solref=[], solimp=[], margin=[], gap=[])
        for _, _, _, _, a, b_ in pairs:
            pa, pb = G["priority"][a], G["priority"][b_]
            if pa != pb:
                w = a if pa > pb else b_
                condim, fr, sr, si = G["condim"][w], G["friction"][w], G["solref"][w], G["solimp"][w]
            else:
                condim = max(G["condim"][a], G["condim"][b_])
                fr = [max(x, y) for x, y in zip(G["friction"][a], G["friction"][b_])]
                ma, mb = G["solmix"][a], G["solmix"][b_]
                mix = 0.5 if (ma < MINVAL and mb < MINVAL) else ma / (ma + mb)
                sra, srb = G["solref"][a], G["solref"][b_]
                if sra[0] > 0 and srb[0] > 0:
                    sr = [mix * x + (1 - mix) * y for x, y in zip(sra, srb)]
                else:
                    sr = [min(x, y) for x, y in zip(sra, srb)]
                si = [mix * x + (1 - mix) * y for x, y in zip(G["solimp"][a], G["solimp"][b_])]
            P["geom1"].append(a); P["geom2"].append(b_); P["condim"].append(condim)
            P["friction"].append([fr[0], fr[0], fr[1], fr[2], fr[2]]); P["solref"].append(sr); P["solimp"].append(si)
            P["margin"].append(max(G["margin"][a], G["margin"][b_])); P["gap"].append(max(G["gap"][a], G["gap"][b_]))

        # ---------------- sensors (the three the reference reads: sensor_manager.py:65-89)
        site_id = {n: i for i, n in enumerate(site_names) if n}
        imu_site, lidar_sites, lidar_cutoff = -1, [], 0.0
        for s in self.sensors:
            if s["type"] in ("gyro", "accelerometer"):
                imu_site = site_id[s["site"]]
            elif s["type"] == "rangefinder":
                base = s["site"]
                lidar_sites = [i for i, n in enumerate(site_names) if n.startswith(base) and n[len(base):].isdigit()]
                lidar_cutoff = float(s.get("cutoff", "0"))

        o = self.option
        m: Dict[str, np.ndarray] = {}
        m["dims"] = np.array([nq, nv, nu, nbody, njnt, ngeom, len(site_bodyid), len(cam_bodyid), len(eq_obj1),
                              len(tendon_adr), len(wrap_objid), len(self.keys), len(P["geom1"]), nhv], np.int32)
        m["opt_timestep"] = np.array([o["timestep"]]); m["opt_gravity"] = np.array(o["gravity"], float)
        m["opt_impratio"] = np.array([o["impratio"]]); m["opt_tolerance"] = np.array([o["tolerance"]])
        m["opt_iterations"] = np.array([o["iterations"]], np.int32)
        m["opt_cone"] = np.array([{"pyramidal": 0, "elliptic": 1}[o["cone"]]], np.int32)
        m["opt_integrator"] = np.array([{"Euler": 0, "RK4": 1, "implicit": 2, "implicitfast": 3}[o["integrator"]]], np.int32)
        m["opt_solver"] = np.array([{"PGS": 0, "CG": 1, "Newton": 2}[o["solver"]]], np.int32)
        m["body_parentid"] = np.array(body_parentid, np.int32); m["body_weldid"] = np.array(body_weldid, np.int32)
        m["body_rootid"] = np.array(body_rootid, np.int32)
        m["body_pos"] = np.array([b.pos for b in bodies]); m["body_quat"] = np.array([b.quat for b in bodies])
        m["body_ipos"], m["body_iquat"], m["body_mass"], m["body_inertia"] = body_ipos, body_iquat, body_mass, body_inertia
        m["body_gravcomp"] = np.array([b.gravcomp for b in bodies])
        m["body_jntadr"] = np.array(body_jntadr, np.int32); m["body_jntnum"] = np.array(body_jntnum, np.int32)
        m["body_dofadr"] = np.array(body_dofadr, np.int32); m["body_dofnum"] = np.array(body_dofnum, np.int32)
        m["jnt_type"] = np.array(jnt_type, np.int32); m["jnt_qposadr"] = np.array(jnt_qposadr, np.int32)
        m["jnt_dofadr"] = np.array(jnt_dofadr, np.int32); m["jnt_bodyid"] = np.array(jnt_bodyid, np.int32)
        m["jnt_pos"] = np.array(jnt_pos, float).reshape(-1, 3); m["jnt_axis"] = np.array(jnt_axis, float).reshape(-1, 3)
        m["jnt_stiffness"] = np.array(jnt_stiffness); m["jnt_range"] = np.array(jnt_range, float).reshape(-1, 2)
        m["jnt_limited"] = np.array(jnt_limited, np.int32); m["jnt_margin"] = np.array(jnt_margin)
        m["jnt_solref"] = np.array(jnt_solref, float).reshape(-1, 2); m["jnt_solimp"] = np.array(jnt_solimp, float).reshape(-1, 5)
        m["dof_bodyid"] = np.array(dof_bodyid, np.int32); m["dof_jntid"] = np.array(dof_jntid, np.int32)
        m["dof_parentid"] = np.array(dof_parentid, np.int32); m["dof_armature"] = np.array(dof_armature)
        m["dof_damping"] = np.array(dof_damping); m["dof_frictionloss"] = np.array(dof_frictionloss)
        m["dof_solref"] = np.array(dof_solref, float).reshape(-1, 2); m["dof_solimp"] = np.array(dof_solimp, float).reshape(-1, 5)
        m["qpos0"] = np.array(qpos0); m["qpos_spring"] = np.array(qpos_spring)
        for k in ("type", "bodyid", "contype", "conaffinity", "condim", "group", "priority", "hulladr", "hullnum", "meshid"):
            m["geom_" + k] = np.array(G[k], np.int32)
        m["geom_pos"] = np.array(G["pos"], float).reshape(-1, 3); m["geom_quat"] = np.array(G["quat"], float).reshape(-1, 4)
        m["geom_size"] = np.array(G["size"], float).reshape(-1, 3); m["geom_rgba"] = np.array(G["rgba"], float).reshape(-1, 4)
        m["geom_rbound"] = np.array(G["rbound"]); m["geom_center"] = np.array(G["center"], float).reshape(-1, 3)
        m["geom_aabb"] = np.array(G["aabb"], float).reshape(-1, 6)
        m["geom_ccenter"] = np.array(G["ccenter"], float).reshape(-1, 3)
        m["geom_friction"] = np.array(G["friction"], float).reshape(-1, 3)
        m["hull_vert"] = np.concatenate(hull_verts, 0) if hull_verts else np.zeros((0, 3))
        m["site_bodyid"] = np.array(site_bodyid, np.int32); m["site_pos"] = np.array(site_pos, float).reshape(-1, 3)
        m["site_quat"] = np.array(site_quat, float).reshape(-1, 4)
        m["cam_bodyid"] = np.array(cam_bodyid, np.int32); m["cam_pos"] = np.array(cam_pos, float).reshape(-1, 3)
        m["cam_quat"] = np.array(cam_quat, float).reshape(-1, 4); m["cam_fovy"] = np.array(cam_fovy, float)
        m["tendon_adr"] = np.array(tendon_adr, np.int32); m["tendon_num"] = np.array(tendon_num, np.int32)
        m["wrap_objid"] = np.array(wrap_objid, np.int32); m["wrap_prm"] = np.array(wrap_prm, float)
        m["eq_obj1id"] = np.array(eq_obj1, np.int32); m["eq_obj2id"] = np.array(eq_obj2, np.int32)
        m["eq_data"] = np.array(eq_data, float).reshape(-1, 5); m["eq_solref"] = np.array(eq_solref, float).reshape(-1, 2)
        m["eq_solimp"] = np.array(eq_solimp, float).reshape(-1, 5); m["eq_active"] = np.array(eq_active, np.int32)
        m["actuator_trntype"] = np.array(A["trntype"], np.int32); m["actuator_trnid"] = np.array(A["trnid"], np.int32)
        m["actuator_gear"] = np.array(A["gear"], float); m["actuator_gainprm"] = np.array(A["gainprm"], float).reshape(-1, 3)
        m["actuator_biasprm"] = np.array(A["biasprm"], float).reshape(-1, 3)
        m["actuator_ctrllimited"] = np.array(A["ctrllimited"], np.int32); m["actuator_forcelimited"] = np.array(A["forcelimited"], np.int32)
        m["actuator_ctrlrange"] = np.array(A["ctrlrange"], float).reshape(-1, 2)
        m["actuator_forcerange"] = np.array(A["forcerange"], float).reshape(-1, 2)
        m["actuator_biastype"] = np.array(A["biastype"], np.int32)
        m["key_ctrl"] = key_ctrl
        m["pair_geom1"] = np.array(P["geom1"], np.int32); m["pair_geom2"] = np.array(P["geom2"], np.int32)
        m["pair_condim"] = np.array(P["condim"], np.int32); m["pair_friction"] = np.array(P["friction"], float).reshape(-1, 5)
        m["pair_solref"] = np.array(P["solref"], float).reshape(-1, 2); m["pair_solimp"] = np.array(P["solimp"], float).reshape(-1, 5)
        m["pair_margin"] = np.array(P["margin"], float); m["pair_gap"] = np.array(P["gap"], float)
        m["sensor_imu_site"] = np.array([imu_site], np.int32); m["sensor_lidar_site"] = np.array(lidar_sites, np.int32)
        m["sensor_lidar_cutoff"] = np.array([lidar_cutoff])
        # ---------------- ray-cast meshes: triangles of the mesh geoms a camera can see ([MJ] mjv_addGeoms draws geom groups
        # 0-2 by default and skips alpha 0; the collision class of stretch.xml is group 3, hidden) or a lidar ray can meet.
        rm_index: Dict[str, int] = {}
        rverts, rfaces, rvadr, rvnum, rfadr, rfnum, geom_rmeshid = [], [], [], [], [], [], []
        nrv = nrf = 0
        laser_weld = body_weldid[site_bodyid[lidar_sites[0]]] if lidar_sites else -1
        for g in range(ngeom):
            name = self._geom_mesh_names[g]
            md = mesh_cache.get(name) if name is not None else None
            camera_sees = G["group"][g] <= 2
            # [MJ] mj_ray tests every geom group; geoms welded to the laser are covered by sensor_lidar_static instead
            lidar_sees = laser_weld >= 0 and body_weldid[G["bodyid"][g]] != laser_weld
            if md is None or G["rgba"][g][3] == 0 or not (camera_sees or lidar_sees):
                geom_rmeshid.append(-1)
                continue
            if name not in rm_index:
                rm_index[name] = len(rvadr)
                # weld exactly repeated vertices (OBJ exporters repeat them per face); geometry is unchanged
                v32 = np.ascontiguousarray(np.asarray(md["v"], np.float32))
                uv, inv = np.unique(v32, axis=0, return_inverse=True)
                fw = inv.reshape(-1)[np.asarray(md["f"], np.int64)].astype(np.int32)
                rvadr.append(nrv); rvnum.append(len(uv)); rfadr.append(nrf); rfnum.append(len(fw))
                rverts.append(uv); rfaces.append(fw)
                nrv += len(uv); nrf += len(fw)
            geom_rmeshid.append(rm_index[name])
        m["geom_rmeshid"] = np.array(geom_rmeshid, np.int32).reshape(-1)
        m["rmesh_vert"] = np.concatenate(rverts, 0).astype(np.float32) if rverts else np.zeros((0, 3), np.float32)
        m["rmesh_face"] = np.concatenate(rfaces, 0).astype(np.int32) if rfaces else np.zeros((0, 3), np.int32)
        m["rmesh_vertadr"] = np.array(rvadr, np.int32); m["rmesh_vertnum"] = np.array(rvnum, np.int32)
        m["rmesh_faceadr"] = np.array(rfadr, np.int32); m["rmesh_facenum"] = np.array(rfnum, np.int32)
        extent = self.stat_extent if self.stat_extent is not None else _stat_extent(m)
        m["vis_znear_zfar_extent"] = np.array([self.znear, self.zfar, extent])
        names = dict(body=[b.name or "" for b in bodies], joint=jnt_names, geom=G["name"], site=site_names, camera=cam_names,
                     actuator=A["name"], tendon=ten_names, key=[k.get("name", "") for k in self.keys], mesh=mesh_names,
                     missing_meshes=self.meshes_missing)
        m["names_json"] = np.frombuffer(json.dumps(names).encode(), np.uint8)
        _set_const(m)
        m["sensor_lidar_static"] = self._lidar_static(m, mesh_cache, G)
        # per-body gravity-compensation mass/point and per-geom invweight0 are carried explicitly so that
        # static-body fusion (model_fuse.py) preserves them exactly
        m["body_gcmass"] = m["body_mass"] * m["body_gravcomp"]
        m["body_gcipos"] = m["body_ipos"].copy()
        m["geom_invweight0"] = m["body_invweight0"][m["geom_bodyid"]] if ngeom else np.zeros((0, 2))
        return m


# ----------------------------------------------------------------------------- constants at qpos0
def _stat_extent(m):
    """[MJ] mj_setConst / set0: bounding box at qpos0 of body frames, body coms, joint anchors, sites and geoms (padded by
    rbound; an infinite plane counts with rbound 1, a finite one with a tenth of its larger side); stat.extent is the
    longest side of that box.  (MuJoCo is not importable here: restated from memory, flagged in DESIGN.md.)"""
    xpos, xquat, xanchor, _ = _fk(m, m["qpos0"])
    pts_lo, pts_hi = [], []

    def add(p, r=0.0):
        pts_lo.append(np.asarray(p) - r); pts_hi.append(np.asarray(p) + r)
    for b in range(1, len(xpos)):
        add(xpos[b]); add(xpos[b] + quat2mat(xquat[b]) @ m["body_ipos"][b])
    for j in range(len(xanchor)):
        add(xanchor[j])
    for i, b in enumerate(m["site_bodyid"]):
        add(xpos[b] + quat2mat(xquat[b]) @ m["site_pos"][i])
    for g, b in enumerate(m["geom_bodyid"]):
        r = m["geom_rbound"][g]
        if m["geom_type"][g] == GEOM_PLANE:
            sz = m["geom_size"][g]
            r = 0.1 * max(sz[0], sz[1]) if (sz[0] > 0 or sz[1] > 0) else 1.0
        add(xpos[b] + quat2mat(xquat[b]) @ m["geom_pos"][g], r)
    if not pts_lo:
        return 1.0
    lo, hi = np.min(pts_lo, axis=0), np.max(pts_hi, axis=0)
    return float(max(1e-5, np.max(hi - lo)))


def _lidar_static_impl(m, mesh_cache, G_mesh_names, lidar_sites):
    """Distance of every lidar ray to the geoms that are rigidly attached to the laser (same weld group, i.e. no joint
    in between): these hits do not depend on the state, so they are ray-cast ONCE here against the true triangle
    meshes ([MJ] mj_ray intersects mesh geoms by their triangles, in all groups, skipping the site's own body and
    geoms with alpha 0) and stored per ray.  -1 = no static hit."""
    n = len(lidar_sites)
    out = np.full(n, -1.0)
    if n == 0:
        return out
    xpos, xquat, _, _ = _fk(m, m["qpos0"])
    sb = m["site_bodyid"][lidar_sites[0]]
    weld = m["body_weldid"][sb]
    orig = np.zeros((n, 3)); dirs = np.zeros((n, 3))
    for i, sid in enumerate(lidar_sites):
        b = m["site_bodyid"][sid]
        R = quat2mat(xquat[b])
        orig[i] = xpos[b] + R @ m["site_pos"][sid]
        dirs[i] = (R @ quat2mat(m["site_quat"][sid]))[:, 2]
    best = np.full(n, np.inf)
    for g in range(len(m["geom_type"])):
        b = m["geom_bodyid"][g]
        if m["body_weldid"][b] != weld or b == sb or m["geom_rgba"][g][3] == 0:
            continue
        Rb = quat2mat(xquat[b])
        gp = xpos[b] + Rb @ m["geom_pos"][g]
        Rg = Rb @ quat2mat(m["geom_quat"][g])
        lo = (orig - gp) @ Rg       # rays in the geom frame
        ld = dirs @ Rg
        t = m["geom_type"][g]
        if t == GEOM_MESH:
            md = mesh_cache.get(G_mesh_names[g])
            if md is None:
                continue
            cen, half = m["geom_aabb"][g][:3], m["geom_aabb"][g][3:] + 1e-9
            # slab test against the mesh AABB to skip rays that cannot hit
            with np.errstate(divide="ignore", invalid="ignore"):
                t1 = (cen - half - lo) / ld; t2 = (cen + half - lo) / ld
            tmin = np.nanmax(np.minimum(t1, t2), axis=1); tmax = np.nanmin(np.maximum(t1, t2), axis=1)
            cand = np.nonzero((tmax >= np.maximum(tmin, 0)))[0]
            if len(cand) == 0:
                continue
            tri = md["v"][md["f"]]
            d = ray_triangles(lo[cand], ld[cand], tri)
            hit = d >= 0
            best[cand[hit]] = np.minimum(best[cand[hit]], d[hit])
        elif t != GEOM_PLANE:
            for i in range(n):
                x = _ray_primitive(t, m["geom_size"][g], lo[i], ld[i])
                if x >= 0:
                    best[i] = min(best[i], x)
    return np.where(np.isfinite(best), best, -1.0)


def _fk(m, qpos):
    nbody = len(m["body_parentid"])
    xpos = np.zeros((nbody, 3)); xquat = np.tile([1.0, 0, 0, 0], (nbody, 1))
    xanchor = np.zeros((len(m["jnt_type"]), 3)); xaxis = np.zeros((len(m["jnt_type"]), 3))
    for b in range(1, nbody):
        p = m["body_parentid"][b]
        if m["body_jntnum"][b] == 1 and m["jnt_type"][m["body_jntadr"][b]] == JNT_FREE:
            a = m["jnt_qposadr"][m["body_jntadr"][b]]
            pos, quat = qpos[a:a + 3].copy(), quat_norm(qpos[a + 3:a + 7])
            xanchor[m["body_jntadr"][b]] = pos; xaxis[m["body_jntadr"][b]] = [0, 0, 1]
        else:
            pos = xpos[p] + quat2mat(xquat[p]) @ m["body_pos"][b]
            quat = quat_mul(xquat[p], m["body_quat"][b])
            for j in range(m["body_jntadr"][b], m["body_jntadr"][b] + m["body_jntnum"][b]):
                R = quat2mat(quat)
                xaxis[j] = R @ m["jnt_axis"][j]
                xanchor[j] = pos + R @ m["jnt_pos"][j]
                q = qpos[m["jnt_qposadr"][j]] - m["qpos0"][m["jnt_qposadr"][j]]
                if m["jnt_type"][j] == JNT_SLIDE:
                    pos = pos + xaxis[j] * q
                else:
                    quat = quat_mul(quat, axisangle2quat(m["jnt_axis"][j], q))
                    pos = xanchor[j] - quat2mat(quat) @ m["jnt_pos"][j]
        xpos[b], xquat[b] = pos, quat_norm(quat)
    return xpos, xquat, xanchor, xaxis


def _body_jac(m, xanchor, xaxis, point, body):
    """6 x nv Jacobian (rows 0-2 translational at `point`, 3-5 rotational) of `body`.  [MJ] mj_jac."""
    nv = len(m["dof_bodyid"])
    J = np.zeros((6, nv))
    b = body
    while b > 0 and m["body_dofnum"][b] == 0:
        b = m["body_parentid"][b]
    if b == 0:
        return J
    d = m["body_dofadr"][b] + m["body_dofnum"][b] - 1
    while d >= 0:
        j = m["dof_jntid"][d]
        t = m["jnt_type"][j]
        if t == JNT_FREE:
            k = d - m["jnt_dofadr"][j]
            if k < 3:
                J[k, d] = 1.0
            else:
                ax = m["_free_R"][j][:, k - 3]
                J[3:, d] = ax
                J[:3, d] = np.cross(ax, point - xanchor[j])
        elif t == JNT_SLIDE:
            J[:3, d] = xaxis[j]
        else:
            J[3:, d] = xaxis[j]
            J[:3, d] = np.cross(xaxis[j], point - xanchor[j])
        d = m["dof_parentid"][d]
    return J


def mass_matrix(m, qpos):
    """Dense joint-space inertia at qpos via body Jacobians (compile-time helper, not the hot path)."""
    xpos, xquat, xanchor, xaxis = _fk(m, qpos)
    m["_free_R"] = {j: quat2mat(xquat[m["jnt_bodyid"][j]]) for j in range(len(m["jnt_type"])) if m["jnt_type"][j] == JNT_FREE}
    nv = len(m["dof_bodyid"])
    M = np.diag(m["dof_armature"].astype(float))
    jacs = {}
    for b in range(1, len(m["body_parentid"])):
        R = quat2mat(xquat[b])
        xipos = xpos[b] + R @ m["body_ipos"][b]
        J = _body_jac(m, xanchor, xaxis, xipos, b)
        jacs[b] = J
        if m["body_mass"][b] > 0:
            Ri = R @ quat2mat(m["body_iquat"][b])
            Iw = Ri @ np.diag(m["body_inertia"][b]) @ Ri.T
            M += m["body_mass"][b] * J[:3].T @ J[:3] + J[3:].T @ Iw @ J[3:]
    del m["_free_R"]
    return M, jacs


def _set_const(m):
    """dof_invweight0, body_invweight0, tendon_invweight0, stat_meaninertia at qpos0.  [MJ] mj_setConst/set0."""
    nv = len(m["dof_bodyid"])
    nbody = len(m["body_parentid"])
    M, jacs = mass_matrix(m, m["qpos0"])
    Minv = np.linalg.inv(M) if nv else np.zeros((0, 0))
    m["stat_meaninertia"] = np.array([float(np.mean(np.diag(M))) if nv else 1.0])
    dinv = np.zeros(nv)
    for j in range(len(m["jnt_type"])):
        a = m["jnt_dofadr"][j]
        if m["jnt_type"][j] == JNT_FREE:
            dinv[a:a + 3] = np.mean(np.diag(Minv)[a:a + 3])
            dinv[a + 3:a + 6] = np.mean(np.diag(Minv)[a + 3:a + 6])
        else:
            dinv[a] = Minv[a, a]
    m["dof_invweight0"] = dinv
    binv = np.zeros((nbody, 2))
    for b in range(1, nbody):
        if m["body_weldid"][b] == 0:
            continue
        A = jacs[b] @ Minv @ jacs[b].T
        binv[b, 0] = max(MINVAL, np.trace(A[:3, :3]) / 3)
        binv[b, 1] = max(MINVAL, np.trace(A[3:, 3:]) / 3)
    m["body_invweight0"] = binv
    tinv = np.zeros(len(m["tendon_adr"]))
    for t in range(len(m["tendon_adr"])):
        J = np.zeros(nv)
        for w in range(m["tendon_adr"][t], m["tendon_adr"][t] + m["tendon_num"][t]):
            J[m["jnt_dofadr"][m["wrap_objid"][w]]] = m["wrap_prm"][w]
        tinv[t] = J @ Minv @ J
    m["tendon_invweight0"] = tinv
    sub = m["body_mass"].copy()
    for b in range(nbody - 1, 0, -1):
        sub[m["body_parentid"][b]] += sub[b]
    m["body_subtreemass"] = sub


# ----------------------------------------------------------------------------- public helpers
def compile_file(path: str) -> Dict[str, np.ndarray]:
    c = MjcfCompiler()
    c.parse_file(path)
    return c.compile()


def compile_string(xml: str, base_dir: str = ".") -> Dict[str, np.ndarray]:
    c = MjcfCompiler()
    c.parse_string(xml, base_dir)
    return c.compile()


def empty_scene_xml(stretch_xml_path: str) -> str:
    """The build's definition of the "empty scene" (SURVEY.md §0 finding 6): stretch.xml + one ground plane."""
    return (f'<mujoco model="stretch_empty"><include file="{stretch_xml_path}"/>'
            '<worldbody><geom name="floor" type="plane" size="0 0 0.05"/></worldbody></mujoco>')


def kitchen_standin_xml(stretch_xml_path: str, free_ball: bool = False, free_objects: bool = False) -> str:
    """Synthetic kitchen stand-in (Robocasa itself is unavailable here): the robot at
    the origin in a 5 m x 5 m room with a counter run on its arm side (-y), wall cabinets above it, an island in front, a
    fridge, a table and a few fixtures -- 24 STATIC boxes on the world body, all colliding with the robot ([MJ] default
    contype = conaffinity = 1) and visible to lidar and depth cameras.  free_ball adds ONE free object, a ball on the
    countertop: 6 more dofs fill the step kernel's 32-dof capacity exactly."""
    boxes = []

    def box(name, pos, half, rgba="0.7 0.7 0.7 1"):
        boxes.append(f'<geom name="{name}" type="box" pos="{pos[0]} {pos[1]} {pos[2]}" size="{half[0]} {half[1]} {half[2]}" rgba="{rgba}"/>')
    # walls (0.1 thick, 2.4 high)
    box("wall_xp", (2.55, 0, 1.2), (0.05, 2.6, 1.2)); box("wall_xn", (-2.55, 0, 1.2), (0.05, 2.6, 1.2))
    box("wall_yp", (0, 2.55, 1.2), (2.6, 0.05, 1.2)); box("wall_yn", (0, -2.55, 1.2), (2.6, 0.05, 1.2))
    # counter run along the arm side: four base cabinets 0.6 wide, front face at y = -0.78, top at 0.88, slab on top
    for i, x in enumerate((-0.9, -0.3, 0.3, 0.9)):
        box(f"counter_{i}", (x, -1.08, 0.44), (0.30, 0.30, 0.44), "0.55 0.4 0.3 1")
    box("countertop", (0, -1.07, 0.90), (1.22, 0.32, 0.02), "0.85 0.85 0.8 1")
    # wall cabinets above the counter
    for i, x in enumerate((-0.9, -0.3, 0.3, 0.9)):
        box(f"wallcab_{i}", (x, -1.22, 1.75), (0.30, 0.17, 0.30), "0.55 0.4 0.3 1")
    box("island", (1.45, 0.2, 0.45), (0.35, 0.6, 0.45), "0.5 0.5 0.55 1")
    box("fridge", (-1.95, -1.95, 0.9), (0.35, 0.35, 0.9), "0.9 0.9 0.95 1")
    box("hood", (0.3, -1.15, 1.35), (0.3, 0.22, 0.05), "0.6 0.6 0.6 1")
    box("sink_block", (-0.9, -1.02, 0.96), (0.2, 0.15, 0.04), "0.75 0.75 0.8 1")
    # table with four legs
    box("table_top", (0.0, 1.6, 0.74), (0.6, 0.4, 0.02), "0.6 0.45 0.3 1")
    for i, (x, y) in enumerate(((-0.55, 1.25), (0.55, 1.25), (-0.55, 1.95), (0.55, 1.95))):
        box(f"table_leg_{i}", (x, y, 0.36), (0.025, 0.025, 0.36), "0.6 0.45 0.3 1")
    box("shelf", (-2.3, 0.8, 1.0), (0.2, 0.5, 0.02), "0.55 0.4 0.3 1")
    box("stool", (1.0, 1.2, 0.25), (0.18, 0.18, 0.25), "0.3 0.3 0.3 1")
    assert len(boxes) == 24
    # one free object: a ball resting on the countertop within reach of the gripper (6 dofs -- together with the robot's 26
    # exactly the kernel's 32-dof capacity; sphere contacts are single-point exact, no multi-contact manifold needed)
    ball = ('<body name="ball" pos="-0.3 -0.95 0.9597"><freejoint name="ball_free"/>'
            '<geom name="ball" type="sphere" size="0.04" mass="0.1" rgba="0.9 0.3 0.2 1" condim="4" friction="1 0.01 0.001"/></body>') if free_ball else ""
    if free_objects:
        # config 4 as SURVEY.md 8(d) specifies it: 4 free objects (2 boxes, 2 cylinders), two on the countertop within reach of
        # the gripper and two on the table.  Box / cylinder resting contacts are the multi-point cases (box-box, multiccd).
        ball = ('<body name="box_a" pos="-0.45 -0.95 0.96"><freejoint/><geom name="box_a" type="box" size="0.03 0.03 0.04" mass="0.2" rgba="0.2 0.2 0.8 1"/></body>'
                '<body name="cyl_a" pos="-0.15 -0.95 0.96"><freejoint/><geom name="cyl_a" type="cylinder" size="0.03 0.04" mass="0.2" rgba="0.8 0.2 0.2 1"/></body>'
                '<body name="box_b" pos="-0.2 1.45 0.80"><freejoint/><geom name="box_b" type="box" size="0.04 0.03 0.04" mass="0.3" rgba="0.2 0.7 0.3 1"/></body>'
                '<body name="cyl_b" pos="0.2 1.45 0.80"><freejoint/><geom name="cyl_b" type="cylinder" size="0.03 0.04" mass="0.3" rgba="0.8 0.7 0.2 1"/></body>')
    return (f'<mujoco model="stretch_kitchen_standin"><include file="{stretch_xml_path}"/>'
            '<worldbody><geom name="floor" type="plane" size="0 0 0.05"/>' + "".join(boxes) + ball + '</worldbody></mujoco>')


def scene_table_xml(stretch_xml_path: str) -> str:
    """The reference's own default scene (stretch_mujoco/models/scene.xml:21-35): floor, a table (a jointless body: welded to the
    world) and two free objects on it, a box and a cylinder.  The docking station of scene.xml is left out: its mesh
    link_docking_base.obj is absent from the checkout (.MISSING_LARGE_BLOBS:2)."""
    return (f'<mujoco model="stretch scene"><include file="{stretch_xml_path}"/>'
            '<worldbody><geom name="floor" size="0 0 0.05" type="plane"/>'
            '<body name="table" pos="0 -1 .24"><geom type="box" size=".6 .5 .24" mass="1"/></body>'
            '<body name="object1" pos="-.02 -0.55 .6"><freejoint/><geom type="box" size=".02 .04 .04" mass=".5" rgba=".2 .2 .5 1"/></body>'
            '<body name="object2" pos=".08 -0.55 .6"><freejoint/><geom type="cylinder" size=".02 .04 .04" mass=".5" rgba=".8 .2 .2 1"/></body>'
            '</worldbody></mujoco>')

