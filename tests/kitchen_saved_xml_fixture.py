"""TEST FIXTURE -- the generated kitchen of kitchen_robocasa_fixture.py re-written in the SHAPE of a real export: what
`env.sim.model.get_xml()` (MuJoCo's saved XML, robocasa_gen.py:196-197, absolute asset paths) hands the reference's generator, as far
as its shape is known without Robocasa:

  * mesh assets ON DISK under absolute paths -- Wavefront OBJ (with normals / texture coordinates / polygon faces, the things a
    parser trips over) and binary STL, a `scale` on some -- instead of inline vertex data;
  * <compiler angle="radian" autolimits="true" meshdir=... texturedir=...>, <size>, <visual> with a <map>, <statistic>;
  * nested <default class> chains (main -> fixture -> collision / visual; main -> object -> ...): geoms carry a class, bodies a
    childclass, the per-geom attributes of the inline fixture move into the defaults;
  * <texture> (2d files, a builtin skybox, a cube) and <material texture=... texrepeat=... specular=...> blocks, geoms that name a
    material; <light>s and free <camera>s in the world and on fixtures;
  * site-heavy fixture bodies in robosuite's conventions (`*_int_p0 / px / py / pz` interior markers, `*_ext_*` exterior markers,
    handle / spout sites, in the marker colours the reference's clean-up looks for);
  * the robot `robot0_base` with its own class, actuators and sensors referring to it, <contact><exclude>, <keyframe>-free.

`saved_kitchen_xml(asset_dir)` writes the assets and returns (document, stats).  The physics is the inline fixture's, so the compiled
models must agree (tests/test_robocasa_import.py)."""
import math
import os
import struct
import xml.etree.ElementTree as ET

import numpy as np

from kitchen_robocasa_fixture import kitchen_xml


def _hull_faces(v):
    from scipy.spatial import ConvexHull

    h = ConvexHull(v)
    c = v.mean(0)
    f = []
    for s in h.simplices:
        a, b, d = v[s]
        f.append(list(s) if np.dot(np.cross(b - a, d - a), a - c) > 0 else [s[0], s[2], s[1]])
    return np.array(f)


def _write_obj(path, v, f):
    with open(path, "w") as fh:
        fh.write("# exported\nmtllib none.mtl\no part\n")
        for p in v:
            fh.write("v %.9g %.9g %.9g\n" % tuple(p))
        fh.write("vt 0 0\nvt 1 0\nvt 1 1\n")
        for k in range(len(f)):
            n = np.cross(v[f[k][1]] - v[f[k][0]], v[f[k][2]] - v[f[k][0]])
            n = n / (np.linalg.norm(n) + 1e-30)
            fh.write("vn %.6f %.6f %.6f\n" % tuple(n))
        fh.write("usemtl none\ns off\n")
        for k, t in enumerate(f):
            fh.write("f " + " ".join(f"{i + 1}/{1 + j}/{k + 1}" for j, i in enumerate(t)) + "\n")


def _write_stl(path, v, f):
    with open(path, "wb") as fh:
        fh.write(b"binary stl".ljust(80, b" "))
        fh.write(struct.pack("<I", len(f)))
        for t in f:
            a, b, c = v[t]
            n = np.cross(b - a, c - a)
            n = n / (np.linalg.norm(n) + 1e-30)
            fh.write(struct.pack("<12fH", *n, *a, *b, *c, 0))


def saved_kitchen_xml(asset_dir):
    xml, stats = kitchen_xml()
    src = ET.fromstring(xml)
    os.makedirs(os.path.join(asset_dir, "meshes"), exist_ok=True)
    os.makedirs(os.path.join(asset_dir, "textures"), exist_ok=True)
    root = ET.Element("mujoco", {"model": "kitchen_saved"})
    ET.SubElement(root, "compiler", {"angle": "radian", "autolimits": "true", "meshdir": os.path.join(asset_dir, "unused"), "texturedir": os.path.join(asset_dir, "textures")})
    ET.SubElement(root, "option", dict(src.find("option").attrib))
    ET.SubElement(root, "size", {"njmax": "5000", "nconmax": "5000", "nstack": "9000000"})
    vis = ET.SubElement(root, "visual")
    ET.SubElement(vis, "map", {"znear": "0.001", "zfar": "50"})
    ET.SubElement(vis, "quality", {"shadowsize": "4096"})
    ET.SubElement(vis, "headlight", {"ambient": "0.4 0.4 0.4"})
    ET.SubElement(root, "statistic", {"extent": "6.5", "center": "0 0 1"})
    # defaults: main -> fixture -> (fixture_col, fixture_vis); main -> object -> (object_col, object_vis); robot0
    dflt = ET.SubElement(root, "default")
    main = ET.SubElement(dflt, "default", {"class": "main"})
    ET.SubElement(main, "geom", {"solref": "0.02 1"})
    fx = ET.SubElement(main, "default", {"class": "fixture"})
    ET.SubElement(fx, "joint", {"armature": "0"})
    fcol = ET.SubElement(fx, "default", {"class": "fixture_col"})
    ET.SubElement(fcol, "geom", {"group": "0", "rgba": "0.5 0 0 1"})
    fvis = ET.SubElement(fx, "default", {"class": "fixture_vis"})
    ET.SubElement(fvis, "geom", {"group": "1", "contype": "0", "conaffinity": "0", "mass": "0", "material": "wood_mat"})
    ob = ET.SubElement(main, "default", {"class": "object"})
    ocol = ET.SubElement(ob, "default", {"class": "object_col"})
    ET.SubElement(ocol, "geom", {"group": "0", "rgba": "0.5 0 0 1"})
    ovis = ET.SubElement(ob, "default", {"class": "object_vis"})
    ET.SubElement(ovis, "geom", {"group": "1", "contype": "0", "conaffinity": "0", "mass": "0"})
    rb = ET.SubElement(dflt, "default", {"class": "robot0"})
    ET.SubElement(rb, "geom", {"rgba": "0.2 0.2 0.2 1"})
    ET.SubElement(ET.SubElement(main, "default", {"class": "marker"}), "site", {"size": "0.01", "group": "3"})
    # assets: textures, materials, meshes as files (absolute paths)
    asset = ET.SubElement(root, "asset")
    ET.SubElement(asset, "texture", {"type": "skybox", "builtin": "gradient", "rgb1": "0.9 0.9 1", "rgb2": "0.2 0.3 0.4", "width": "256", "height": "1536"})
    for k, name in enumerate(("wood", "marble", "steel")):
        with open(os.path.join(asset_dir, "textures", name + ".png"), "wb") as fh:
            fh.write(b"\x89PNG\r\n\x1a\n")   # never decoded: the compiler must not open textures
        ET.SubElement(asset, "texture", {"type": "2d" if k else "cube", "name": name + "_tex", "file": os.path.join(asset_dir, "textures", name + ".png")})
        ET.SubElement(asset, "material", {"name": name + "_mat", "texture": name + "_tex", "texrepeat": "3 3", "specular": "0.4", "shininess": "0.1",
                                          **({"rgba": "0.6 0.45 0.3 1"} if name == "wood" else {})})
    nfile = {"obj": 0, "stl": 0, "scaled": 0}
    for k, me in enumerate(src.find("asset").findall("mesh")):
        v = np.array([float(x) for x in me.get("vertex").split()]).reshape(-1, 3)
        f = _hull_faces(v)
        a = {"name": me.get("name")}
        if k % 3 == 2:   # stored at double size, scaled back by the asset (a saved XML keeps the user's scale)
            v = v * 2.0
            a["scale"] = "0.5 0.5 0.5"
            nfile["scaled"] += 1
        if k % 2:
            path = os.path.join(asset_dir, "meshes", me.get("name") + ".stl")
            _write_stl(path, v, f)
            nfile["stl"] += 1
        else:
            path = os.path.join(asset_dir, "meshes", me.get("name") + ".obj")
            _write_obj(path, v, f)
            nfile["obj"] += 1
        a["file"] = path
        ET.SubElement(asset, "mesh", a)
    # bodies: classes instead of per-geom attributes, sites, cameras, lights
    wb = ET.SubElement(root, "worldbody")
    ET.SubElement(wb, "light", {"pos": "0 0 3", "dir": "0 0 -1", "diffuse": "0.8 0.8 0.8", "castshadow": "false"})
    ET.SubElement(wb, "camera", {"name": "robot0_agentview_center", "pos": "0 -2.5 2", "quat": "0.9 0.4 0 0", "fovy": "60"})
    ET.SubElement(wb, "camera", {"name": "robot0_frontview", "pos": "2.5 0 1.5", "xyaxes": "0 1 0 -0.4 0 0.9"})
    nsite = 0

    def convert(e, out, is_robot=False):
        nonlocal nsite
        for ch in e:
            if ch.tag == "body":
                name = ch.get("name", "")
                robot = is_robot or name.startswith("robot0")
                free = ch.find("freejoint") is not None
                a = dict(ch.attrib)
                a["childclass"] = "robot0" if robot else "object" if free else "fixture"
                b = ET.SubElement(out, "body", a)
                convert(ch, b, robot)
                if not robot:
                    base = name.replace("_main", "")
                    for suf, pos in (("int_p0", "-0.1 -0.1 -0.1"), ("int_px", "0.1 -0.1 -0.1"), ("int_py", "-0.1 0.1 -0.1"), ("int_pz", "-0.1 -0.1 0.1"),
                                     ("ext_p0", "-0.2 -0.2 -0.2"), ("ext_px", "0.2 -0.2 -0.2"), ("default_site", "0 0 0")):
                        ET.SubElement(b, "site", {"name": f"{base}_{suf}", "pos": pos, "class": "marker", "rgba": "0.5 0 0 1"})
                        nsite += 1
                    if "door" in name or "drawer" in name:
                        ET.SubElement(b, "site", {"name": f"{base}_handle_site", "pos": "0 0.02 0", "class": "marker", "rgba": "0.3 0.4 1 0.5"})
                        nsite += 1
                    if name in ("hood", "fridge", "island"):
                        ET.SubElement(b, "camera", {"name": f"{base}_cam", "pos": "0 0 0.3", "euler": "0.3 0 0"})
                        ET.SubElement(b, "light", {"pos": "0 0 0.5", "dir": "0 0 -1"})
            elif ch.tag == "geom":
                a = dict(ch.attrib)
                if is_robot:
                    a.pop("rgba", None)
                elif a.get("group") == "0" and a.get("rgba") == "0.5 0 0 1":
                    a.pop("group"); a.pop("rgba")
                    a["class"] = "object_col" if out.get("childclass") == "object" else "fixture_col"
                elif a.get("group") == "1":
                    for k in ("group", "contype", "conaffinity", "mass"):
                        a.pop(k, None)
                    a["class"] = "object_vis" if out.get("childclass") == "object" else "fixture_vis"
                    if out.get("childclass") == "fixture" and a.get("rgba") == "0.6 0.45 0.3 1":
                        a.pop("rgba")            # comes from wood_mat through the class
                    elif out.get("childclass") == "fixture":
                        a["material"] = "steel_mat" if a.get("rgba", "").startswith("0.8 0.8") else a.get("material", "marble_mat")
                ET.SubElement(out, "geom", a)
            else:
                out.append(ch)

    convert(src.find("worldbody"), wb)
    for sec in ("contact", "actuator", "sensor"):
        root.append(src.find(sec))
    stats = dict(stats, mesh_files=nfile, sites=nsite, cameras=5, textures=4, materials=3)
    return ET.tostring(root, encoding="unicode"), stats
