"""Pre-exported Robocasa / robosuite kitchen XML -> a scene this build can compile.

Robocasa and robosuite (the reference's `third_party/` submodules) and their assets are not available here, so kitchens
cannot be GENERATED; what can be done is the second half of the reference's generator, applied to a kitchen XML exported
elsewhere (`env.sim.model.get_xml()` with absolute asset paths, robocasa_gen.py:196-197, written by
`model_generation_wizard(write_to_file=...)` before its own clean-up or by any robosuite script).  The contract restated from
stretch_mujoco/robocasa_gen.py:

  * custom_cleanups (:242-264): the red / blue marker boxes around geoms and sites of interest become invisible (alpha 0 --
    invisible to cameras and rangefinders too); the <actuator>, <sensor> and <option> sections go; the body `robot0_base`
    (robosuite's mobile manipulator) goes, and its pos / quat attributes are kept as the robot's spawn pose;
  * add_stretch_to_kitchen (:267-280) + utils.get_absolute_path_stretch_xml (utils.py:311-349): `stretch.xml` is included
    right after the <mujoco> tag with `base_link` placed at that pose (pos "x y z", quat as written in the kitchen XML, i.e.
    MuJoCo's w x y z; the reference's docstring says x y z w but passes the attribute through unchanged).

Here the pose is RETURNED instead of being patched into stretch.xml: StretchBatchSimulator takes it as start_translation /
start_rotation_quat (per env), which is what `change_start_pose` does in the reference (mujoco_server.py:206-229).
"""
from __future__ import annotations

import xml.etree.ElementTree as ET
from typing import Dict, Optional, Tuple

# (tag, attribute, value to look for, replacement) -- robocasa_gen.py:249-252
_MARKERS = (("geom", "rgba", "0.5 0 0 0.5", "0.5 0 0 0"), ("geom", "rgba", "0.5 0 0 1", "0.5 0 0 0"),
            ("site", "rgba", "0.5 0 0 1", "0.5 0 0 0"), ("site", "actuator", "0.3 0.4 1 0.5", "0.3 0.4 1 0"))
_DROP_SECTIONS = ("actuator", "sensor", "option")   # robocasa_gen.py:254-259
ROBOT_BODY = "robot0_base"                          # robocasa_gen.py:263


def convert_kitchen_xml(xml: str, stretch_xml_path: str, robot_spawn_pose: Optional[Dict[str, str]] = None) -> Tuple[str, Dict[str, list]]:
    """Returns (scene MJCF that includes stretch.xml, {"pos": [x, y, z], "quat": [w, x, y, z]} of the removed robot base).
    `robot_spawn_pose` ({"pos": "x y z", "quat": "w x y z"}) overrides the fixture pose, like the wizard's argument."""
    root = ET.fromstring(xml)
    if root.tag != "mujoco":
        raise ValueError("not an MJCF document")
    for tag, attr, old, new in _MARKERS:
        for e in root.iter(tag):
            if e.get(attr) == old:
                e.set(attr, new)
    for name in _DROP_SECTIONS:
        for e in list(root.findall(name)):
            root.remove(e)
    parent = {c: p for p in root.iter() for c in p}
    removed, gone = None, set()
    for e in root.iter("body"):
        if e.get("name") == ROBOT_BODY:
            removed = dict(e.attrib)
            gone = {x.get("name") for x in e.iter() if x.get("name")}   # bodies, joints, geoms, sites of the robot
            parent[e].remove(e)
            break
    if removed is None and robot_spawn_pose is None:
        raise ValueError(f"the kitchen XML has no body '{ROBOT_BODY}' and no robot_spawn_pose was given")
    pose = robot_spawn_pose if robot_spawn_pose is not None else removed
    # what referred to the removed robot would dangle (MuJoCo refuses such a file): contact excludes / pairs, equalities,
    # tendons over its joints -- the actuator and sensor sections are gone already
    for section in ("contact", "equality", "tendon"):
        for sec in root.findall(section):
            for e in list(sec):
                refs = [e.get(a) for a in ("body1", "body2", "geom1", "geom2", "joint1", "joint2", "site1", "site2")] + \
                       [j.get("joint") for j in e.iter("joint")] + [j.get("site") for j in e.iter("site")]
                if any(r in gone for r in refs if r):
                    sec.remove(e)
    inc = ET.Element("include", {"file": stretch_xml_path})
    root.insert(0, inc)
    out = {"pos": [float(v) for v in pose.get("pos", "0 0 0").split()],
           "quat": [float(v) for v in pose.get("quat", "1 0 0 0").split()]}
    return ET.tostring(root, encoding="unicode"), out
