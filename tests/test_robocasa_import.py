"""Converter for a pre-exported Robocasa kitchen XML (robocasa_import.py; contract: stretch_mujoco/robocasa_gen.py:242-280).
Robocasa itself is unavailable: the input here is a small hand-written robosuite-style document with the features the
clean-up touches."""
import os
import xml.etree.ElementTree as ET

import numpy as np
import pytest

from stretch_mujoco_amd import mjcf_compiler, model_fuse
from stretch_mujoco_amd.robocasa_import import convert_kitchen_xml

KITCHEN = """<mujoco model="kitchen">
  <option timestep="0.001" integrator="Euler"/>
  <worldbody>
    <geom name="floor" type="plane" size="0 0 0.05"/>
    <body name="counter_main" pos="1.0 -1.2 0.45"><geom name="counter" type="box" size="0.6 0.3 0.45"/>
      <geom name="counter_reg" type="box" size="0.5 0.2 0.01" pos="0 0 0.46" rgba="0.5 0 0 0.5" contype="0" conaffinity="0"/>
      <site name="counter_site" pos="0 0 0.5" rgba="0.5 0 0 1"/></body>
    <body name="obj_main" pos="1.0 -1.1 0.95"><freejoint name="obj_joint0"/><geom name="obj_g0" type="box" size="0.03 0.03 0.04" mass="0.2"/></body>
    <body name="robot0_base" pos="0.5 -0.3 0" quat="0.7071068 0 0 0.7071068">
      <joint name="robot0_joint_mobile_forward" type="slide" axis="1 0 0"/>
      <geom name="robot0_g0" type="box" size="0.2 0.2 0.2" mass="10"/>
      <body name="robot0_link1" pos="0 0 0.4"><joint name="robot0_joint1" type="hinge" axis="0 0 1"/><geom name="robot0_g1" type="sphere" size="0.05" mass="1"/></body>
    </body>
  </worldbody>
  <contact><exclude body1="robot0_base" body2="robot0_link1"/><exclude body1="counter_main" body2="obj_main"/></contact>
  <actuator><motor name="robot0_m1" joint="robot0_joint1"/></actuator>
  <sensor><jointpos name="robot0_s1" joint="robot0_joint1"/></sensor>
</mujoco>"""
STRETCH = "/root/reference/stretch_mujoco/models/stretch.xml"


def test_cleanups_follow_the_reference_generator():
    out, pose = convert_kitchen_xml(KITCHEN, "stretch.xml")
    root = ET.fromstring(out)
    assert root[0].tag == "include" and root[0].get("file") == "stretch.xml"           # right after the <mujoco> tag
    assert root.find("actuator") is None and root.find("sensor") is None and root.find("option") is None
    assert all(b.get("name") != "robot0_base" for b in root.iter("body"))
    assert pose == {"pos": [0.5, -0.3, 0.0], "quat": [0.7071068, 0.0, 0.0, 0.7071068]}
    assert root.find(".//geom[@name='counter_reg']").get("rgba") == "0.5 0 0 0"        # marker boxes made invisible
    assert root.find(".//site[@name='counter_site']").get("rgba") == "0.5 0 0 0"
    ex = root.findall(".//contact/exclude")
    assert len(ex) == 1 and ex[0].get("body1") == "counter_main"                        # the robot's own exclude went with it
    out2, pose2 = convert_kitchen_xml(KITCHEN, "stretch.xml", robot_spawn_pose={"pos": "1 2 0", "quat": "1 0 0 0"})
    assert pose2["pos"] == [1.0, 2.0, 0.0]
    with pytest.raises(ValueError):
        convert_kitchen_xml("<mujoco><worldbody/></mujoco>", "stretch.xml")


@pytest.mark.skipif(not os.path.exists(STRETCH), reason="needs the reference's stretch.xml and meshes (this container only)")
def test_converted_kitchen_compiles_with_the_builds_compiler():
    out, pose = convert_kitchen_xml(KITCHEN, STRETCH)
    m = mjcf_compiler.compile_string(out)
    nq, nv = int(m["dims"][0]), int(m["dims"][1])
    assert (nq, nv) == (27 + 7, 26 + 6)                                                   # Stretch + the free object
    f = model_fuse.prepare_for_kernels(m)
    assert int(f["dims"][3]) < int(m["dims"][3])                                          # the jointless counter body fused into the world
    rg = set(int(g) for g in np.asarray(f["k_rgeom"]).ravel()[: int(np.ravel(f["k_nrgeom"])[0])])
    import json
    from stretch_mujoco_amd import model_blob
    names = json.loads(model_blob.get_str(f, "names_json"))["geom"]
    assert names.index("counter") in rg and names.index("counter_reg") not in rg          # alpha 0: invisible to the ray casters
