"""Compile the committed model blobs from the reference's MJCF (runs only where /root/reference is mounted).

    python tools/build_models.py [/root/reference]

Outputs stretch_mujoco_amd/models/*.smjb -- compiled numeric model data (the problem definition the
kernels consume); the GPU box never parses XML.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from stretch_mujoco_amd import mjcf_compiler as C, model_blob as B, model_fuse as F  # noqa: E402


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    stretch = os.path.join(ref, "stretch_mujoco", "models", "stretch.xml")
    out = os.path.join(os.path.dirname(HERE), "stretch_mujoco_amd", "models")
    os.makedirs(out, exist_ok=True)
    m = C.compile_string(C.empty_scene_xml(stretch))
    # body-for-body model (oracle / fusion test); the bulk render meshes live only in the product blob
    full = {k: v for k, v in m.items() if k not in ("rmesh_vert", "rmesh_face")}
    full["rmesh_vert"] = m["rmesh_vert"][:0]; full["rmesh_face"] = m["rmesh_face"][:0]
    full["rmesh_vertnum"] = 0 * m["rmesh_vertnum"]; full["rmesh_facenum"] = 0 * m["rmesh_facenum"]
    B.save(os.path.join(out, "stretch_empty_full.smjb"), full)
    B.save(os.path.join(out, "stretch_empty.smjb"), F.prepare_for_kernels(m))        # fused + kernel tables (product)
    k = C.compile_string(C.kitchen_standin_xml(stretch, free_ball=True))
    # config 4 stand-in (static fixtures + one free ball); contact-rich: the big kernel variant (box-box polygons of the rubber
    # tips on counters need more than 80 rows)
    B.save(os.path.join(out, "stretch_kitchen_standin.smjb"), F.prepare_for_kernels(k, capacity="big"))
    # the reference's own default scene (models/scene.xml: table + 2 free objects) and config 4 as SURVEY.md 8(d) has it
    # (24 static boxes + 2 free boxes + 2 free cylinders): 38 / 50 dofs -> the big variant by size
    for name, xml in (("stretch_scene", C.scene_table_xml(stretch)), ("stretch_kitchen4", C.kitchen_standin_xml(stretch, free_objects=True))):
        mm = C.compile_string(xml)
        B.save(os.path.join(out, name + ".smjb"), F.prepare_for_kernels(mm))
        # the same scene for the satellite builds of the step kernel (csrc/smj_sat.h): the free objects leave the dense problem
        B.save(os.path.join(out, name + "_sat.smjb"), F.prepare_for_kernels(mm, satellites=True))
        print(name + ":", dict(zip("nq nv nu nbody njnt ngeom".split(), [int(x) for x in mm["dims"][:6]])), "npair", int(mm["dims"][12]))
    # the reference's default scene AS SHIPPED TODAY: models/scene.xml compiled from the file itself -- stretch.xml, the docking station
    # (a free body of a plate + 18 convex collision pieces behind the robot; its visual shell link_docking_base.obj is one of the blobs
    # missing from the checkout and is skipped like the robot's two), the table, the two objects: 44 dofs, the 50-column build
    sd = C.compile_file(os.path.join(ref, "stretch_mujoco", "models", "scene.xml"))
    B.save(os.path.join(out, "stretch_scene_docking.smjb"), F.prepare_for_kernels(sd))
    print("stretch_scene_docking:", dict(zip("nq nv nu nbody njnt ngeom".split(), [int(x) for x in sd["dims"][:6]])), "npair", int(sd["dims"][12]))
    # a robosuite-style kitchen export (hand-written test fixture: articulated fixtures, <inertial>, capsule / ellipsoid objects)
    # through the converter for pre-exported Robocasa kitchens (robocasa_import.py; robocasa_gen.py:242-280)
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
    from kitchen_export_fixture import KITCHEN_EXPORT
    from stretch_mujoco_amd.robocasa_import import convert_kitchen_xml
    kx, pose = convert_kitchen_xml(KITCHEN_EXPORT, stretch)
    ke = C.compile_string(kx)
    B.save(os.path.join(out, "stretch_kitchen_export.smjb"), F.prepare_for_kernels(ke))
    B.save(os.path.join(out, "stretch_kitchen_export_sat.smjb"), F.prepare_for_kernels(ke, satellites=True))   # door, drawer and the objects as satellites
    print("stretch_kitchen_export:", dict(zip("nq nv nu nbody njnt ngeom".split(), [int(x) for x in ke["dims"][:6]])), "npair", int(ke["dims"][12]), "spawn pose", pose)
    # a kitchen at Robocasa scale (generated test fixture: 44 fixture bodies, 307 collision geoms incl. 36 convex mesh pieces, 8
    # articulated doors / drawers / knobs, 8 free objects) through the same converter; the removed robot's pose becomes the start
    # pose (what change_start_pose does in the reference, mujoco_server.py:206-229).  Satellite build + static-geometry grid.
    from kitchen_robocasa_fixture import kitchen_xml
    rx, rstats = kitchen_xml()
    rkx, rpose = convert_kitchen_xml(rx, stretch)
    rk = C.compile_string(rkx)
    rk["qpos0"][0:3] = rpose["pos"]; rk["qpos0"][3:7] = rpose["quat"]
    B.save(os.path.join(out, "stretch_kitchen_robocasa.smjb"), F.prepare_for_kernels(rk, satellites=True))
    print("stretch_kitchen_robocasa:", rstats, dict(zip("nq nv nu nbody njnt ngeom".split(), [int(x) for x in rk["dims"][:6]])), "npair", int(rk["dims"][12]))
    print("stretch_kitchen_standin:", dict(zip("nq nv nu nbody njnt ngeom".split(), [int(x) for x in k["dims"][:6]])),
          "npair", int(k["dims"][12]))
    print("stretch_empty:", dict(zip("nq nv nu nbody njnt ngeom nsite ncam neq ntendon nwrap nkey npair nhullvert".split(),
                                     [int(x) for x in m["dims"]])))


if __name__ == "__main__":
    main()
