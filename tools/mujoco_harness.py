"""Live-MuJoCo harness: runs the moment `import mujoco` works (mujoco==3.2.x is the reference's pinned
physics, pyproject.toml:12; it is not installable in this build's containers).  The compiled model blob is exported as MJCF
(stretch_mujoco_amd/mjcf_export.py: explicit inertials, inline hull meshes, explicit contact pairs -- nothing left for MuJoCo's
compiler to decide), loaded into MuJoCo, and compared with the fp64 oracle (and through it with the HIP kernels, whose parity
tests run against the oracle):

  D.1-D.8   model constants MuJoCo derives (dims, invweight0, meaninertia, subtree masses, qpos0) vs the blob
  D.10      trajectories from the settled home pose: fixed ctrl scripts (home / mixed / driving / head_tilt limit) -- per step
            qpos, qvel, qacc, ncon, nefc, efc_force, contacts, sensordata (gyro, accelerometer, lidar with --visual)
  C.3       the status bands the reference's docs print (README.md:136-145, getting_started.ipynb cell 20/23)

    python tools/mujoco_harness.py [scene] [--steps N] [--visual] [--json out.json]

Without MuJoCo it prints the UNVERIFIED notice and exits 3.  TEST INFRASTRUCTURE (uses oracle/).
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

UNVERIFIED = ("MuJoCo oracle unavailable -- parity is against the build's own fp64 CPU restatement (oracle/); "
              "MuJoCo parity UNVERIFIED (the restatement is pinned to the MuJoCo output the reference's notebook prints: joint status "
              "at t = 8.26 s to 1e-6..2e-5, head_tilt limit stop to 2e-9, 30 depth pixels of both cameras to the printed 1e-3, and to the images it stores: "
              "53 400 pixels of the wrist depth map to 0.3 grey levels, the other camera frames by horizon / colour class, the lidar figure ray by ray, "
              "all at the base pose it prints; tests/test_oracle_physics.py, tests/test_depth_oracle.py, tests/test_notebook_images.py)")
SCRIPTS = {"home": [0, 0, 0.6, 0.1, 0, 0, 0, 0, 0, 0], "mixed": [2, -1, 0.6, 0.1, 1, -0.4, 0.5, 0.02, 0.3, -0.2],
           "driving": [3, -3, 0.2, 0.3, 2, -1, -1, 0.03, -2, 0.5], "head_tilt_limit": [0, 0, 0.6, 0.1, 0, 0, 0, 0, 0, -2.0]}


def have_mujoco():
    try:
        import mujoco  # noqa: F401
        return True
    except Exception:
        return False


def compare(scene="stretch_empty", steps=1000, visual=False):
    """Returns a dict of comparisons MuJoCo vs oracle.  Raises ImportError without MuJoCo."""
    import mujoco

    from oracle.oracle import Oracle
    from stretch_mujoco_amd import model_blob
    from stretch_mujoco_amd.mjcf_export import export_mjcf

    with open(os.path.join(ROOT, "stretch_mujoco_amd", "models", scene + ".smjb"), "rb") as f:
        blob = f.read()
    m = model_blob.loads(blob)
    xml = export_mjcf(m, with_visual=visual)
    mjm = mujoco.MjModel.from_xml_string(xml)
    res = {"mujoco_version": mujoco.__version__, "scene": scene, "model": {}, "trajectories": {}}
    nq, nv, nu = int(m["dims"][0]), int(m["dims"][1]), int(m["dims"][2])
    M = res["model"]
    M["dims"] = {"nq": (mjm.nq, nq), "nv": (mjm.nv, nv), "nu": (mjm.nu, nu), "nbody": (mjm.nbody, int(m["dims"][3])), "neq": (mjm.neq, len(m["eq_obj1id"]))}

    def rel(a, b):
        a, b = np.asarray(a, float).ravel(), np.asarray(b, float).ravel()
        return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))
    M["qpos0"] = rel(mjm.qpos0, m["qpos0"])
    M["body_mass"] = rel(mjm.body_mass, m["body_mass"])
    M["body_subtreemass"] = rel(mjm.body_subtreemass, m["body_subtreemass"])
    M["dof_invweight0"] = rel(mjm.dof_invweight0, m["dof_invweight0"])
    M["body_invweight0"] = rel(mjm.body_invweight0, m["body_invweight0"])
    M["stat_meaninertia"] = rel([mjm.stat.meaninertia], m["stat_meaninertia"])
    M["actuator_gainprm"] = rel(mjm.actuator_gainprm[:, :3], m["actuator_gainprm"])
    M["actuator_biasprm"] = rel(mjm.actuator_biasprm[:, :3], m["actuator_biasprm"])
    M["jnt_range"] = rel(mjm.jnt_range, m["jnt_range"])
    for name, ctrl in SCRIPTS.items():
        d = mujoco.MjData(mjm)
        o = Oracle(blob)
        o.set_option("solver", 2)
        d.ctrl[:] = SCRIPTS["home"][:nu]; o.arr("ctrl")[:nu] = SCRIPTS["home"][:nu]
        for _ in range(500):
            mujoco.mj_step(mjm, d)
        # continue BOTH from MuJoCo's settled state: the comparison is of the step arithmetic on identical inputs
        o.arr("qpos")[:] = d.qpos; o.arr("qvel")[:] = d.qvel; o.arr("qacc_warmstart")[:] = d.qacc_warmstart
        d.ctrl[:] = ctrl[:nu]; o.arr("ctrl")[:nu] = ctrl[:nu]
        drift_q, drift_v, dncon, dnefc, dforce, dsens = [], [], 0, 0, [], []
        for k in range(steps):
            mujoco.mj_step(mjm, d); o.step(1)
            drift_q.append(float(np.abs(d.qpos - o.arr("qpos")).max())); drift_v.append(float(np.abs(d.qvel - o.arr("qvel")).max()))
            dncon += int(d.ncon != o.ncon); dnefc += int(d.nefc != o.nefc)
            if d.nefc == o.nefc and d.nefc:
                f = np.asarray(d.efc_force[: d.nefc]); dforce.append(float(np.abs(f - o.arr("efc_force")[: d.nefc]).max() / max(1.0, np.abs(f).max())))
            if k % 50 == 49:
                o.sensors(bool(visual))
                sd = np.asarray(d.sensordata)
                e = [float(np.abs(sd[0:3] - o.arr("gyro")).max()), float(np.abs(sd[3:6] - o.arr("accel")).max())]
                if visual:
                    e.append(float(np.abs(sd[6:6 + 360] - o.arr("lidar")).max()))
                dsens.append(e)
        st = {"lift": float(d.actuator_length[2]), "arm": float(d.actuator_length[3]), "head_tilt": float(d.actuator_length[9])}
        res["trajectories"][name] = {"steps": steps, "max_qpos_drift": max(drift_q), "qpos_drift_at_100": drift_q[min(99, steps - 1)],
                                     "max_qvel_drift": max(drift_v), "steps_ncon_differs": dncon, "steps_nefc_differs": dnefc,
                                     "max_rel_efc_force_err": max(dforce) if dforce else None,
                                     "sensor_err_gyro_accel_lidar": np.max(np.array(dsens), 0).tolist() if dsens else None, "final_status": st}
    bands = res["trajectories"]["home"]["final_status"]
    res["doc_bands"] = {"lift 0.589/0.5906 (README.md:138, notebook)": bands["lift"], "arm 0.0981/0.1000": bands["arm"],
                        "head_tilt -0.004519": bands["head_tilt"],
                        "head_tilt limit stop -1.52257 (notebook cell 23)": res["trajectories"]["head_tilt_limit"]["final_status"]["head_tilt"]}
    ok = [t["max_qpos_drift"] < 1e-4 for t in res["trajectories"].values()]
    res["verdict"] = ("MuJoCo %s: qpos drift of the fp64 restatement < 1e-4 over %d steps on %d of %d scripts" %
                      (mujoco.__version__, steps, sum(ok), len(ok)))
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scene", nargs="?", default="stretch_empty")
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--visual", action="store_true", help="export the render meshes too and compare the 360 rangefinders")
    ap.add_argument("--json")
    a = ap.parse_args()
    if not have_mujoco():
        print(UNVERIFIED)
        sys.exit(3)
    res = compare(a.scene, a.steps, a.visual)
    print(json.dumps(res, indent=1))
    if a.json:
        with open(a.json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
