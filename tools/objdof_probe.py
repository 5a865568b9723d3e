"""CPU probe (lane emulator + fp64 oracle): one-step accelerations along the KERNEL's trajectory, robot dofs and object dofs each on
their own scale, with the manifold cache on / off.  VERDICT r5 "weak" 1-2.  TEST INFRASTRUCTURE (imports oracle/).
usage: python tools/objdof_probe.py [scene] [steps] [envs] [cache values ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rollout_common as rc
import stretch_mujoco_amd.model_blob as mb
from oracle.oracle import Oracle


def run(scene, steps, B, cache, oracle_opts=None, seed=7, verbose=False):
    blob = open(os.path.join(ROOT, "stretch_mujoco_amd", "models", scene + ".smjb"), "rb").read()
    model = mb.loads(blob)
    be = rc.EmulBackend(blob, B)
    be.e.set_option("manifold_cache", cache)
    oracles = rc.settled_oracles(blob, B, 2)
    nu, nv = oracles[0].dim("nu"), oracles[0].dim("nv")
    be.upload(*rc.state_of(oracles))
    sched = rc.ctrl_schedule(model, nu, B, (steps + rc.HOLD - 1) // rc.HOLD, seed)
    o = [Oracle(blob) for _ in range(B)]
    for x in o:
        x.set_option("solver", 2)
        for k, v in (oracle_opts or {}).items():
            x.set_option(k, v)
    rob, obj, objabs = [], [], []
    for s in range(steps):
        c = sched[s // rc.HOLD]
        be.set_ctrl(c)
        pre = be.download()
        be.upload(pre["qpos"], pre["qvel"], pre["warm"])
        be.step(1)
        post = be.download()
        for b in range(B):
            x = o[b]
            x.arr("qpos")[:] = pre["qpos"][:, b]; x.arr("qvel")[:] = pre["qvel"][:, b]; x.arr("qacc_warmstart")[:] = pre["warm"][:, b]
            x.arr("ctrl")[:nu] = c[:, b]
            x.step(1)
            qa, qk = x.arr("qacc").copy(), post["qacc"][:nv, b]
            e = np.abs(qk - qa)
            rob.append(e[:26].max() / max(1.0, np.abs(qa[:26]).max()))
            obj.append(e[26:].max() / max(1.0, np.abs(qa[26:]).max()))
            objabs.append(e[26:].max())
            if verbose and obj[-1] > 2e-2:
                k = 26 + int(np.argmax(e[26:]))
                print(f"   step {s} env {b}: obj rel {obj[-1]:.2e} dof {k}: kernel {qk[k]:+.4f} oracle {qa[k]:+.4f}; ncon {int(post['info'][1, b])} / {x.ncon}")
    rob, obj, objabs = np.array(rob), np.array(obj), np.array(objabs)
    pc = lambda a: f"p50 {np.percentile(a, 50):.1e} p99 {np.percentile(a, 99):.1e} max {a.max():.1e}"
    print(f"[{scene} cache {cache} oracle {oracle_opts}] {len(rob)} env-steps: robot {pc(rob)} | object (own scale) {pc(obj)}; above 2e-2: {(obj > 2e-2).sum()} | abs {pc(objabs)}")
    return rob, obj


if __name__ == "__main__":
    scene = sys.argv[1] if len(sys.argv) > 1 else "stretch_kitchen_robocasa"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    for cache in ([int(v) for v in sys.argv[4:]] or [1, 0]):
        run(scene, steps, B, cache, verbose=os.environ.get("V") == "1")
        if cache:   # against the oracle's twin of the keep rule (option manifold_keep): the implementation, not the rule
            run(scene, steps, B, cache, oracle_opts={"manifold_keep": 1}, verbose=os.environ.get("V") == "1")
