"""Floating-point operations per env-step of the bench workload, from the instrumented fp64 oracle (oracle/smj_oracle.c: FL()
counters in the vector primitives and at the dense loops of every stage; an FMA counts 2).  Writes
profiles/flops_per_env_step.json, the constant bench.py's fp32 figure uses.  TEST INFRASTRUCTURE (runs the oracle).

    python tools/flop_count.py [envs] [windows] [scene ...]      (scenes given: only those are counted, merged into the file)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle, lib  # noqa: E402
from stretch_mujoco_amd import model_blob  # noqa: E402


def main():
    envs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    windows = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    path = os.path.join(ROOT, "profiles", "flops_per_env_step.json")
    out = json.load(open(path))["scenes"] if len(sys.argv) > 3 and os.path.exists(path) else {}
    # the kitchen at Robocasa scale: also under PGS (north_star's solver for config 4) -- key "<scene>:pgs"
    for scene in (sys.argv[3:] or ["stretch_empty", "stretch_kitchen_standin", "stretch_scene", "stretch_kitchen4", "stretch_kitchen_robocasa", "stretch_kitchen_robocasa:pgs"]):
        key, scene, solver = scene, scene.split(":")[0], (0 if scene.endswith(":pgs") else 2)
        blob = open(os.path.join(ROOT, "stretch_mujoco_amd", "models", scene + ".smjb"), "rb").read()
        m = model_blob.loads(blob)
        cr = np.asarray(m["actuator_ctrlrange"])
        total, steps, iters = 0, 0, 0
        for e in range(envs):
            o = Oracle(blob)
            o.set_option("solver", solver)
            nu = o.dim("nu")
            o.arr("ctrl")[:nu] = [0, 0, 0.6, 0.1, 0, 0, 0, 0, 0, 0][:nu]
            o.step(500)
            rng = np.random.default_rng(1234 + e)
            lib().smjo_flops(1)
            for w in range(windows):
                o.arr("ctrl")[:nu] = cr[:, 0] + (cr[:, 1] - cr[:, 0]) * rng.random(nu)
                for _ in range(50):
                    o.step(1)
                    iters += int(o.iarr("solver_niter")[0])
            total += lib().smjo_flops(1)
            steps += 50 * windows
        out[key] = {"flops_per_env_step": total / steps, ("pgs_sweeps_per_step" if solver == 0 else "newton_iterations_per_step"): iters / steps, "env_steps_counted": steps}
        print(key, out[key])
    res = {"flops_per_env_step": out["stretch_empty"]["flops_per_env_step"], "scenes": out,
           "source": "tools/flop_count.py: instrumented fp64 oracle (FL() counters: vector primitives + dense loops; FMA = 2), bench workload "
                     "(random ctrl every 50 steps, Newton, multiccd on), mean over envs and steps"}
    with open(path, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
