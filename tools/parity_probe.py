"""CPU probe of step-level parity: the kernel source through the lane emulator (tests/emul, fp32) against the fp64 oracle on
STATE-SYNCHRONISED steps of a random-action rollout (the oracle's state is copied into the emulator before every step), so
that every discrepancy is attributed to the step that produced it.  Prints the steps where the contact lists differ or the
one-step error of qvel exceeds a threshold, with both contact lists.  Test infrastructure (uses oracle/ and tests/emul).

    python tools/parity_probe.py [scene] [envs] [windows] [seed]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from emul.emul import Emul  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

import stretch_mujoco_amd.model_blob as mb  # noqa: E402


def contacts_e(e):
    from stretch_mujoco_amd.lib import debug_layout

    n = int(e.info[1, 0])
    D = debug_layout(e.nvp, e.ncon_max)
    c = e.debug[D["con"]:D["con"] + 8 * e.ncon_max, 0].reshape(-1, 8)[:n]
    code = c[:, 7].astype(np.int64)
    return [(int((k >> 4) & 1023), int(k >> 14), float(d)) for k, d in zip(code, c[:, 0])], c


def contacts_o(o):
    n = o.ncon
    c = o.arr("contact").reshape(n, -1) if n else np.zeros((0, 29))
    ints = [c[k, -2:].copy().view(np.int32) for k in range(n)]
    return [(int(i[1]), int(i[2]), float(c[k, 0])) for k, i in enumerate(ints)], c


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "stretch_empty"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 7
    blob = open(os.path.join(ROOT, "stretch_mujoco_amd", "models", scene + ".smjb"), "rb").read()
    model = mb.loads(blob)
    cr = np.asarray(model["actuator_ctrlrange"])
    o0 = Oracle(blob)
    nq, nv, nu = o0.dim("nq"), o0.dim("nv"), o0.dim("nu")
    rng = np.random.default_rng(seed)
    ctrls = [cr[:, 0][:, None] + (cr[:, 1] - cr[:, 0])[:, None] * rng.random((nu, B)) for _ in range(W)]
    events = 0
    for env in range(B):
        o = Oracle(blob)
        o.set_option("solver", 2)
        e = Emul(blob, dict(nq=nq, nv=nv, nu=nu, nlidar=360), num_envs=1, debug=True)
        e.set_option("solver", 2)
        o.arr("ctrl")[:nu] = [0, 0, 0.6, 0.1, 0, 0, 0, 0, 0, 0][:nu]
        o.step(500)
        for w in range(W):
            o.arr("ctrl")[:nu] = ctrls[w][:, env].astype(np.float32)
            for s in range(50):
                e.ctrl[:, 0] = o.arr("ctrl")[:nu]; e.qpos[:, 0] = o.arr("qpos"); e.qvel[:, 0] = o.arr("qvel"); e.warm[:, 0] = o.arr("qacc_warmstart")
                e.step(1)
                o.step(1)
                le, ce = contacts_e(e)
                lo, co = contacts_o(o)
                dv = np.abs(e.qvel[:, 0] - o.arr("qvel")).max()
                pairs_e, pairs_o = sorted((a, b) for a, b, _ in le), sorted((a, b) for a, b, _ in lo)
                if pairs_e != pairs_o or dv > 2e-3:
                    events += 1
                    print(f"env {env} window {w} step {s}: one-step |dqvel| {dv:.2e}  ncon e/o {len(le)}/{len(lo)} flags {int(e.info[3, 0])}")
                    for a, b, d in le:
                        if (a, b) not in pairs_o:
                            print(f"    kernel-only contact geoms ({a},{b}) dist {d:.6f}")
                    for a, b, d in lo:
                        if (a, b) not in pairs_e:
                            print(f"    oracle-only contact geoms ({a},{b}) dist {d:.6f}")
    print("events:", events)


if __name__ == "__main__":
    main()
