"""Constraint-row / contact demand of the headline random-action workload, from the capacity-free fp64 oracle (CPU).
Usage: python tools/row_demand.py [envs] [windows]   -- prints percentiles and the share of 50-step windows above each capacity."""
import os, sys
import numpy as np
from multiprocessing import Pool

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(args):
    seed, windows = args
    from oracle.oracle import Oracle
    from stretch_mujoco_amd import model_blob
    blob = open(os.path.join(ROOT, "stretch_mujoco_amd/models/stretch_empty.smjb"), "rb").read()
    m = model_blob.loads(blob)
    o = Oracle(blob)
    o.set_option("solver", 2)
    rng = np.random.default_rng(seed)
    lo, hi = m["actuator_ctrlrange"][:10, 0], m["actuator_ctrlrange"][:10, 1]
    o.reset()
    o.arr("ctrl")[:10] = m["key_ctrl"][0, :10]
    o.step(500)
    out = []
    for w in range(windows):
        o.arr("ctrl")[:10] = lo + (hi - lo) * rng.random(10)
        ne = nc = 0
        for _ in range(50):
            o.step(1)
            ne = max(ne, o.nefc); nc = max(nc, o.ncon)
        out.append((ne, nc))
    return out


if __name__ == "__main__":
    envs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    windows = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    with Pool(8) as p:
        res = p.map(run, [(1000 + i, windows) for i in range(envs)])
    a = np.array(res).reshape(-1, 2)
    print("windows:", len(a))
    for name, col, caps in (("nefc", 0, (64, 80, 96, 112, 128)), ("ncon", 1, (16, 20, 24, 32))):
        v = a[:, col]
        print(name, "p50/p90/p99/max:", np.percentile(v, 50), np.percentile(v, 90), np.percentile(v, 99), v.max(),
              " share of windows above capacity:", {c: round(float((v > c).mean()), 4) for c in caps})
