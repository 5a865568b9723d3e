"""Collision sweep of tests/test_collision_fuzz.py on the device: StretchBatchSimulator with robot-less model blobs."""
import sys, math, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_collision_fuzz as T
from stretch_mujoco_amd import StretchBatchSimulator, model_fuse as F, model_blob as B, mjcf_compiler as C
from oracle.oracle import Oracle
rng = np.random.default_rng(7)
bad = n = 0
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    a, b = ("box", "box") if trial % 2 else (rng.choice(list(T.SHAPES)), rng.choice(list(T.SHAPES)))
    ya, yb = rng.choice(T.YAWS), rng.choice(T.YAWS)
    off = [rng.choice([0, 0.05, -0.1, 0.13]), rng.choice([0, 0.05, -0.08]), 0]
    ta, sa, ha = T.SHAPES[a]; tb, sb, hb = T.SHAPES[b]
    za = 0.3; zb = za + ha + hb - 0.001
    static = rng.random() < 0.5
    A = (f'<geom type="{ta}" size="{sa}" pos="0 0 {za}" euler="0 0 {ya}"/>' if static else f'<body pos="0 0 {za}" euler="0 0 {ya}"><freejoint/><geom type="{ta}" size="{sa}" mass="1"/></body>')
    scene = ('<mujoco><compiler angle="radian"/>' + T.OPT + '<worldbody>' + A + f'<body pos="{off[0]} {off[1]} {zb}" euler="0 0 {yb}"><freejoint/><geom type="{tb}" size="{sb}" mass="1"/></body></worldbody></mujoco>')
    blob = B.dumps(F.prepare_for_kernels(C.compile_string(scene)))
    o = Oracle(blob); o.set_option("solver", 2)
    sim = StretchBatchSimulator(num_envs=1, device="cuda:0", model_blob_bytes=blob, debug=True)
    sim.start(home=False)
    sim.qpos[:, 0] = torch.tensor(o.arr("qpos"), dtype=torch.float32, device=sim.device)
    o.step(1); sim.step(1); torch.cuda.synchronize()
    D = sim.debug_layout
    nk = int(sim.info[1, 0])
    dk = np.sort(sim.debug[D["con"]:D["con"] + 8 * nk, 0].cpu().numpy().reshape(nk, 8)[:, 0])
    do = np.sort(o.arr("contact").reshape(o.ncon, -1)[:, 0]) if o.ncon else np.zeros(0)
    n += 1
    if nk != o.ncon or (nk and np.abs(dk - do).max() > 2e-5):
        bad += 1; print("MISMATCH", a, b, ya, yb, off, static, "device", nk, np.round(dk, 5), "oracle", o.ncon, np.round(do, 5))
    sim.stop()
print("configs", n, "mismatches", bad)
