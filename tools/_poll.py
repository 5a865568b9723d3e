import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from stretch_mujoco_amd import StretchBatchSimulator
from oracle.oracle import Oracle
from conftest import HOME_CTRL, MODELS
import os
blob = open(os.path.join(MODELS, "stretch_empty.smjb"), "rb").read()
o = Oracle(blob); o.set_option("solver", 2); o.reset()
ctrl = [0, 0, 0.0, 0.3, 0, -1.57, 0, 0, 0, 0]
o.arr("ctrl")[:10] = ctrl
o.step(8)
q0, v0, w0 = o.arr("qpos").copy(), o.arr("qvel").copy(), o.arr("qacc_warmstart").copy()
rows = []
for k in range(32):
    o.step(1); rows.append((o.nefc, o.ncon))
print("oracle rows", rows)
B = 1100
res = {}
for name, opts in (("sweep", dict(pipeline=0)), ("pipe+sweep", dict(pipeline=5, pollers=0)), ("pollers", dict(pipeline=5, pollers=8)), ("pollers10", dict(pipeline=10, pollers=16))):
    sim = StretchBatchSimulator(num_envs=B, device="cuda:0", solver="newton"); sim.start(home=False)
    for k, v in opts.items(): sim.set_option(k, v)
    sim.ctrl[:] = torch.tensor(HOME_CTRL, dtype=torch.float32, device=sim.device).unsqueeze(1)
    idx = torch.arange(0, B, 7, device=sim.device)
    sim.ctrl[:, idx] = torch.tensor(ctrl, dtype=torch.float32, device=sim.device).unsqueeze(1)
    sim.qpos[:, idx] = torch.tensor(q0, dtype=torch.float32, device=sim.device).unsqueeze(1)
    sim.qvel[:, idx] = torch.tensor(v0, dtype=torch.float32, device=sim.device).unsqueeze(1)
    sim.qacc_warmstart[:, idx] = torch.tensor(w0, dtype=torch.float32, device=sim.device).unsqueeze(1)
    sim.step(32)
    torch.cuda.synchronize()
    qv = sim.qvel[:, idx].cpu().numpy(); qp = sim.qpos[:, idx].cpu().numpy()
    res[name] = (qp, qv)
    print(name, "flags", int(sim.info[3].max()), "nstep", int(sim.nstep.min()), int(sim.nstep.max()),
          "spread among copies %.3g" % np.abs(qv - qv[:, :1]).max(),
          "vs oracle dq %.3g dv %.3g" % (np.abs(qp[:, 0] - o.arr("qpos")).max(), np.abs(qv[:, 0] - o.arr("qvel")).max()),
          "others moved", float(sim.qvel[:, 1].abs().max()))
    sim.stop()
for n in res:
    print(n, "vs sweep dq %.3g dv %.3g" % (np.abs(res[n][0] - res["sweep"][0]).max(), np.abs(res[n][1] - res["sweep"][1]).max()))
