import sys, numpy as np, torch
sys.path.insert(0, ".")
from stretch_mujoco_amd import StretchBatchSimulator
scene = "stretch_kitchen4_sat"; B = 256
sims = []
for two in (1, 0):
    sim = StretchBatchSimulator(num_envs=B, device="cuda:0", scene=scene, solver="pgs")
    sim.start(home=False); sim.set_option("pgs_two_waves", two); sim.home(settle=False); sims.append(sim)
a, b = sims
a.step(50)
cr = torch.tensor(np.asarray(a.model["actuator_ctrlrange"]), dtype=torch.float32, device=a.device)
g = torch.Generator(device=a.device); g.manual_seed(11)
for w in range(2):
    a.ctrl[:] = cr[:, :1] + (cr[:, 1:] - cr[:, :1]) * torch.rand(a.nu, B, generator=g, device=a.device)
    for k in range(6):
        a.step(3)
        b.qpos[:] = a.qpos; b.qvel[:] = a.qvel; b.qacc_warmstart[:] = a.qacc_warmstart; b.ctrl[:] = a.ctrl
        a.step(1); b.step(1)
        torch.cuda.synchronize()
        sc = 1.0 + a.qvel.abs().amax(0)
        d = ((a.qvel - b.qvel).abs().amax(0) / sc)
        bad = (d > 1e-3).nonzero().flatten().tolist()
        print(w, k, "max %.2e med %.2e nbad %d" % (d.max(), d.median(), len(bad)), bad[:6], "flags a", a.info[3, bad[:4]].tolist(), "b", b.info[3, bad[:4]].tolist(), "nefc", a.info[0, bad[:4]].tolist(), b.info[0, bad[:4]].tolist(), "nstep", a.nstep[bad[:2]].tolist(), b.nstep[bad[:2]].tolist())
