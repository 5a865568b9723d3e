import sys, torch, numpy as np
sys.path.insert(0, ".")
from stretch_mujoco_amd import StretchBatchSimulator
scene = sys.argv[1]; steps = int(sys.argv[2]); B = 4096
sim = StretchBatchSimulator(num_envs=B, device="cuda:0", scene=scene); sim.start(home=True)
dev = sim.device
g = torch.Generator(device=dev).manual_seed(99)
lo = torch.tensor(sim.model["actuator_ctrlrange"][:, 0], dtype=torch.float32, device=dev).unsqueeze(1)
hi = torch.tensor(sim.model["actuator_ctrlrange"][:, 1], dtype=torch.float32, device=dev).unsqueeze(1)
seen = 0
for k in range(steps // 50):
    sim.ctrl.copy_(lo + (hi - lo) * torch.rand(sim.nu, B, generator=g, device=dev))
    sim.step(50)
    z = sim.qpos[2]
    hiz = (z > 0.5).nonzero().flatten()
    if len(hiz) and seen < 6:
        for e in hiz[:2].tolist():
            print(scene, "launch", k, "env", e, "z %.2f" % float(z[e]), "vz %.1f" % float(sim.qvel[2, e]), "flags", hex(int(sim.info[3, e])), "|qvel|max %.1f" % float(sim.qvel[:, e].abs().max()), "ncon", int(sim.info[1, e]))
        seen += 1
print(scene, "done: max z", float(sim.qpos[2].max()), "resets", float(((sim.info[3] & 4) != 0).float().mean()))
