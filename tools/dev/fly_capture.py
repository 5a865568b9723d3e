import sys, torch, numpy as np
sys.path.insert(0, ".")
from stretch_mujoco_amd import StretchBatchSimulator
scene, env, k0, k1 = "stretch_kitchen4_sat", 3699, 618, 628
B = 4096
sim = StretchBatchSimulator(num_envs=B, device="cuda:0", scene=scene); sim.start(home=True)
dev = sim.device
g = torch.Generator(device=dev).manual_seed(99)
lo = torch.tensor(sim.model["actuator_ctrlrange"][:, 0], dtype=torch.float32, device=dev).unsqueeze(1)
hi = torch.tensor(sim.model["actuator_ctrlrange"][:, 1], dtype=torch.float32, device=dev).unsqueeze(1)
rec = {}
for k in range(k1 + 1):
    sim.ctrl.copy_(lo + (hi - lo) * torch.rand(sim.nu, B, generator=g, device=dev))
    if k >= k0:
        rec[f"qpos{k}"] = sim.qpos[:, env].cpu().numpy(); rec[f"qvel{k}"] = sim.qvel[:, env].cpu().numpy(); rec[f"warm{k}"] = sim.qacc_warmstart[:, env].cpu().numpy(); rec[f"ctrl{k}"] = sim.ctrl[:, env].cpu().numpy()
    sim.step(50)
    if k >= k0: print(k, "z after", float(sim.qpos[2, env]), "flags", hex(int(sim.info[3, env])))
np.savez("gpurun_out/fly_capture.npz", **rec)
