"""Capture the pre-launch state (qpos, qvel, ctrl, warm start) of envs that a 50-step launch ends with a bad-state reset
(info flag bit 2: non-finite or absurd state, [MJ] mj_checkPos / mj_checkVel) under random actions, for replay in the oracle / lane
emulator on the CPU:  python tools/gpu_badstate_capture.py [scene]  ->  gpurun_out/nan_cases.npz"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from stretch_mujoco_amd import StretchBatchSimulator
scene = sys.argv[1] if len(sys.argv) > 1 else "stretch_scene"
B = 4096
sim = StretchBatchSimulator(num_envs=B, device="cuda:0", solver="newton", scene=scene); sim.start(home=True)
dev = sim.device
g = torch.Generator(device=dev).manual_seed(99)
lo = torch.tensor(sim.model["actuator_ctrlrange"][:, 0], dtype=torch.float32, device=dev).unsqueeze(1)
hi = torch.tensor(sim.model["actuator_ctrlrange"][:, 1], dtype=torch.float32, device=dev).unsqueeze(1)
found = []
for k in range(160):
    sim.ctrl.copy_(lo + (hi - lo) * torch.rand(sim.nu, B, generator=g, device=dev))
    pre = [t.clone() for t in (sim.qpos, sim.qvel, sim.ctrl, sim.qacc_warmstart)]
    sim.info[3].zero_()
    sim.step(50)
    bad = ((sim.info[3] & 4) != 0).nonzero().flatten()
    for e in bad.tolist()[:3]:
        found.append(dict(launch=k, env=e, qpos=pre[0][:, e].cpu().numpy(), qvel=pre[1][:, e].cpu().numpy(), ctrl=pre[2][:, e].cpu().numpy(), warm=pre[3][:, e].cpu().numpy()))
    if len(found) >= 6: break
print("found", len(found), [(f["launch"], f["env"]) for f in found])
np.savez("gpurun_out/nan_cases.npz", **{f"{i}_{k}": v for i, f in enumerate(found) for k, v in f.items() if k not in ("launch", "env")})
