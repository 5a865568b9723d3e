"""Capture the window in which an env of a satellite scene gets flagged: state at the start of the 50-step window + ctrl."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stretch_mujoco_amd import StretchBatchSimulator
scene = sys.argv[1] if len(sys.argv) > 1 else "stretch_scene_sat"
B, HOLD, WIN = 4096, 50, 15
sim = StretchBatchSimulator(num_envs=B, device="cuda:0", scene=scene)
sim.start(home=False); sim.home(settle=False); sim.step(500)
cr = torch.tensor(np.asarray(sim.model["actuator_ctrlrange"]), dtype=torch.float32, device=sim.device)
g = torch.Generator(device=sim.device); g.manual_seed(1)
caps = []
for w in range(WIN):
    sim.ctrl[:] = cr[:, :1] + (cr[:, 1:] - cr[:, :1]) * torch.rand(sim.nu, B, generator=g, device=sim.device)
    pre = (sim.qpos.clone(), sim.qvel.clone(), sim.qacc_warmstart.clone(), sim.ctrl.clone(), sim.info[3].clone())
    sim.step(HOLD); torch.cuda.synchronize()
    new = ((sim.info[3] != 0) & (pre[4] == 0)).nonzero().flatten().tolist()
    for e in new[:4]:
        caps.append(dict(env=e, window=w, flags=int(sim.info[3, e]), qpos=pre[0][:, e].cpu().numpy(), qvel=pre[1][:, e].cpu().numpy(), warm=pre[2][:, e].cpu().numpy(),
                         ctrl=pre[3][:, e].cpu().numpy(), nefc=int(sim.info[0, e]), ncon=int(sim.info[1, e])))
        print("flagged env", e, "window", w, "flags", hex(int(sim.info[3, e])), "nefc", int(sim.info[0, e]), "ncon", int(sim.info[1, e]), flush=True)
np.savez(os.path.join(ROOT, "gpurun_out", f"sat_flag_capture_{scene}.npz"), caps=np.array(caps, dtype=object))
print("captured", len(caps))
