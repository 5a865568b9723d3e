"""Which capacity of the 16-satellite build binds on the random-action workload: escalation off, the diagnostic flag bits per env."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stretch_mujoco_amd import StretchBatchSimulator
scene = sys.argv[1] if len(sys.argv) > 1 else "stretch_kitchen_robocasa"
B = 4096
sim = StretchBatchSimulator(num_envs=B, device="cuda:0", scene=scene); sim.start(home=False)
sim.set_option("escalate", 0)
dev = sim.device
lo = torch.tensor(sim.model["actuator_ctrlrange"][:, 0], dtype=torch.float32, device=dev).unsqueeze(1); hi = torch.tensor(sim.model["actuator_ctrlrange"][:, 1], dtype=torch.float32, device=dev).unsqueeze(1)
g = torch.Generator(device=dev).manual_seed(1234)
sim.ctrl[:] = torch.tensor([0, 0, 0.6, 0.1, 0, 0, 0, 0, 0, 0], dtype=torch.float32, device=dev).unsqueeze(1); sim.step(300)
sim.info[3].zero_()
names = {0x100: "items", 0x200: "coupled satellites", 0x400: "sat-sat rows", 0x800: "broadphase lists", 0x1000: "dense rows", 0x2000: "rows", 0x4000: "contacts", 4: "bad state"}
for w in range(10):
    sim.ctrl.copy_(lo + (hi - lo) * torch.rand(10, B, generator=g, device=dev)); sim.step(50)
torch.cuda.synchronize()
fl = sim.info[3]
print(scene, "fraction of envs that hit each capacity at least once in 500 random-action steps (escalation off):")
for bit, n in names.items():
    print(f"   {n:20s} {float(((fl & bit) != 0).float().mean()):.4f}")
print("   any", float((fl != 0).float().mean()))
