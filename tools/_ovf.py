import sys, torch
sys.path.insert(0, ".")
from stretch_mujoco_amd import StretchBatchSimulator
for ser in (1, 0):
    sim = StretchBatchSimulator(num_envs=4, device="cuda:0", solver="pgs"); sim.start(home=False)
    sim.set_option("multi_serial", ser)
    sim.ctrl[:] = torch.tensor([0, 0, 0.0, 0.3, 0, -1.57, 0, 0, 0, 0], dtype=torch.float32, device=sim.device).unsqueeze(1)
    hist = []
    for k in range(15):
        sim.step(100); torch.cuda.synchronize(); hist.append(tuple(int(v) for v in sim.info[:, 0].tolist()))
    print(ser, hist, float(sim.qpos[9, 0]), flush=True)
    sim.stop()
