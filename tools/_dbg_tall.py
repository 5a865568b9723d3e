import sys, numpy as np, torch
sys.path.insert(0, ".")
from stretch_mujoco_amd import StretchBatchSimulator, model_blob
mode = sys.argv[1]
if mode == "empty_tall":
    m = model_blob.loads(open("stretch_mujoco_amd/models/stretch_empty.smjb", "rb").read())
    m["k_capacity_hint"] = np.array([1], np.int32)
    sim = StretchBatchSimulator(num_envs=8, device="cuda:0", model_blob_bytes=model_blob.dumps(m))
elif mode == "kitchen_pgs":
    sim = StretchBatchSimulator(num_envs=8, device="cuda:0", scene="stretch_kitchen_standin", solver="pgs")
else:
    sim = StretchBatchSimulator(num_envs=8, device="cuda:0", scene="stretch_kitchen_standin")
sim.start(home=False)
print(mode, "variant caps", sim.nv_max, sim.nefc_max, sim.ncon_max, flush=True)
sim.ctrl[:] = torch.tensor([0, 0, 0.6, 0.5, 0, 0, 0, 0, 0, 0], dtype=torch.float32, device=sim.device).unsqueeze(1)
for k in range(30):
    sim.step(1); torch.cuda.synchronize()
    print(k, sim.info[:, 0].tolist(), flush=True)
sim.step(50); torch.cuda.synchronize(); print("50 ok", sim.info[:, 0].tolist(), flush=True)
