"""Throughput of the settled and the random-action regime (and an md5 of the final state) for a set of smj_set_option values and a scene:
   python tools/gpu_options_probe.py [scene=stretch_kitchen_standin] [envs=16384] [pipeline=0] [pollers=-2] [multi_serial=1] ...  (SMJ_LIB_PATH selects another build)"""
import sys, time, torch
sys.path.insert(0, ".")
from stretch_mujoco_amd import StretchBatchSimulator
B = 4096
opts = {a.split("=")[0]: a.split("=")[1] for a in sys.argv[1:]}
scene = opts.pop("scene", None)
B = int(opts.pop("envs", B))
opts = {k: float(v) for k, v in opts.items()}
sim = StretchBatchSimulator(num_envs=B, device='cuda:0', **({"scene": scene} if scene else {})); sim.start(home=False)
for k, v in opts.items(): sim.set_option(k, v)
dev = sim.device
lo = torch.tensor(sim.model["actuator_ctrlrange"][:, 0], dtype=torch.float32, device=dev).unsqueeze(1); hi = torch.tensor(sim.model["actuator_ctrlrange"][:, 1], dtype=torch.float32, device=dev).unsqueeze(1)
g = torch.Generator(device=dev).manual_seed(1234)
sim.ctrl[:] = torch.tensor([0, 0, 0.6, 0.1, 0, 0, 0, 0, 0, 0], dtype=torch.float32, device=dev).unsqueeze(1); sim.step(300)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(6): sim.step(50)
torch.cuda.synchronize(); settled = B * 300 / (time.perf_counter() - t) / 1e6
for _ in range(4): sim.ctrl.copy_(lo + (hi - lo) * torch.rand(10, B, generator=g, device=dev)); sim.step(50)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): sim.ctrl.copy_(lo + (hi - lo) * torch.rand(10, B, generator=g, device=dev)); sim.step(50)
torch.cuda.synchronize(); dt = time.perf_counter() - t
import os, hashlib
h = hashlib.md5(torch.cat([sim.qpos.flatten(), sim.qvel.flatten()]).cpu().numpy().tobytes()).hexdigest()[:10]
print(h, os.environ.get("SMJ_LIB_PATH", "default")[-24:], scene, opts, 'settled %.2f M, random %.2f M env-steps/s' % (settled, B * 500 / dt / 1e6), 'flagged', float((sim.info[3] != 0).float().mean()), 'solver iterations of the last step: mean %.1f, at the cap of 100: %.3f' % (float(sim.info[2].float().mean()), float((sim.info[2] >= 100).float().mean())), flush=True)
