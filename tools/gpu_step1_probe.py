"""step(1) and step(2) calls on the headline workload under a set of options (what an RL loop that acts every step pays):
   python tools/gpu_step1_probe.py [envs=4096] [balance_min=4] ..."""
import sys, time
import torch
sys.path.insert(0, ".")
from stretch_mujoco_amd import StretchBatchSimulator
opts = {a.split("=")[0]: a.split("=")[1] for a in sys.argv[1:]}
B = int(opts.pop("envs", 4096)); scene = opts.pop("scene", None)
dev = torch.device("cuda:0")
sim = StretchBatchSimulator(num_envs=B, device=str(dev), solver="newton", **({"scene": scene} if scene else {}))
sim.start(home=False)
for k, v in opts.items(): sim.set_option(k, float(v))
lo = torch.tensor(sim.model["actuator_ctrlrange"][:, 0], dtype=torch.float32, device=dev).unsqueeze(1)
hi = torch.tensor(sim.model["actuator_ctrlrange"][:, 1], dtype=torch.float32, device=dev).unsqueeze(1)
g = torch.Generator(device=dev); g.manual_seed(1)
def act(): sim.ctrl.copy_(lo + (hi - lo) * torch.rand(sim.ctrl.shape, device=dev, generator=g))
sim.step(300)
for _ in range(6): act(); sim.step(50)
for n in (1, 2, 5):
    total = 0; torch.cuda.synchronize(); t = time.perf_counter()
    while total < 400:
        if total % 50 < n: act()
        sim.step(n); total += n
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(opts, "step(%d): %.3f ms per call, %.2f M env-steps/s" % (n, dt / (total / n) * 1e3, B * total / dt / 1e6), flush=True)
