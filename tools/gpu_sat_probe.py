"""GPU probe of the satellite builds: parity (state-synchronised vs the fp64 oracle) and throughput against the dense builds.
    python tools/gpu_sat_probe.py [parity|speed|both]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from stretch_mujoco_amd import StretchBatchSimulator, model_blob as mb

what = sys.argv[1] if len(sys.argv) > 1 else "both"
MODELS = os.path.join(ROOT, "stretch_mujoco_amd", "models")
if what in ("parity", "both"):
    import rollout_common as rc
    for scene in ("stretch_scene_sat", "stretch_kitchen4_sat", "stretch_kitchen_export_sat", "stretch_kitchen_robocasa"):
        blob = open(f"{MODELS}/{scene}.smjb", "rb").read()
        be = rc.HipBackend(scene, 8)
        rel, events = rc.state_synchronised(be, blob, mb.loads(blob), 8, 6, seed=3)
        be.close()
        c = rc.state_synchronised.contacts
        bad = [e for e in events if not e["explained"]]
        print(f"[{scene}] state-synchronised {len(rel)} env-steps: rel qacc p50 {np.percentile(rel, 50):.1e} p99 {np.percentile(rel, 99):.1e} max {rel.max():.1e}; "
              f"events {len(events)} unexplained {len(bad)}; contacts {c['n']} mismatched steps {c['mismatched_steps']}", flush=True)
        for e in bad: print("   ", e)
if what in ("speed", "both"):
    B, HOLD, WIN = 4096, 50, 12
    for scene in ("stretch_scene", "stretch_scene_sat", "stretch_kitchen4", "stretch_kitchen4_sat", "stretch_kitchen_export", "stretch_kitchen_export_sat", "stretch_kitchen_robocasa"):
        sim = StretchBatchSimulator(num_envs=B, device="cuda:0", scene=scene)
        sim.start(home=False)
        sim.home(settle=False)
        sim.step(500)
        cr = torch.tensor(np.asarray(sim.model["actuator_ctrlrange"]), dtype=torch.float32, device=sim.device)
        g = torch.Generator(device=sim.device); g.manual_seed(1)
        def act():
            sim.ctrl[:] = cr[:, :1] + (cr[:, 1:] - cr[:, :1]) * torch.rand(sim.nu, B, generator=g, device=sim.device)
        for _ in range(3): act(); sim.step(HOLD)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(WIN): act(); sim.step(HOLD)
        torch.cuda.synchronize(); dt = time.time() - t0
        fl = sim.info[3]
        print(f"[{scene}] {B * HOLD * WIN / dt / 1e6:.2f} M env-steps/s; flagged envs {int((fl != 0).sum())} (bits {hex(sum(int(((fl >> b) & 1).any()) << b for b in range(16)))}); nefc mean {float(sim.info[0].float().mean()):.1f} max {int(sim.info[0].max())}; ncon max {int(sim.info[1].max())}", flush=True)
        sim.stop()
