"""Cost of the lidar readout: step(1) with IMU + lidar every step (config 3 as mj_step has it) and every 33 steps, with and without the
scan-plane cull (option lidar_cull), empty scene and kitchen stand-in.   python tools/gpu_lidar_probe.py [envs]"""
import sys, time, torch
sys.path.insert(0, ".")
from stretch_mujoco_amd import StretchBatchSimulator, StretchSensors
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for scene in ("stretch_empty", "stretch_kitchen_standin", "stretch_kitchen_robocasa"):
    for cull in (1, 0):
        sim = StretchBatchSimulator(num_envs=B, device="cuda:0", scene=scene, sensors_to_use=StretchSensors.all()); sim.start(home=False)
        sim.set_option("lidar_cull", cull)
        dev = sim.device
        lo = torch.tensor(sim.model["actuator_ctrlrange"][:, 0], dtype=torch.float32, device=dev).unsqueeze(1); hi = torch.tensor(sim.model["actuator_ctrlrange"][:, 1], dtype=torch.float32, device=dev).unsqueeze(1)
        g = torch.Generator(device=dev).manual_seed(1)
        sim.ctrl[:] = torch.tensor([0, 0, 0.6, 0.1, 0, 0, 0, 0, 0, 0], dtype=torch.float32, device=dev).unsqueeze(1); sim.step(300)
        sim.ctrl.copy_(lo + (hi - lo) * torch.rand(10, B, generator=g, device=dev)); sim.step(50)
        out = []
        for n, reps in ((1, 100), (33, 10)):
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(reps): sim.step(n)
            torch.cuda.synchronize(); dt = time.perf_counter() - t
            out.append(f"step({n}) + readout: {dt / reps * 1e3:.3f} ms per call, {B * n * reps / dt / 1e6:.2f} M env-steps/s")
        print(scene, "lidar_cull", cull, "|", " | ".join(out), flush=True)
        sim.stop()
