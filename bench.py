#!/usr/bin/env python
"""Throughput bench of the batched physics path (contract: see the task's bench.py section).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one physics step (mj_step) of every environment on every rank.  Workload: 4096
Stretch envs per GPU (weak scaling), stretch.xml + ground plane, home keyframe settled, then synthetic random
actions drawn uniformly in ctrlrange every 50 steps.  Envs are independent: no data-path collective; RCCL is
used once at the end to gather per-env returns (and for the barrier / max-over-ranks timing).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_ENV_STEP = 672.0  # physics-only algorithmic state traffic
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(blob: bytes, ctrl_script: np.ndarray, hold: int, seconds: float = 12.0, solver: str = "newton"):
    """fp64 CPU restatement (oracle, 'port') timed single-threaded on this host on a bounded sample."""
    from oracle.oracle import Oracle

    o = Oracle(blob)
    o.set_option("solver", 2 if solver == "newton" else 0)
    o.arr("ctrl")[:] = [0, 0, 0.6, 0.1, 0, 0, 0, 0, 0, 0]
    o.step(500)
    t0 = time.perf_counter()
    n = 0
    i = 0
    while time.perf_counter() - t0 < seconds:
        o.arr("ctrl")[:] = ctrl_script[i % len(ctrl_script)]
        o.step(hold)
        n += hold
        i += 1
    dt = time.perf_counter() - t0
    return dict(value=n / dt, unit="env-steps/s", cores=1, kind="port",
                sample=f"1 env, {n} steps, {solver} solver, same scene and action schedule "
                       f"(stand-in fp64 CPU restatement, not MuJoCo)")


def other_configs(B, dev, hold, solver):
    """Short single-GPU runs of the other BASELINE.json configs (reported beside the headline line, never as `value`)."""
    from stretch_mujoco_amd import StretchBatchSimulator, StretchSensors
    from stretch_mujoco_amd.enums import StretchCameras

    res = {}

    def rollout(sim, n, chunk):
        gen = torch.Generator(device=dev).manual_seed(99)
        lo = torch.tensor(sim.model["actuator_ctrlrange"][:, 0], dtype=torch.float32, device=dev).unsqueeze(1)
        hi = torch.tensor(sim.model["actuator_ctrlrange"][:, 1], dtype=torch.float32, device=dev).unsqueeze(1)
        sim.ctrl[:] = torch.tensor(sim.model["key_ctrl"][0, : sim.nu], dtype=torch.float32, device=dev).unsqueeze(1)
        sim.step(500)
        sim.ctrl.copy_(lo + (hi - lo) * torch.rand(sim.nu, B, generator=gen, device=dev))
        sim.step(chunk)
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        d = 0
        while d < n:
            if d % hold == 0:
                sim.ctrl.copy_(lo + (hi - lo) * torch.rand(sim.nu, B, generator=gen, device=dev))
            sim.step(chunk)
            d += chunk
        torch.cuda.synchronize(dev)
        return B * d / (time.perf_counter() - t)

    # config 3: + joint readout, IMU, 2-D lidar; at 15 Hz sim-time (every 33 steps) and every step
    sim = StretchBatchSimulator(num_envs=B, device=str(dev), solver=solver, sensors_to_use=StretchSensors.all())
    sim.start(home=False)
    res["lidar_imu_every_33_steps"] = {"value": rollout(sim, 330, 33), "unit": "env-steps/s"}
    res["lidar_imu_every_step"] = {"value": rollout(sim, 50, 1), "unit": "env-steps/s"}
    sim.stop()
    # config 4 stand-in: 24 static kitchen fixtures around the robot (no free objects), physics only
    sim = StretchBatchSimulator(num_envs=B, device=str(dev), solver=solver, scene="stretch_kitchen_standin")
    sim.start(home=False)
    res["kitchen_standin_physics"] = {"value": rollout(sim, 200, hold), "unit": "env-steps/s",
                                      "overflow_flags": int(sim.info[3].max().item())}
    sim.stop()
    # config 5 ingredient: both depth cameras, kitchen stand-in, rendered from the poses of the last step
    sim = StretchBatchSimulator(num_envs=B, device=str(dev), solver=solver, scene="stretch_kitchen_standin",
                                cameras_to_use=StretchCameras.depth())
    sim.start(home=False)
    rollout(sim, hold, hold)
    sim.pull_camera_data()
    torch.cuda.synchronize(dev)
    t = time.perf_counter()
    for _ in range(3):
        sim.pull_camera_data()
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t) / 3
    rays = B * (270 * 480 + 240 * 424)
    res["depth_both_cameras"] = {"ms_per_render": dt * 1e3, "rays_per_s": rays / dt, "bytes_written_per_s": 4 * rays / dt,
                                 "note": "d405 270x480 + d435i 240x424 per env, kitchen stand-in; at 30 Hz sim-time one render per 17 steps"}
    # config 5 shape on one GPU: kitchen stand-in, both depth cameras every 17 steps (30 Hz sim-time), physics in between
    torch.cuda.synchronize(dev)
    t = time.perf_counter()
    n = 0
    for _ in range(6):
        sim.step(17)
        sim.pull_camera_data()
        n += 17
    torch.cuda.synchronize(dev)
    res["kitchen_with_depth_30hz"] = {"value": B * n / (time.perf_counter() - t), "unit": "env-steps/s",
                                      "note": "whole loop: 17 physics steps + one render of both depth cameras, repeated"}
    sim.stop()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--hold", type=int, default=50, help="physics steps per launch / per random action")
    ap.add_argument("--solver", choices=["newton", "pgs"], default="newton",
                    help="newton = the reference model's own solver (stretch.xml names none -> MuJoCo default); "
                         "pgs = the solver named by BASELINE.json north_star")
    ap.add_argument("--no-second-solver", action="store_true", help="skip the short run of the other solver")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the short runs of the other BASELINE.json configs")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist_on = world > 1
    if dist_on:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from stretch_mujoco_amd import StretchBatchSimulator
    from stretch_mujoco_amd.parallel import gather_returns

    B = args.envs_per_gpu
    sim = StretchBatchSimulator(num_envs=B, device=str(dev), solver=args.solver)
    sim.start(home=False)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    lo = torch.tensor(sim.model["actuator_ctrlrange"][:, 0], dtype=torch.float32, device=dev).unsqueeze(1)
    hi = torch.tensor(sim.model["actuator_ctrlrange"][:, 1], dtype=torch.float32, device=dev).unsqueeze(1)

    def random_action():
        sim.ctrl.copy_(lo + (hi - lo) * torch.rand(sim.nu, B, generator=gen, device=dev))

    # settle at the home keyframe
    sim.ctrl[:] = torch.tensor(sim.model["key_ctrl"][0, : sim.nu], dtype=torch.float32, device=dev).unsqueeze(1)
    hold = max(1, args.hold)
    all_events = []   # every smj_step_kernel launch of this process (what `rocprofv3 --stats` averages over)

    def timed_step(k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sim.step(k)
        e1.record()
        all_events.append((e0, e1, k))
        return e0, e1

    for _ in range(500 // hold):
        timed_step(hold)
    done = 0
    while done < args.warmup:
        k = min(hold, args.warmup - done)
        random_action()
        timed_step(k)
        done += k
    returns = torch.zeros(B, device=dev)
    events = []

    def barrier():
        torch.cuda.synchronize(dev)
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize(dev)

    barrier()
    t0 = time.perf_counter()
    done = 0
    while done < args.steps:
        k = min(hold, args.steps - done)
        random_action()
        e0, e1 = timed_step(k)
        events.append((e0, e1, k))
        returns += sim.base_pose[0]  # synthetic per-env return: accumulated forward displacement
        done += k
    barrier()
    dt = time.perf_counter() - t0
    if dist_on:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    all_returns = gather_returns(returns)  # RCCL all-gather of per-env returns (the only collective)
    flags = int(sim.info[3].max().item())
    flagged = float((sim.info[3] != 0).float().mean().item())   # envs in which some step since reset exceeded the 80-row / 16-contact capacity (flags are sticky until reset)
    kern_ms = sum(a.elapsed_time(b) for a, b, _ in events)
    kern_steps = sum(k for _, _, k in events)
    total_env_steps = float(B) * world * args.steps
    value = total_env_steps / dt
    if rank == 0:
        per_launch_envsteps = B * hold
        avg_launch_s = (kern_ms / 1e3) / max(1, len(events))
        achieved = per_launch_envsteps * BYTES_PER_ENV_STEP / avg_launch_s / 1e9
        all_ms = [a.elapsed_time(b) for a, b, _ in all_events][1:]   # the first launch also pays the one-time code-object load
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_path):   # HBM bytes per launch measured with rocprofv3 --pmc on this same command
            with open(pmc_path) as f:
                pm = json.load(f)
            if pm.get("envs_per_gpu") == B and pm.get("steps_per_launch") == hold and pm.get("solver") == args.solver:
                traffic = pm["hbm_bytes_per_launch"]
        out = {
            "metric": "env-steps/sec (whole node), 4096 parallel Stretch envs per MI355X",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{B} parallel Stretch envs per GPU, stretch.xml + ground plane (empty scene), "
                                   f"physics-only, random ctrl in ctrlrange every {hold} steps, {args.solver} solver "
                                   f"(iterations<=100, tol 1e-8), elliptic cones impratio 20, implicitfast, dt=0.002",
                       "solver": args.solver,
                       "envs_per_gpu": B, "steps_per_launch": hold, "parallelism": f"env-sharded x{world}",
                       "returns_gathered": int(all_returns.numel()), "overflow_flags": flags,
                       "envs_over_capacity_since_reset": flagged},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "all_launches": {"count": len(all_ms), "avg_ms": sum(all_ms) / len(all_ms), "ms": [round(x, 2) for x in all_ms],
                                          "note": "settle + warmup + timed launches except the very first (code-object load): the population "
                                                  "rocprofv3 --stats averages (profiles/r01_rocprof_summary.md)"},
                         "kernel": "smj_step_kernel", "avg_launch_ms": avg_launch_s * 1e3,
                         "us_per_env_step_latency": avg_launch_s * 1e6 / hold,
                         "note": "algorithmic bytes = 672 B/env-step x envs x steps per launch; the path is "
                                 "bound by the instruction issue of one wavefront per SIMD (8.3 cycles per instruction), HBM does not bind "
                                 "(DESIGN.md section 4)"},
        }
        if not args.no_second_solver and world == 1:
            other = "pgs" if args.solver == "newton" else "newton"
            sim.set_option("solver", {"pgs": 0, "newton": 2}[other])
            n2 = min(args.steps, 100)
            random_action(); sim.step(hold)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            d2 = 0
            while d2 < n2:
                k = min(hold, n2 - d2)
                random_action(); sim.step(k); d2 += k
            torch.cuda.synchronize(dev)
            out["other_solver"] = {"solver": other, "value": B * n2 / (time.perf_counter() - t1), "unit": "env-steps/s",
                                   "n_gpus": 1, "steps": n2, "note": "rank 0 only, same workload, measured after the timed region"}
        if not args.no_cpu_baseline and world == 1:   # the contract: rank 0 at N=1 only
            rng = np.random.default_rng(1234)
            cr = np.asarray(sim.model["actuator_ctrlrange"])
            script = cr[:, 0] + (cr[:, 1] - cr[:, 0]) * rng.random((64, sim.nu))
            out["cpu_baseline"] = cpu_baseline(sim._blob, script, hold, args.cpu_seconds, args.solver)
            out["cpu_baseline"]["host_cpus"] = os.cpu_count()
        import __graft_entry__ as _ge
        out["parity_oracle"] = _ge.mujoco_status()
        if not args.no_extra and world == 1:
            sim.stop()
            out["other_configs"] = other_configs(B, dev, hold, args.solver)
        print(json.dumps(out))
    sim.stop()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
