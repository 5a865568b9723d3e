#!/usr/bin/env python
"""Throughput bench of the batched physics path (contract: see the task's bench.py section).

    python bench.py --gpus N --steps K --warmup W            (N > 1 without WORLD_SIZE: re-executes itself under
                                                               torch.distributed.run, one rank per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one physics step (mj_step) of every environment on every rank.  Workload: Stretch envs,
stretch.xml + ground plane, home keyframe settled (500 steps), then synthetic random actions drawn uniformly in ctrlrange
every 50 steps.  Before anything is timed the rollout runs 200 random-action steps so that the timed region is the steady
state of that workload whatever --warmup / --steps are; the W warm-up steps and the K timed steps continue the same
50-step action schedule (a launch never spans an action change, so K < 50 simply times a shorter launch).
--scaling weak (default): 4096 envs per GPU.  --scaling strong: 4096 envs in total (BASELINE.json's metric), split over
the ranks.  Envs are independent: no data-path collective; RCCL is used once at the end to gather per-env returns (through
the library's own smj_allgather_returns) and for the barrier / max-over-ranks timing.
"""
import argparse
import json
import multiprocessing as mp
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_ENV_STEP = 672.0  # physics-only algorithmic state traffic
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FLOPS_PER_ENV_STEP_FILE = os.path.join(ROOT, "profiles", "flops_per_env_step.json")
HOME_CTRL = [0, 0, 0.6, 0.1, 0, 0, 0, 0, 0, 0]


# ---------------------------------------------------------------------------------------------- CPU baseline
def _cpu_worker(args):
    """One oracle process: `seconds` of the bench workload (or its C2 / C3 variants) on one core; returns (steps, seconds)."""
    blob_path, seed, hold, seconds, solver, sensors = args
    from oracle.oracle import Oracle

    with open(blob_path, "rb") as f:
        blob = f.read()
    o = Oracle(blob)
    o.set_option("solver", 2 if solver == "newton" else 0)
    nu = o.dim("nu")
    o.arr("ctrl")[:nu] = HOME_CTRL[:nu]
    o.step(500)
    import stretch_mujoco_amd.model_blob as mb

    cr = np.asarray(mb.loads(blob)["actuator_ctrlrange"])
    rng = np.random.default_rng(seed)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        o.arr("ctrl")[:nu] = cr[:, 0] + (cr[:, 1] - cr[:, 0]) * rng.random(nu)
        if sensors:   # C2: what mj_step does with the 360 rangefinders + IMU enabled: sensors evaluated every step
            for _ in range(hold):
                o.step(1)
                o.sensors(True)
        else:
            o.step(hold)
        n += hold
    return n, time.perf_counter() - t0


def usable_cores() -> int:
    """Cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (a GPU box shows 256 CPUs to
    os.cpu_count() while its container is limited to a fraction of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(hold: int, seconds: float, solver: str):
    """BASELINE.md section 3, C1-C4 on the fp64 CPU restatement ('port'; MuJoCo itself is not installable here).
    `value` is C4, the whole-host aggregate (one independent oracle process per core)."""
    models = os.path.join(ROOT, "stretch_mujoco_amd", "models")
    empty = os.path.join(models, "stretch_empty.smjb")
    scene = os.path.join(models, "stretch_scene.smjb")          # scene.xml equivalent: table + 2 free objects
    if not os.path.exists(scene):
        scene = os.path.join(models, "stretch_kitchen_standin.smjb")
    kitchen = os.path.join(models, "stretch_kitchen_robocasa.smjb")   # config 4's scene (the generated kitchen at Robocasa scale, 82 dofs)
    ncpu = usable_cores()
    from oracle import oracle as _o

    _o.lib()                      # make sure the oracle library is built before the workers start
    ctx = mp.get_context("spawn")   # the parent holds a HIP context: no fork
    single = {}
    jobs = [(empty, 1, hold, seconds / 2, solver, False), (empty, 2, hold, seconds / 2, solver, True), (scene, 3, hold, seconds / 2, solver, False)]
    keys = ["C1_empty_sensors_off", "C2_empty_lidar_imu_every_step", "C3_" + os.path.basename(scene)[:-5]]
    if os.path.exists(kitchen):
        jobs.append((kitchen, 4, hold, seconds / 2, solver, False)); keys.append("C3_kitchen_robocasa")
    with ctx.Pool(min(len(jobs), ncpu)) as pool:   # single-thread rates, measured side by side on otherwise idle cores
        r = pool.map_async(_cpu_worker, jobs).get(timeout=4 * seconds + 120)
    for key, (n, dt) in zip(keys, r):
        single[key] = {"value": n / dt, "unit": "env-steps/s", "steps": n}
    with ctx.Pool(ncpu) as pool:
        r = pool.map_async(_cpu_worker, [(empty, 100 + i, hold, seconds, solver, False) for i in range(ncpu)]).get(timeout=4 * seconds + 120)
    total = sum(n / dt for n, dt in r)
    kit = None
    if os.path.exists(kitchen):   # C4 on config 4's scene: the comparator of the kitchen figures (BASELINE.md section 3 asks for one per reported config)
        with ctx.Pool(ncpu) as pool:
            rk = pool.map_async(_cpu_worker, [(kitchen, 200 + i, hold, seconds / 2, solver, False) for i in range(ncpu)]).get(timeout=4 * seconds + 120)
        kit = {"value": sum(n / dt for n, dt in rk), "unit": "env-steps/s", "cores": ncpu, "kind": "port",
               "sample": f"C4 on the kitchen at Robocasa scale: {ncpu} independent oracle processes x {seconds / 2:.0f} s (1 env each, 82 dofs, random ctrl every {hold} steps, {solver}); "
                         f"{sum(n for n, _ in rk)} steps in total; single thread: single_thread.C3_kitchen_robocasa"}
    return dict(value=total, unit="env-steps/s", cores=ncpu, kind="port", kitchen_robocasa=kit,
                sample=f"C4 whole-host aggregate: {ncpu} independent oracle processes, one per usable core ({os.cpu_count()} visible, cgroup quota / affinity allow {ncpu}), {seconds:.0f} s each of the bench "
                       f"workload (1 env, empty scene, random ctrl every {hold} steps, {solver}); {sum(n for n, _ in r)} steps in total. "
                       f"Stand-in fp64 CPU restatement, NOT MuJoCo.",
                single_thread=single, host_cpus=os.cpu_count(), usable_cpus=ncpu,
                reference_as_shipped="<= 500 steps/s by construction (realtime sleep, mujoco_server.py:381-384); not measured")


# ---------------------------------------------------------------------------------------------- other configs
def other_configs(B, dev, hold, solver):
    """Short single-GPU runs of the other BASELINE.json configs (reported beside the headline line, never as `value`)."""
    import torch

    from stretch_mujoco_amd import StretchBatchSimulator, StretchSensors
    from stretch_mujoco_amd.enums import StretchCameras

    res = {}

    def rollout(sim, n, chunk, settle=500, preroll=4, events=None):
        """`events` (a list): HIP events around every timed sim.step (torch's current stream = the launch stream) are appended."""
        gen = torch.Generator(device=dev).manual_seed(99)
        lo = torch.tensor(sim.model["actuator_ctrlrange"][:, 0], dtype=torch.float32, device=dev).unsqueeze(1)
        hi = torch.tensor(sim.model["actuator_ctrlrange"][:, 1], dtype=torch.float32, device=dev).unsqueeze(1)
        sim.ctrl[:] = torch.tensor(sim.model["key_ctrl"][0, : sim.nu], dtype=torch.float32, device=dev).unsqueeze(1)
        sim.step(settle)
        for _ in range(preroll):   # into the steady state of the random-action rollout
            sim.ctrl.copy_(lo + (hi - lo) * torch.rand(sim.nu, sim.num_envs, generator=gen, device=dev))
            sim.step(hold)
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        d = 0
        while d < n:
            if d % hold == 0:
                sim.ctrl.copy_(lo + (hi - lo) * torch.rand(sim.nu, sim.num_envs, generator=gen, device=dev))
            if events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            sim.step(chunk)
            if events is not None:
                e1.record()
                events.append((e0, e1, chunk))
            d += chunk
        torch.cuda.synchronize(dev)
        return sim.num_envs * d / (time.perf_counter() - t)

    flops_table = {}
    if os.path.exists(FLOPS_PER_ENV_STEP_FILE):
        with open(FLOPS_PER_ENV_STEP_FILE) as fh:
            flops_table = json.load(fh).get("scenes", {})

    def roofline_of(sim, events, scene_key, kernel):
        """The contract's roofline block for a scene other than the headline's, from THIS run's HIP events: algorithmic bytes = the state
        a step has to read (qpos, qvel, ctrl, warm start) and write (qpos, qvel, warm start), fp32, x envs x steps of the timed launches
        / their summed duration."""
        nq, nv, nu = int(sim.qpos.shape[0]), int(sim.qvel.shape[0]), int(sim.nu)
        bytes_step = 4.0 * ((nq + nv + nu + nv) + (nq + nv + nv))
        ms = sum(a.elapsed_time(b) for a, b, _ in events)
        steps = sum(k for _, _, k in events)
        ach = sim.num_envs * steps * bytes_step / (ms / 1e3) / 1e9
        traffic, tsrc = None, None
        tp = os.path.join(ROOT, "profiles", "pmc_traffic_scenes.json")
        if os.path.exists(tp) and sim.num_envs == 4096 and hold == 50:   # HBM bytes per 50-step launch, rocprofv3 --pmc passes of the same workload on an earlier run
            with open(tp) as fh:
                pm = json.load(fh)
            if scene_key in pm.get("scenes", {}):
                traffic, tsrc = pm["scenes"][scene_key]["hbm_bytes_per_launch"], pm.get("source", "") + " -- NOT measured inside this run"
        r = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": tsrc,
             "bytes_per_env_step": bytes_step, "kernel": kernel, "timed_launches": len(events), "timed_kernel_ms": ms,
             "kernel_ms_per_step": ms / max(1, steps)}
        fl = flops_table.get(scene_key, {}).get("flops_per_env_step")
        if fl:
            r["fp32"] = {"flops_per_env_step": fl, "achieved_tflops": fl * sim.num_envs * steps / (ms / 1e3) / 1e12, "peak_tflops": 157.3,
                         "frac": fl * sim.num_envs * steps / (ms / 1e3) / 1e12 / 157.3}
        return r

    def flags_of(sim):
        f = sim.info[3]
        # bits: 1 row overflow, 2 contact overflow, 4 bad-state reset, 8 pipeline time-out (the env ran fewer steps than asked)
        return {"overflow_flags": int(torch.bitwise_or(torch.bitwise_or((f & 1).max(), (f & 2).max()), torch.bitwise_or((f & 4).max(), (f & 8).max())).item()),
                "envs_flagged": float((f != 0).float().mean().item()),
                "steps_min_max": [int(sim.nstep.min().item()), int(sim.nstep.max().item())]}

    # config 2: 1024 envs, physics only -- and the per-rank shares of BASELINE.json's metric (4096 envs IN TOTAL at 2 / 4 / 8 GPUs =
    # 2048 / 1024 / 512 envs per rank), measured on this one GPU: what a strong-scaling run can reach per rank.  At <= 1024 envs
    # every env has a wave slot of its own (256 CUs x 4), so a launch lasts as long as its slowest env.
    share = {}
    for nb in (2048, 1024, 512):
        sim = StretchBatchSimulator(num_envs=nb, device=str(dev), solver=solver)
        sim.start(home=False)
        share[str(nb)] = {"value": rollout(sim, 200, hold), "unit": "env-steps/s", **flags_of(sim)}
        sim.stop()
    res["config2_1024_envs_physics"] = {"value": share["1024"]["value"], "unit": "env-steps/s"}
    res["strong_scaling_rank_share"] = {"envs_per_rank": share, "note": "empty scene, physics only, one GPU: the per-rank batch of 4096 envs in total at 2 / 4 / 8 GPUs"}
    # config 3: + joint readout, IMU, 2-D lidar; at 15 Hz sim-time (every 33 steps) and every step
    sim = StretchBatchSimulator(num_envs=B, device=str(dev), solver=solver, sensors_to_use=StretchSensors.all())
    sim.start(home=False)
    res["lidar_imu_every_33_steps"] = {"value": rollout(sim, 330, 33), "unit": "env-steps/s"}
    res["lidar_imu_every_step"] = {"value": rollout(sim, 50, 1), "unit": "env-steps/s"}
    sim.stop()
    # config 4 stand-in: static kitchen fixtures around the robot + free objects, physics only
    # `_sat`: the same scene on the satellite builds of the step kernel (free objects / fixture parts one lane each, csrc/smj_sat.h).
    # `stretch_kitchen_robocasa`: the generated kitchen at Robocasa scale -- 44 fixture bodies, 307 collision geoms (36 convex mesh
    # pieces) behind the static-geometry broadphase, 8 articulated doors / drawers / knobs, 8 free objects: 82 dofs, 16 satellites.
    for scene in ("stretch_kitchen_standin", "stretch_kitchen4", "stretch_kitchen4_sat", "stretch_scene", "stretch_scene_sat", "stretch_kitchen_robocasa"):
        if not os.path.exists(os.path.join(ROOT, "stretch_mujoco_amd", "models", scene + ".smjb")):
            continue
        sim = StretchBatchSimulator(num_envs=B, device=str(dev), solver=solver, scene=scene)
        sim.start(home=False)
        evs = []
        res[scene + "_physics"] = {"value": rollout(sim, 500, hold, events=evs), "unit": "env-steps/s", **flags_of(sim)}   # 10 launches: one with a hand-over to the larger variant costs +30 %
        res[scene + "_physics"]["roofline"] = roofline_of(sim, evs, scene[:-4] if scene.endswith("_sat") else scene,
                                                          "smj_step_kernel_sat2 (two wavefronts per env; + _sat32 workers)" if "robocasa" in scene or scene.endswith("_sat") else "smj_step_kernel (variant by model size)")
        if scene == "stretch_kitchen_robocasa":
            res[scene + "_physics"].update(dofs=sim.nv, kernel_variant="sat2 (16 satellites, 208 rows, 2 envs per CU, two wavefronts per env: the second takes the moving-moving pairs and the satellites' lane-serial stages) -> sat32 (320 rows) for steps beyond it; under PGS the two-wavefront build satp (satellite islands swept beside the dense system)",
                                           note="overflow_flags bit 2 = more than 64 contacts in one env (a lane count); rows / dense rows / coupled satellites hand over and are not flagged")
            if solver == "newton":   # the same rollout on the one-wavefront kernel (smj_kernels_sat.hip; bit-identical states, tests/test_satellites.py): what the second wavefront buys
                sim.stop()
                sim = StretchBatchSimulator(num_envs=B, device=str(dev), solver=solver, scene=scene)
                sim.start(home=False)
                sim.set_option("newton_two_waves", 0)
                res[scene + "_physics"]["one_wavefront_per_env"] = {"value": rollout(sim, 500, hold), "unit": "env-steps/s", "kernel": "smj_step_kernel_sat (option newton_two_waves = 0)"}
        sim.stop()
    # north_star's target sentence (>= 1 M env-steps/s on 4096 kitchen envs at 8 GPUs): the per-rank share of 4096 kitchen envs in
    # total, on this one GPU -- 512 envs per rank at 8 GPUs, 1024 at 4
    kshare = {}
    for scene in ("stretch_kitchen_robocasa", "stretch_kitchen4_sat"):
        if not os.path.exists(os.path.join(ROOT, "stretch_mujoco_amd", "models", scene + ".smjb")):
            continue
        kshare[scene] = {}
        for nb in (1024, 512):
            sim = StretchBatchSimulator(num_envs=nb, device=str(dev), solver=solver, scene=scene)
            sim.start(home=False)
            kshare[scene][str(nb)] = {"value": rollout(sim, 200, hold), "unit": "env-steps/s", **flags_of(sim)}
            sim.stop()
            if solver == "newton" and scene == "stretch_kitchen_robocasa":   # north_star names PGS for this configuration: the same share under PGS
                sim = StretchBatchSimulator(num_envs=nb, device=str(dev), solver="pgs", scene=scene)
                sim.start(home=False)
                kshare[scene][str(nb) + "_pgs"] = {"value": rollout(sim, 200, hold), "unit": "env-steps/s", **flags_of(sim)}
                sim.stop()
    res["kitchen_rank_share"] = {"envs_per_rank": kshare, "note": "one GPU, physics only, random actions: x8 (512 envs) / x4 (1024) = what 4096 kitchen envs in total can reach on a node.  "
                                 "Under PGS a launch lasts as long as its slowest env -- one at the cap of 100 sweeps takes ~4 ms per step on the 16-satellite build, one handed to the 32-satellite build "
                                 "(more than 96 rows on the robot's island) ~15 ms -- whatever the batch: the PGS share scales with the batch size (DESIGN.md section 7)"}
    # config 4 as north_star words it ("contact-rich PGS solve"): the same scenes under PGS.  The sweeps are serial over the rows
    # (100 sweeps x ~100 rows at one wavefront per env), so this is the slowest path of the library; Newton is the model's own solver.
    if solver != "pgs":
        # `_sat` / the Robocasa-scale kitchen: PGS with constraint islands (csrc/smj_sat_pgs.h) -- the robot's rows as one dense system,
        # every free object / fixture part that touches only the static world swept by its own lane, all in the same iteration
        for scene, n in (("stretch_kitchen_standin", 200), ("stretch_kitchen4", 100), ("stretch_kitchen4_sat", 200), ("stretch_scene", 100), ("stretch_scene_sat", 200),
                         ("stretch_kitchen_robocasa", 200)):
            if not os.path.exists(os.path.join(ROOT, "stretch_mujoco_amd", "models", scene + ".smjb")):
                continue
            sim = StretchBatchSimulator(num_envs=B, device=str(dev), solver="pgs", scene=scene)
            sim.start(home=False)
            evs = []
            res[scene + "_physics_pgs"] = {"value": rollout(sim, n, hold, events=evs), "unit": "env-steps/s", **flags_of(sim)}   # same settle / pre-roll as the Newton runs: the steady state of the random-action workload
            res[scene + "_physics_pgs"]["sweeps_last_step"] = {"mean": float(sim.info[2].float().mean().item()), "at_cap_of_100": float((sim.info[2] >= 100).float().mean().item())}
            if scene == "stretch_kitchen_robocasa":
                res[scene + "_physics_pgs"]["roofline"] = roofline_of(sim, evs, scene + ":pgs", "smj_step_kernel_satp (two wavefronts per env)")
                res[scene + "_physics_pgs"]["solver_iterations_mean"] = float(sim.info[2].float().mean().item())
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
                                 "hbm_frac": 4 * rays / dt / 1e9 / HBM_PEAK_GBS,
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
    # config 5's per-rank shape on the kitchen at Robocasa scale (satellite builds + both depth cameras every 17 steps)
    if os.path.exists(os.path.join(ROOT, "stretch_mujoco_amd", "models", "stretch_kitchen_robocasa.smjb")):
        sim = StretchBatchSimulator(num_envs=B, device=str(dev), solver=solver, scene="stretch_kitchen_robocasa", cameras_to_use=StretchCameras.depth())
        sim.start(home=False)
        rollout(sim, hold, hold)
        sim.pull_camera_data()
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        n = 0
        for _ in range(6):
            sim.step(17)
            sim.pull_camera_data()
            n += 17
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t
        t = time.perf_counter()
        for _ in range(3):
            sim.pull_camera_data()
        torch.cuda.synchronize(dev)
        res["kitchen_robocasa_with_depth_30hz"] = {"value": B * n / dt, "unit": "env-steps/s", "ms_per_render": (time.perf_counter() - t) / 3 * 1e3,
                                                   "note": "config 5's per-rank share (4096 of 32768 envs): the generated Robocasa-scale kitchen, 17 physics steps + one render of both depth cameras, repeated"}
        sim.stop()
    return res


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def shard_sizes(envs, rank, world, scaling):
    """(envs of this rank, envs in total): strong = `envs` in total in contiguous shards (parallel.shard_range), weak = `envs` per rank."""
    from stretch_mujoco_amd import parallel

    if scaling == "strong":
        lo_e, hi_e = parallel.shard_range(envs, rank, world)
        return hi_e - lo_e, envs
    return envs, envs * world


def measure_one(args, rank, world, hooks, scaling, gather=None):
    """One measurement of the contract on this rank: settle, pre-roll, W warm-up steps, then EXACTLY K timed steps between two
    barrier + synchronize brackets, max over ranks.  Returns everything the JSON line is made of.  Device and torch.distributed come in
    through `hooks` so that the control flow (shards, brackets, max over ranks, gather, ranks_seen) runs on the CPU under gloo as well."""
    import torch

    B, B_total = shard_sizes(args.envs, rank, world, scaling)
    sim = hooks.make_sim(B)
    dev = hooks.device
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    lo = torch.tensor(sim.model["actuator_ctrlrange"][:, 0], dtype=torch.float32, device=dev).unsqueeze(1)
    hi = torch.tensor(sim.model["actuator_ctrlrange"][:, 1], dtype=torch.float32, device=dev).unsqueeze(1)

    def random_action():
        sim.ctrl.copy_(lo + (hi - lo) * torch.rand(sim.nu, B, generator=gen, device=dev))

    hold = max(1, args.hold)
    all_events = []   # every smj_step_kernel launch of this process (what `rocprofv3 --stats` averages over)
    phase = [0]       # steps since the last action change

    def timed_step(k, tag):
        e0, e1 = hooks.event(), hooks.event()
        e0.record()
        sim.step(k)
        e1.record()
        all_events.append((e0, e1, k, tag))
        return e0, e1

    def rollout(nsteps, tag, on_launch=None):
        """nsteps of the random-action schedule, continuing it: a new action whenever `hold` steps have passed."""
        done = 0
        while done < nsteps:
            if phase[0] == 0:
                random_action()
            k = min(hold - phase[0], nsteps - done)
            ev = timed_step(k, tag)
            if on_launch:
                on_launch(ev, k)
            phase[0] = (phase[0] + k) % hold
            done += k

    # settle at the home keyframe, then into the steady state of the random-action rollout
    sim.ctrl[:] = torch.tensor(sim.model["key_ctrl"][0, : sim.nu], dtype=torch.float32, device=dev).unsqueeze(1)
    for _ in range(500 // hold):
        timed_step(hold, "settle")
    rollout(max(200, 4 * hold), "preroll")
    rollout(args.warmup, "warmup")
    returns = torch.zeros(B, device=dev)

    def barrier():
        hooks.sync()
        hooks.barrier()
        hooks.sync()

    # The timed region: EXACTLY K steps between two barrier + synchronize brackets, max over ranks.  A K below one action interval is a
    # single launch of a few milliseconds -- one sample, which moved the figure by +-4 % from round to round -- so such a region is
    # measured `reps` times (each repetition its own bracket of exactly K steps at the SAME phase of the action schedule: the untimed
    # remainder of the action interval runs between two brackets, so the repetitions differ by their random actions only) and the MEDIAN
    # repetition is reported -- its wall time AND its kernel time (one statistic for ms_per_step and kernel_ms_per_step); every
    # repetition is listed in `timed_region_ms`.
    reps = 3 if args.steps < hold else 1
    samples, rep_events = [], []
    for rep in range(reps):
        if rep:
            rollout((hold - args.steps) % hold, "between")
        evs = []

        def on_launch(ev, k, evs=evs):
            evs.append((ev[0], ev[1], k))
            returns.add_(sim.base_pose[0])   # synthetic per-env return: accumulated forward displacement

        barrier()
        t0 = time.perf_counter()
        rollout(args.steps, "timed", on_launch)
        barrier()
        samples.append(hooks.max_over_ranks(time.perf_counter() - t0))
        rep_events.append(evs)
    pick = sorted(range(reps), key=lambda i: samples[i])[reps // 2]
    dt, events = samples[pick], rep_events[pick]
    all_returns, gather_path = (gather or hooks.gather)(sim, returns)
    f = sim.info[3]
    flags = int(((f & 1).max() | (f & 2).max() | (f & 4).max() | (f & 8).max()).item())   # union of the sticky overflow / bad-state / time-out bits
    flagged = float((f != 0).float().mean().item())
    # bit 8 = a pipelined chunk gave up waiting: that env ran fewer steps than asked and the env-step count below would be wrong
    if flags & 8 or int(sim.nstep.min().item()) != int(sim.nstep.max().item()):
        raise SystemExit(f"bench.py: envs ran different step counts ({int(sim.nstep.min())}..{int(sim.nstep.max())}, flags {flags}): pipeline time-out, the measurement is void")
    kern_ms = sum(a.elapsed_time(b) for a, b, _ in events)
    kern_steps = sum(k for _, _, k in events)
    return dict(sim=sim, B=B, B_total=B_total, hold=hold, dt=dt, samples=samples, reps=reps, value=float(B_total) * args.steps / dt, events=events, all_events=all_events,
                kern_ms=kern_ms, kern_steps=kern_steps, flags=flags, flagged=flagged, all_returns=all_returns, gather_path=gather_path, returns=returns,
                random_action=random_action, scaling=scaling)


def measure_ranks(args, rank, world, hooks):
    """The headline measurement (--scaling, default strong: BASELINE.json's 4096 envs IN TOTAL) and, at N > 1, the other scaling mode
    beside it (key `other_scaling` of the result: weak = --envs per GPU), measured back to back by the same ranks."""
    m = measure_one(args, rank, world, hooks, args.scaling)
    m["other_scaling"] = None
    if world > 1:
        other = "weak" if args.scaling == "strong" else "strong"
        from stretch_mujoco_amd import parallel

        # (the second measurement gathers through torch.distributed: the library's own RCCL communicator is brought up once per process, by the headline)
        o = measure_one(args, rank, world, hooks, other, gather=lambda sim, r: (parallel.gather_returns(r), "torch.distributed all_gather_into_tensor"))
        o["sim"].stop()
        m["other_scaling"] = {"scaling": other, "value": o["value"], "unit": "env-steps/s", "ms_per_step": o["dt"] * 1e3 / args.steps, "envs_total": o["B_total"],
                              "envs_per_gpu": o["B"], "ranks_seen": int(o["all_returns"].numel()) // max(1, int(o["returns"].numel())),
                              "timed_region_ms": [round(x * 1e3, 3) for x in o["samples"]], "overflow_flags": o["flags"]}
    return m


def line_head(args, world, m):
    """The contract's keys of the JSON line (everything but the roofline / baselines / other configs), from a measure_ranks result."""
    B, B_total, hold = m["B"], m["B_total"], m["hold"]
    all_returns, returns = m["all_returns"], m["returns"]
    return {
        "metric": "env-steps/sec (whole node), 4096 parallel Stretch envs" + (" per MI355X" if args.scaling == "weak" else " in total"),
        "value": m["value"], "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": m["dt"] * 1e3 / args.steps, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{B_total} parallel Stretch envs ({B} per GPU), stretch.xml + ground plane (empty scene), "
                               f"physics-only, random ctrl in ctrlrange every {hold} steps (steady state: 500 settle + 200 "
                               f"untimed random-action steps precede warm-up), {args.solver} solver "
                               f"(iterations<=100, tol 1e-8), elliptic cones impratio 20, implicitfast, dt=0.002",
                   "solver": args.solver, "envs_total": B_total,
                   "envs_per_gpu": B, "steps_per_action": hold, "parallelism": f"env-sharded x{world}",
                   "returns_gathered": int(all_returns.numel()), "ranks_seen": int(all_returns.numel()) // max(1, int(returns.numel())),
                   "returns_gather_path": m["gather_path"], "timed_region_ms": [round(x * 1e3, 3) for x in m["samples"]],
                   "timed_region_reported": "median (wall and kernel time of the same repetition)" if m["reps"] > 1 else "the one bracket",
                   "overflow_flags": m["flags"], "envs_over_capacity_since_reset": m["flagged"]},
        **({m["other_scaling"]["scaling"]: m["other_scaling"]} if m.get("other_scaling") else {}),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="strong (default): --envs in total, split over the ranks -- BASELINE.json's metric is 4096 envs at 1 / 2 / 4 / 8 GPUs; at N > 1 the same "
                         "line also carries the weak figure (--envs per GPU) under `weak`.  weak: --envs per GPU is `value`")
    ap.add_argument("--envs", "--envs-per-gpu", dest="envs", type=int, default=4096)
    ap.add_argument("--hold", type=int, default=50, help="physics steps per random action (and per launch)")
    ap.add_argument("--solver", choices=["newton", "pgs"], default="newton",
                    help="newton = the reference model's own solver (stretch.xml names none -> MuJoCo default); "
                         "pgs = the solver named by BASELINE.json north_star")
    ap.add_argument("--no-second-solver", action="store_true", help="skip the short run of the other solver")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the short runs of the other BASELINE.json configs")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, RCCL rendezvous on 127.0.0.1)
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(torch.distributed.run --nproc-per-node {args.gpus}) or drop WORLD_SIZE")
    if torch.cuda.device_count() < max(1, local_rank + 1):
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank}, this host exposes {torch.cuda.device_count()}")
    dist_on = world > 1
    if dist_on:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from stretch_mujoco_amd import StretchBatchSimulator
    from stretch_mujoco_amd import parallel

    class Hooks:   # what the N-rank control flow needs from the device and from torch.distributed (tests/test_bench_ranks.py swaps in CPU stand-ins)
        device = dev

        @staticmethod
        def make_sim(B):
            sim = StretchBatchSimulator(num_envs=B, device=str(dev), solver=args.solver)
            sim.start(home=False)
            return sim

        @staticmethod
        def sync():
            torch.cuda.synchronize(dev)

        @staticmethod
        def event():
            return torch.cuda.Event(enable_timing=True)

        @staticmethod
        def barrier():
            if dist_on:
                dist.barrier()

        @staticmethod
        def max_over_ranks(x):
            if not dist_on:
                return x
            t = torch.tensor([x], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())

        @staticmethod
        def gather(sim, returns):
            try:   # RCCL all-gather of per-env returns inside libsmj.so (the only collective of the path); after the timed region
                return parallel.gather_returns_native(sim, returns)
            except Exception as e:   # never lose the measured line to the gather: fall back to torch.distributed's and say so
                return parallel.gather_returns(returns), f"torch.distributed all_gather_into_tensor (the library's RCCL path failed: {type(e).__name__}: {e})"

    m = measure_ranks(args, rank, world, Hooks)
    sim, B, B_total, hold, dt, samples, reps, value = m["sim"], m["B"], m["B_total"], m["hold"], m["dt"], m["samples"], m["reps"], m["value"]
    events, all_events, kern_ms, kern_steps, flags, flagged = m["events"], m["all_events"], m["kern_ms"], m["kern_steps"], m["flags"], m["flagged"]
    all_returns, gather_path, returns = m["all_returns"], m["gather_path"], m["returns"]
    if rank == 0:
        # roofline of the dominant kernel: algorithmic bytes of the timed launches / their summed duration (HIP events on the
        # launch stream), i.e. 672 B x envs x steps-per-launch / average launch duration
        achieved = B * kern_steps * BYTES_PER_ENV_STEP / (kern_ms / 1e3) / 1e9
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc_path):   # HBM bytes per 50-step launch measured with rocprofv3 --pmc on this same command
            with open(pmc_path) as fh:
                pm = json.load(fh)
            if pm.get("envs_per_gpu") == B and pm.get("steps_per_launch") == hold and pm.get("solver") == args.solver:
                traffic = pm["hbm_bytes_per_launch"]
        flops = None
        if os.path.exists(FLOPS_PER_ENV_STEP_FILE):
            with open(FLOPS_PER_ENV_STEP_FILE) as fh:
                flops = json.load(fh)
        launches = [{"tag": tag, "steps": k, "ms": round(a.elapsed_time(b), 3)} for a, b, k, tag in all_events]
        out = {
            **line_head(args, world, m),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": "profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on an earlier run "
                                           "(read side x2 per MI355X_MICROARCH.md), NOT measured inside this run" if traffic is not None else None,
                         "kernel": "smj_step_kernel", "timed_launches": len(events), "timed_kernel_ms": kern_ms,
                         "kernel_ms_per_step": kern_ms / max(1, kern_steps),
                         "us_per_env_step_latency": kern_ms * 1e3 / max(1, kern_steps),
                         "launches": launches,
                         "note": "achieved = 672 B/env-step x envs x steps of the timed launches / their summed duration (HIP events); "
                                 "`traffic` = HBM bytes per 50-step launch from the rocprofv3 --pmc passes of this command (profiles/); "
                                 "the path is bound by the instruction issue of one wavefront per SIMD, HBM does not bind (DESIGN.md section 4)"},
        }
        if flops:
            fl = flops.get("flops_per_env_step")
            out["roofline"]["fp32"] = {"flops_per_env_step": fl, "achieved_tflops": fl * B * kern_steps / (kern_ms / 1e3) / 1e12,
                                       "peak_tflops": 157.3, "source": flops.get("source")}
        if not args.no_second_solver and world == 1:
            other = "pgs" if args.solver == "newton" else "newton"
            sim.set_option("solver", {"pgs": 0, "newton": 2}[other])
            n2 = 300   # six launches whatever --steps is: two made the figure swing by 5 %
            random_action = m["random_action"]
            random_action(); sim.step(hold)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            d2 = 0
            while d2 < n2:
                k = min(hold, n2 - d2)
                random_action(); sim.step(k); d2 += k
            torch.cuda.synchronize(dev)
            f2 = sim.info[3]
            out["other_solver"] = {"solver": other, "value": B * n2 / (time.perf_counter() - t1), "unit": "env-steps/s",
                                   "n_gpus": 1, "steps": n2, "envs_flagged": float((f2 != 0).float().mean().item()),
                                   "flags_union": int(((f2 & 1).max() | (f2 & 2).max() | (f2 & 4).max() | (f2 & 8).max()).item()),
                                   "note": "rank 0 only, same workload, measured after the timed region; envs_flagged counts envs that ran out of "
                                           "constraint rows / contacts since the reset (flags are sticky: the Newton phase before contributes none)"}
        import __graft_entry__ as _ge
        out["parity_oracle"] = _ge.mujoco_status()
        oc = cb = None
        if not args.no_extra and world == 1:
            sim.stop()
            oc = other_configs(B, dev, hold, args.solver)
        if not args.no_cpu_baseline and world == 1:   # the contract: rank 0 at N=1 only
            cb = cpu_baseline(hold, args.cpu_seconds, args.solver)
        if oc and oc.get("stretch_kitchen_robocasa_physics"):
            # config 4 -- the configuration north_star's target sentence is quoted on -- as a top-level key ahead of the long blocks, so that a
            # truncating reader keeps it: 4096 kitchens on this one GPU under the model's own solver and under PGS, their rooflines and flags,
            # the per-rank shares of 4096 kitchens in total, and the CPU comparator on the same scene
            kn, kp = oc["stretch_kitchen_robocasa_physics"], oc.get("stretch_kitchen_robocasa_physics_pgs")
            out["config4"] = {"workload": f"{B} parallel Stretch envs in the generated kitchen at Robocasa scale (44 fixture bodies, 307 collision geoms, 8 articulated parts, 8 free objects: "
                                          f"82 dofs), physics only, random ctrl every {hold} steps, 1 MI355X; Robocasa itself is unavailable here (stand-in, DESIGN.md section 1)",
                              "newton": {k: kn.get(k) for k in ("value", "unit", "overflow_flags", "envs_flagged", "roofline", "kernel_variant")},
                              "pgs": {k: kp.get(k) for k in ("value", "unit", "overflow_flags", "envs_flagged", "sweeps_last_step", "roofline")} if kp else None,
                              "rank_share_of_4096_in_total": oc.get("kitchen_rank_share", {}).get("envs_per_rank", {}).get("stretch_kitchen_robocasa"),
                              "cpu_baseline": {"aggregate": cb.get("kitchen_robocasa"), "single_thread": cb.get("single_thread", {}).get("C3_kitchen_robocasa")} if cb else None,
                              "note": "delivered under Newton, the model's own solver (stretch.xml names none: MuJoCo's default); PGS, which north_star names, is the slower option (DESIGN.md section 7)"}
        if oc:
            out["other_configs"] = oc
        if cb:
            out["cpu_baseline"] = cb
        print(json.dumps(out))
    sim.stop()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
