"""Soak run: 4096 envs, random actions every 50 steps for N steps; reports bad-state / overflow flags, finiteness, drift of
invariants (unit quaternion, equality constraints), per-launch time spread."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from stretch_mujoco_amd import StretchBatchSimulator  # noqa: E402


def main(B=4096, steps=20000, scene="stretch_empty", solver="newton", options=None, capture=None, capture_z=0.35, capture_max=12, seed=99):
    """capture: path of an .npz that receives, for the first `capture_max` envs whose base rises above capture_z, the env's state and
    ctrl at the START of the launch in which it happens and of the launch before (so the CPU oracle / the lane emulator can run the same
    100 steps: is the robot thrown by the physics or by the kernel?)."""
    sim = StretchBatchSimulator(num_envs=B, device="cuda:0", solver=solver, scene=scene)
    sim.start(home=True)
    for k_, v_ in (options or {}).items():
        sim.set_option(k_, v_)
    zmax_ever, zmax_at, zmax_who = -1.0, -1, ""
    dev = sim.device
    g = torch.Generator(device=dev).manual_seed(seed)
    lo = torch.tensor(sim.model["actuator_ctrlrange"][:, 0], dtype=torch.float32, device=dev).unsqueeze(1)
    hi = torch.tensor(sim.model["actuator_ctrlrange"][:, 1], dtype=torch.float32, device=dev).unsqueeze(1)
    t0 = time.perf_counter()
    worst_q = 0.0
    per_launch = []
    ever = torch.zeros(B, dtype=torch.int32, device=dev)
    capped = 0   # launches x envs whose last step's solver ended at the iteration cap
    itmax = int(sim.model.get('opt_iterations', [100])[0]) if hasattr(sim.model, 'get') else 100
    caps, seen, prev = [], set(), None
    for k in range(steps // 50):
        sim.ctrl.copy_(lo + (hi - lo) * torch.rand(sim.nu, B, generator=g, device=dev))
        sim.info[3].zero_()          # flags are sticky: clear them to count per launch
        if capture:
            cur = (sim.qpos.clone(), sim.qvel.clone(), sim.qacc_warmstart.clone(), sim.ctrl.clone())
        sim.step(50)
        if capture:
            high = torch.nonzero(sim.qpos[2] > capture_z).flatten().tolist()
            for e_ in high:
                if e_ not in seen and len(caps) < capture_max and prev is not None:
                    seen.add(e_)
                    caps.append(dict(env=e_, launch=k, flags=int(sim.info[3, e_]), z_after=float(sim.qpos[2, e_]),
                                     **{f"{nm}{j}": t[:, e_].cpu().numpy() for j, st in enumerate((prev, cur)) for nm, t in zip(("qpos", "qvel", "warm", "ctrl"), st)}))
                    print(f"  captured env {e_} at launch {k}: base z {float(sim.qpos[2, e_]):.3f}, flags {hex(int(sim.info[3, e_]))}, contacts {int(sim.info[1, e_])}", flush=True)
            prev = cur
        per_launch.append(float(((sim.info[3] & 1) != 0).float().mean()))
        ever |= sim.info[3]
        capped += int((sim.info[2] >= itmax).sum())
        zk = float(sim.qpos[2].max())
        if zk > 0.6:   # a 23 kg robot does not get there by itself: report the env while it happens
            e_ = int(sim.qpos[2].argmax())
            print(f"  launch {k}: env {e_} base z {zk:.3f}, |qvel| max {float(sim.qvel[:, e_].abs().max()):.1f}, flags of this launch {hex(int(sim.info[3, e_]))} (ever {hex(int(ever[e_]))}), "
                  f"rows / contacts / iterations of its last step {int(sim.info[0, e_])} / {int(sim.info[1, e_])} / {int(sim.info[2, e_])}", flush=True)
        if zk > zmax_ever:
            zmax_ever, zmax_at = zk, k
            e_ = int(sim.qpos[2].argmax())
            zmax_who = f"env {e_}, its flag bits so far {hex(int(ever[e_]))}, rows / contacts of its last step {int(sim.info[0, e_])} / {int(sim.info[1, e_])}"
        if k % 40 == 39:
            q = sim.qpos
            assert torch.isfinite(q).all() and torch.isfinite(sim.qvel).all(), k
            worst_q = max(worst_q, float((q[3:7].norm(dim=0) - 1).abs().max()))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if capture:
        np.savez(capture, n=len(caps), **{f"c{i}_{k_}": np.asarray(v_) for i, c in enumerate(caps) for k_, v_ in c.items()})
    fl = ever
    z = sim.qpos[2]
    up = 1 - 2 * (sim.qpos[4] ** 2 + sim.qpos[5] ** 2)
    print(f"{scene} [{solver}]: {B} envs x {steps} steps in {dt:.1f} s = {B*steps/dt/1e6:.2f} M env-steps/s; "
          f"flags: rows {float(((fl & 1) != 0).float().mean()):.3f} contacts {float(((fl & 2) != 0).float().mean()):.3f} "
          f"(per 50-step launch: mean {np.mean(per_launch):.4f}, max {np.max(per_launch):.4f}) "
          f"bad-state resets {float(((fl & 4) != 0).float().mean()):.4f}; pipeline timeouts {int(((fl & 8) != 0).sum())}; "
          f"steps per env min {int(sim.nstep.min())} max {int(sim.nstep.max())}; |quat|-1 max {worst_q:.1e}; "
          f"base z in [{float(z.min()):.3f}, {float(z.max()):.3f}], upright (R22>0.9) {float((up > 0.9).float().mean()):.3f}, "
          f"|x|,|y| max {float(sim.qpos[0:2].abs().max()):.1f} m; highest base z of the run {zmax_ever:.3f} (launch {zmax_at}: {zmax_who}); {options or ''} solver at its iteration cap on the last step of a launch: {capped} of {B * (steps // 50)} (env, launch) samples; union of all flag bits {hex(int(np.bitwise_or.reduce(fl.cpu().numpy())))} (satellite builds: include/smj.h names the capacity bits)")


if __name__ == "__main__":
    main(steps=int(sys.argv[1]) if len(sys.argv) > 1 else 20000, scene=sys.argv[2] if len(sys.argv) > 2 else "stretch_empty", solver=sys.argv[3] if len(sys.argv) > 3 else "newton",
         options={a.split("=")[0]: float(a.split("=")[1]) for a in sys.argv[4:] if not a.startswith(("capture=", "seed="))},
         seed=int(next((a.split("=", 1)[1] for a in sys.argv[4:] if a.startswith("seed=")), 99)),
         capture=next((a.split("=", 1)[1] for a in sys.argv[4:] if a.startswith("capture=")), None))
