"""GPU diagnostics: stage-by-stage parity of one step against the fp64 oracle + per-stage cycle profile."""
import sys, os, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stretch_mujoco_amd import StretchBatchSimulator
from oracle.oracle import Oracle
np.set_printoptions(precision=5, suppress=True, linewidth=200)
PROF = ["kin", "comcrb", "smooth", "factor", "collision", "makecon", "project", "warm", "pgs", "post", "integrate", "total", "sweeps", "setup", "n_update", "n_grad", "n_xa", "n_hmfma", "n_gauss_jordan", "n_solve", "n_prep", "n_ls", "n_lsevals", "c_pose", "c_sphere", "c_obb", "c_narrow", "c_nsphere", "c_nobb", "c_nhit", "c_nmulti", "c_tboxbox", "c_tmpr1", "c_tmulti", "c_rounds", "h_k", "h_cone", "h_store", "h_nks", "_", "s_broad", "s_narrow", "mc_rows", "mc_con", "mc_jac", "mc_items", "mc_imp", "sat_h"]

def stage_parity(nsteps=1):
    sim = StretchBatchSimulator(num_envs=4, device="cuda:0", debug=True)
    sim.start(home=False)
    ctrl = np.array([2, -1, 0.6, 0.1, 1, -0.4, 0.5, 0.02, 0.3, -0.2], np.float32)
    sim.ctrl[:] = torch.tensor(ctrl, device=sim.device).unsqueeze(1)
    o = Oracle(sim._blob); o.arr("ctrl")[:] = ctrl
    if nsteps > 1:
        o.step(nsteps - 1)
    o.forward()
    sim.step(nsteps); torch.cuda.synchronize()
    dbg = sim.debug[:, 0].cpu().numpy(); ne = o.nefc
    r = {}
    r["M"] = abs(dbg[0:1024].reshape(32, 32)[:26, :26] - o.arr("qM").reshape(26, 26)).max()
    r["xpos"] = abs(dbg[1408:1468].reshape(20, 3) - o.arr("xpos")).max()
    r["bias"] = abs(dbg[1504:1530] - o.arr("qfrc_bias")).max()
    r["passive"] = abs(dbg[1536:1562] - o.arr("qfrc_passive")).max()
    r["act"] = abs(dbg[1568:1594] - o.arr("qfrc_actuator")).max()
    r["nefc"] = (int(sim.info[0, 0]), ne); r["niter"] = (int(sim.info[2, 0]), int(o.iarr("solver_niter")[0]))
    r["R"] = abs(dbg[1216:1216 + ne] - o.arr("efc_R")).max() / abs(o.arr("efc_R")).max()
    r["aref"] = abs(dbg[1280:1280 + ne] - o.arr("efc_aref")).max()
    r["b"] = abs(dbg[1152:1152 + ne] - o.arr("efc_b")).max()
    AR = dbg[1728:1728 + 4096].reshape(64, 64)[:ne, :ne]; ARo = o.arr("efc_AR").reshape(ne, ne)
    r["AR"] = abs(AR - ARo).max(); r["ARmax"] = abs(ARo).max()
    r["force"] = abs(dbg[1088:1088 + ne] - o.arr("efc_force")).max()
    r["qacc"] = abs(dbg[1056:1082] - o.arr("qacc")).max(); r["qacc_max"] = abs(o.arr("qacc")).max()
    print(f"stage parity after {nsteps} step(s):", {k: (float(v) if not isinstance(v, tuple) else v) for k, v in r.items()})
    ident = float((sim.qpos - sim.qpos[:, :1]).abs().max())
    print("env-to-env max diff (identical inputs):", ident)
    sim.stop()

def profile(B, random_ctrl, steps=50, solver="pgs", scene=None):
    sim = StretchBatchSimulator(num_envs=B, device="cuda:0", debug=True, solver=solver, **({"scene": scene} if scene else {}))
    sim.start(home=False)
    import os
    if os.environ.get("SMJ_REP"): sim.set_option("pgs_fixed_iter", int(os.environ["SMJ_REP"]))
    if os.environ.get("SMJ_NEWTON_TWO_WAVES"):   # 0: the one-wavefront kernel of the 16-satellite build; 1 (default): two wavefronts -- the table then shows the FIRST wavefront's cycles (sat_h: its wait for the second one's blocks)
        sim.set_option("newton_two_waves", int(os.environ["SMJ_NEWTON_TWO_WAVES"]))
    dev = sim.device
    sim.ctrl[:] = torch.tensor(sim.model["key_ctrl"][0, :10], dtype=torch.float32, device=dev).unsqueeze(1)
    sim.step(500)
    if random_ctrl:
        g = torch.Generator(device=dev).manual_seed(1)
        lo = torch.tensor(sim.model["actuator_ctrlrange"][:, 0], dtype=torch.float32, device=dev).unsqueeze(1)
        hi = torch.tensor(sim.model["actuator_ctrlrange"][:, 1], dtype=torch.float32, device=dev).unsqueeze(1)
        sim.ctrl.copy_(lo + (hi - lo) * torch.rand(10, B, generator=g, device=dev))
        sim.step(100)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); sim.step(steps); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    p = sim.prof.cpu().numpy() / steps
    info = sim.info.cpu().numpy()
    print(f"[{solver}] B={B} random={random_ctrl}: launch {ms:.2f} ms for {steps} steps -> {B*steps/ms*1e3:.0f} env-steps/s; nefc mean {info[0].mean():.1f} max {info[0].max()}, ncon mean {info[1].mean():.1f}, flags {np.bitwise_or.reduce(info[3])}")
    print("  cycles/step (mean over envs):", {n: int(p[i].mean()) for i, n in enumerate(PROF)})
    print("  counts/step (mean over envs):", {n: round(float(p[i].mean()), 3) for i, n in enumerate(PROF) if n.startswith("c_n") or n in ("sweeps", "n_lsevals", "c_sphere", "h_nks", "c_rounds")})
    hist = np.bincount(np.round(p[PROF.index("c_nobb")] * steps).astype(int) // steps, minlength=6)
    print("  envs by narrowphase calls per step (floor):", hist.tolist())
    print("  cycles/step (max over envs): ", {n: int(p[i].max()) for i, n in enumerate(PROF)})
    sim.stop()

if __name__ == "__main__":
    if len(sys.argv) > 1:   # python tools/gpu_diag.py <scene>: the cycle table of another scene (a build whose variant has the counters)
        profile(1024, False, solver="newton", scene=sys.argv[1])
        profile(1024, True, solver="newton", scene=sys.argv[1])
        sys.exit(0)
    for solver in ("newton",):
        profile(1024, False, solver=solver)
        profile(1024, True, solver=solver)
    # Newton parity on the GPU vs the fp64 Newton oracle, from reset
    sim = StretchBatchSimulator(num_envs=2, device="cuda:0", solver="newton"); sim.start(home=False)
    ctrl = np.array([2, -1, 0.6, 0.1, 1, -0.4, 0.5, 0.02, 0.3, -0.2], np.float32)
    sim.ctrl[:] = torch.tensor(ctrl, device=sim.device).unsqueeze(1)
    o = Oracle(sim._blob); o.set_option("solver", 2); o.arr("ctrl")[:] = ctrl
    q0 = o.arr("qpos").copy(); q0[9] = 0.6; q0[10:14] = 0.025; o.arr("qpos")[:] = q0
    sim.qpos[:] = torch.tensor(q0, dtype=torch.float32, device=sim.device).unsqueeze(1)
    for k in range(10):
        o.step(100); sim.step(100); torch.cuda.synchronize()
        print("newton gpu-vs-oracle step", (k + 1) * 100, "max|dq| = %.2e" % np.abs(sim.qpos[:, 0].cpu().numpy() - o.arr("qpos")).max(), "iters", int(sim.info[2, 0]), int(o.iarr("solver_niter")[0]))
