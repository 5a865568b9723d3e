import sys, os, numpy as np, torch
sys.path.insert(0, ".")
sys.path.insert(0, "tools")
from stretch_mujoco_amd import StretchBatchSimulator, model_blob
PROF = ["kin", "comcrb", "smooth", "factor", "collision", "makecon", "project", "warm", "pgs", "post", "integrate", "total", "sweeps", "setup", "n_update", "n_grad", "n_xa", "n_hmfma", "n_gauss_jordan", "n_solve", "n_prep", "n_ls", "n_lsevals"]
def profile(B, scene, opts, steps=50, force=None):
    if force:
        m = model_blob.loads(open(f"stretch_mujoco_amd/models/{scene}.smjb", "rb").read())
        m["k_capacity_hint"] = np.array([1], np.int32)
        sim = StretchBatchSimulator(num_envs=B, device="cuda:0", debug=True, model_blob_bytes=model_blob.dumps(m))
    else:
        sim = StretchBatchSimulator(num_envs=B, device="cuda:0", debug=True, scene=scene)
    sim.start(home=False)
    for k, v in opts.items(): sim.set_option(k, v)
    dev = sim.device
    sim.ctrl[:] = torch.tensor(sim.model["key_ctrl"][0, :10], dtype=torch.float32, device=dev).unsqueeze(1)
    sim.step(500)
    g = torch.Generator(device=dev).manual_seed(1)
    lo = torch.tensor(sim.model["actuator_ctrlrange"][:, 0], dtype=torch.float32, device=dev).unsqueeze(1)
    hi = torch.tensor(sim.model["actuator_ctrlrange"][:, 1], dtype=torch.float32, device=dev).unsqueeze(1)
    for _ in range(3):
        sim.ctrl.copy_(lo + (hi - lo) * torch.rand(10, B, generator=g, device=dev)); sim.step(50)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sim.ctrl.copy_(lo + (hi - lo) * torch.rand(10, B, generator=g, device=dev))
    e0.record(); sim.step(steps); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    p = sim.prof.cpu().numpy() / steps
    info = sim.info.cpu().numpy()
    print(f"[{scene} {opts} force={force}] caps {sim.nv_max}/{sim.nefc_max}/{sim.ncon_max} B={B}: {ms:.2f} ms / {steps} steps -> {B*steps/ms*1e3/1e6:.2f} M env-steps/s; nefc mean {info[0].mean():.1f} max {info[0].max()}, ncon mean {info[1].mean():.1f}, flagged {(info[3]!=0).mean():.4f}")
    print("  cycles/step mean:", {n: int(p[i].mean()) for i, n in enumerate(PROF) if p[i].mean() > 0})
    sim.stop()
profile(1024, "stretch_empty", {"escalate": 0})
profile(1024, "stretch_empty", {"escalate": 0, "multiccd": 0})
profile(512, "stretch_empty", {}, force="tall")
profile(512, "stretch_kitchen_standin", {})
profile(256, "stretch_scene", {})
profile(256, "stretch_kitchen4", {})
import time
for opts in ({'balance':0},{'balance':1},{'balance':1,'escalate':0}):
    sim=StretchBatchSimulator(num_envs=4096, device='cuda:0'); sim.start(home=False)
    for k,v in opts.items(): sim.set_option(k,v)
    dev=sim.device
    lo=torch.tensor(sim.model["actuator_ctrlrange"][:,0],dtype=torch.float32,device=dev).unsqueeze(1); hi=torch.tensor(sim.model["actuator_ctrlrange"][:,1],dtype=torch.float32,device=dev).unsqueeze(1)
    g=torch.Generator(device=dev).manual_seed(1234)
    sim.ctrl[:]=torch.tensor([0,0,0.6,0.1,0,0,0,0,0,0],dtype=torch.float32,device=dev).unsqueeze(1); sim.step(500)
    for _ in range(4): sim.ctrl.copy_(lo+(hi-lo)*torch.rand(10,4096,generator=g,device=dev)); sim.step(50)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(10): sim.ctrl.copy_(lo+(hi-lo)*torch.rand(10,4096,generator=g,device=dev)); sim.step(50)
    torch.cuda.synchronize(); dt=time.perf_counter()-t
    print(opts, '%.2f M env-steps/s'%(4096*500/dt/1e6), 'flagged', float((sim.info[3]!=0).float().mean()))
    sim.stop()
