import sys, time, torch
sys.path.insert(0, ".")
from stretch_mujoco_amd import StretchBatchSimulator
for opts in ({'chunk':0},{'chunk':25},{'chunk':10},{'chunk':5},{'chunk':10,'escalate':0},{'chunk':10,'multiccd':0}):
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
    print(opts, '%.2f M env-steps/s'%(4096*500/dt/1e6), 'flagged', float((sim.info[3]!=0).float().mean()), flush=True)
    sim.stop()
