import sys, time, torch
sys.path.insert(0, ".")
from stretch_mujoco_amd import StretchBatchSimulator
B = 4096
final = {}
for opts in ({'pipeline': 0}, {'pipeline': 10, 'pollers': 0}, {'pipeline': 10, 'pollers': 16}, {'pipeline': 5, 'pollers': 16}, {'pipeline': 5, 'pollers': 8}, {'pipeline': 5, 'pollers': 32}, {'pipeline': 10, 'escalate': 0}, {'pipeline': 5, 'pollers': 16, 'multiccd': 0}):
    sim = StretchBatchSimulator(num_envs=B, device='cuda:0'); sim.start(home=False)
    for k, v in opts.items(): sim.set_option(k, v)
    dev = sim.device
    lo = torch.tensor(sim.model["actuator_ctrlrange"][:, 0], dtype=torch.float32, device=dev).unsqueeze(1); hi = torch.tensor(sim.model["actuator_ctrlrange"][:, 1], dtype=torch.float32, device=dev).unsqueeze(1)
    g = torch.Generator(device=dev).manual_seed(1234)
    sim.ctrl[:] = torch.tensor([0, 0, 0.6, 0.1, 0, 0, 0, 0, 0, 0], dtype=torch.float32, device=dev).unsqueeze(1); sim.step(500)
    for _ in range(4): sim.ctrl.copy_(lo + (hi - lo) * torch.rand(10, B, generator=g, device=dev)); sim.step(50)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): sim.ctrl.copy_(lo + (hi - lo) * torch.rand(10, B, generator=g, device=dev)); sim.step(50)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    key = tuple(sorted((k, v) for k, v in opts.items() if k not in ('pipeline', 'pollers')))
    st = torch.cat([sim.qpos.flatten(), sim.qvel.flatten(), sim.qacc_warmstart.flatten(), sim.actuator_length.flatten(), sim.base_pose.flatten()]).clone()
    info = sim.info.clone(); ns = sim.nstep.clone()
    same = ''
    if key in final:
        a, b, c2 = final[key]
        same = 'bit-identical to pipeline=0: %s (info %s, nstep %s), max diff %.3g, nan %d' % (torch.equal(a, st), torch.equal(b, info), torch.equal(c2, ns), float((a - st).abs().nan_to_num().max()), int(torch.isnan(st).sum()))
    else:
        final[key] = (st, info, ns)
    print(opts, '%.2f M env-steps/s' % (B * 500 / dt / 1e6), 'flagged', float((sim.info[3] != 0).float().mean()), 'flags or', int(sim.info[3].max()), same, flush=True)
    sim.stop()
