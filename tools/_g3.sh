cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -s > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -E "passed|failed|^FAILED|^ERROR|Error" gpurun_out/pytest_gpu.log | tail -20
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log > gpurun_out/bench_latest.json; cut -c1-300 gpurun_out/bench_latest.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_latest.json').read())
r=d['roofline']
print([ (l['tag'][0],l['ms']) for l in r['launches']])
print(d.get('other_solver')); print({k:(v.get('value'),v.get('envs_flagged')) for k,v in d['other_configs'].items()})
PY
for esc in 0; do python - <<'PY'
import sys, time, torch
sys.path.insert(0,'.')
from stretch_mujoco_amd import StretchBatchSimulator
for opts in ({'escalate':0},{'multiccd':0,'escalate':0},{'multiccd':0}):
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
PY
done
