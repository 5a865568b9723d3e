cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
timeout 300 python tools/_sched.py
timeout 300 python tools/_sched.py pollers=1
timeout 300 python tools/_sched.py pollers=-2
} > gpurun_out/sched.log 2>&1; grep -v amdgpu.ids gpurun_out/sched.log | tail -60
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log > gpurun_out/bench_latest.json; cut -c1-200 gpurun_out/bench_latest.json
