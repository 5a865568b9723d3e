#!/bin/bash
# first GPU contact: smoke, a small timing sweep, then pytest -m gpu
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
tail -5 gpurun_out/smoke.log
timeout 600 python bench.py --steps 200 --warmup 50 --envs-per-gpu 1024 --cpu-seconds 3 > gpurun_out/bench_1024.log 2>&1; tail -2 gpurun_out/bench_1024.log
timeout 900 python bench.py --steps 200 --warmup 50 --envs-per-gpu 4096 --cpu-seconds 3 > gpurun_out/bench_4096.log 2>&1; tail -2 gpurun_out/bench_4096.log
