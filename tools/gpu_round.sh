#!/bin/bash
# One GPU round on a gpurun box: parity tests, smoke, bench, rocprofv3 evidence (summaries: tools/summarize_prof.py).
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpu_round.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log; tail -2 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log > gpurun_out/bench_latest.json; cut -c1-400 gpurun_out/bench_latest.json
timeout 600 python tools/gpu_diag.py 2>&1 | grep -v amdgpu.ids > gpurun_out/stage_cycles.txt; tail -3 gpurun_out/stage_cycles.txt | cut -c1-200
if [ "$1" != "noprof" ]; then bash tools/gpu_profile.sh | tail -2; fi
