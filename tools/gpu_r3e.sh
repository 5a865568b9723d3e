#!/bin/bash
# round-3 call E: GPU tests + PGS throughput / flagged fraction on the bench workload
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
for o in "solver=0" "solver=0 escalate=0" "solver=2"; do
  timeout 300 python tools/gpu_options_probe.py $o 2>&1 | grep -v amdgpu.ids | tail -1
done | tee gpurun_out/probe_pgs.log
