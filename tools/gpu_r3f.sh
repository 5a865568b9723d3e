#!/bin/bash
# round-3 call F: GPU tests + throughput probes of every scene
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
for sc in stretch_empty stretch_kitchen_standin stretch_scene stretch_kitchen4; do
  timeout 300 python tools/gpu_options_probe.py scene=$sc 2>&1 | grep -v amdgpu.ids | tail -1
done | tee gpurun_out/probe.log
