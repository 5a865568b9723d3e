#!/bin/bash
# throughput probes of the four scenes (settled / random-action env-steps/s, flagged fraction); "$@" = smj_set_option pairs
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for sc in stretch_empty stretch_kitchen_standin stretch_scene stretch_kitchen4; do
  timeout 300 python tools/gpu_options_probe.py scene=$sc "$@" 2>&1 | grep -v amdgpu.ids | tail -1
done | tee gpurun_out/probe.log
