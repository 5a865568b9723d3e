#!/bin/bash
# round-3 experiment: dispatch options of the two-envs-per-CU big builds
cd "$GRAFT_REPO_ROOT" || exit 1
for sc in stretch_scene stretch_kitchen4; do
  for o in "pipeline_big=0" "pipeline=3" "pipeline=5" "pipeline=8" "pipeline=5 balance=0"; do
    timeout 300 python tools/gpu_options_probe.py scene=$sc $o 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
