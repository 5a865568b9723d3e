#!/bin/bash
# PGS on the satellite builds (constraint islands): throughput beside the dense builds, Newton for reference
cd "$GRAFT_REPO_ROOT" || exit 1
for sc in stretch_kitchen4_sat stretch_kitchen4 stretch_scene_sat stretch_scene stretch_kitchen_robocasa; do
  timeout 600 python tools/gpu_options_probe.py solver=0 scene=$sc 2>&1 | grep -v amdgpu
done
for sc in stretch_kitchen4_sat stretch_kitchen_robocasa; do
  timeout 600 python tools/gpu_options_probe.py scene=$sc 2>&1 | grep -v amdgpu
done
