#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for t in none maxilp minreg; do
  SMJ_LIB_PATH=$PWD/stretch_mujoco_amd/csrc/build/exp/libsmj_big50_$t.so timeout 300 python tools/gpu_options_probe.py scene=stretch_kitchen4 2>&1 | grep -v amdgpu.ids | tail -1
done
timeout 300 python tools/gpu_options_probe.py scene=stretch_kitchen4 2>&1 | grep -v amdgpu.ids | tail -1
