#!/bin/bash
# Round 6: the 16-satellite Newton builds at other row capacities (csrc/build/exp/libsmj_cap_<rows>_<dense>.so): which capacity binds
# and what the kitchen's random-action throughput is at 4096 / 1024 / 512 envs.
cd "$GRAFT_REPO_ROOT" || exit 1
for lib in "" $PWD/stretch_mujoco_amd/csrc/build/exp/libsmj_cap_192_112.so $PWD/stretch_mujoco_amd/csrc/build/exp/libsmj_cap_176_128.so; do
  export SMJ_LIB_PATH=$lib
  echo "== ${lib:-default (208 rows / 96 dense)}"
  python tools/gpu_sat_caps.py stretch_kitchen_robocasa 2>&1 | grep -v amdgpu
  for e in 4096 1024 512; do python tools/gpu_options_probe.py scene=stretch_kitchen_robocasa envs=$e 2>&1 | grep -v amdgpu | cut -c1-300; done
  python tools/gpu_options_probe.py scene=stretch_kitchen4_sat 2>&1 | grep -v amdgpu | cut -c1-300
done
