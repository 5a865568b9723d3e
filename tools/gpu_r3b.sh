#!/bin/bash
# round-3 measurement call B: per-stage cycle tables of the big variant (profiling build) + the kitchen parity tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for sc in stretch_scene stretch_kitchen4; do
  SMJ_LIB_PATH=$PWD/stretch_mujoco_amd/csrc/build/exp/libsmj_bigprof.so timeout 600 python tools/gpu_diag.py $sc 2>&1 | grep -v amdgpu.ids > gpurun_out/stage_cycles_$sc.txt
  grep -v "max over\|counts/step\|envs by" gpurun_out/stage_cycles_$sc.txt | cut -c1-900
done
timeout 900 python -m pytest tests/test_gpu_kitchen.py tests/test_rollout_parity.py -m gpu -q -x 2>&1 | tail -5
