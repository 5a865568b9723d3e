#!/bin/bash
# round-3 measurement call A: GPU tests, then the per-stage cycle tables of the big variant (profiling build)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
for sc in stretch_scene stretch_kitchen4; do
  SMJ_LIB_PATH=$PWD/stretch_mujoco_amd/csrc/build/exp/libsmj_bigprof.so timeout 600 python tools/gpu_diag.py $sc 2>&1 | grep -v amdgpu.ids > gpurun_out/stage_cycles_$sc.txt
  cat gpurun_out/stage_cycles_$sc.txt | cut -c1-1200
done
