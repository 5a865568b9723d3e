#!/bin/bash
# round-3 measurement call C: the GPU tests, throughput probes of every scene, per-stage cycle tables of the big variant
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
for sc in stretch_empty stretch_kitchen_standin stretch_scene stretch_kitchen4; do
  timeout 300 python tools/gpu_options_probe.py scene=$sc 2>&1 | grep -v amdgpu.ids | tail -1
done | tee gpurun_out/probe.log
if [ "$1" != "noprof" ]; then
for sc in stretch_scene stretch_kitchen4; do
  SMJ_LIB_PATH=$PWD/stretch_mujoco_amd/csrc/build/exp/libsmj_bigprof.so timeout 600 python tools/gpu_diag.py $sc 2>&1 | grep -v amdgpu.ids > gpurun_out/stage_cycles_$sc.txt
  grep -v "max over\|counts/step\|envs by" gpurun_out/stage_cycles_$sc.txt | cut -c1-900
done
fi
