#!/bin/bash
# scratch: stage cycles with narrowphase sub-counters for the three scenes
mkdir -p gpurun_out
for sc in stretch_scene stretch_kitchen4; do
  SMJ_LIB_PATH=$PWD/stretch_mujoco_amd/csrc/build/exp/libsmj_bigprof.so timeout 600 python tools/gpu_diag.py $sc 2>&1 | grep -v amdgpu.ids > gpurun_out/stage_cycles_$sc.txt
done
timeout 600 python tools/gpu_diag.py 2>&1 | grep -v amdgpu.ids > gpurun_out/stage_cycles.txt
grep -h "random=True" -A2 gpurun_out/stage_cycles*.txt | grep -o "'c_tboxbox.*\|counts.step.*" 
