#!/bin/bash
# Round-4 rocprofv3 evidence: the bench command (kernel trace + PMC passes, each counter group its own run, never combined with the
# sys / hip / hsa trace domains), the kitchen at Robocasa scale on the satellite builds (trace + PMC), per-stage cycle tables.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
CMD="python bench.py --no-second-solver --no-cpu-baseline --no-extra"
$CMD > gpurun_out/prof/bench_plain.log 2>&1; tail -1 gpurun_out/prof/bench_plain.log | cut -c1-200
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/trace -o smj -- $CMD > gpurun_out/prof/bench_trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/prof/pmc_$tag -o smj -- $CMD > gpurun_out/prof/pmc_$tag.log 2>&1
  echo "pmc $tag rc=$?"
done
KCMD="python tools/gpu_options_probe.py scene=stretch_kitchen_robocasa"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/rctrace -o smj -- $KCMD > gpurun_out/prof/rc_trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/prof/rcpmc_$tag -o smj -- $KCMD > gpurun_out/prof/rcpmc_$tag.log 2>&1
  echo "kitchen pmc $tag rc=$?"
done
for sc in stretch_kitchen4_sat stretch_scene_sat; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/trace_$sc -o smj -- python tools/gpu_options_probe.py scene=$sc > gpurun_out/prof/trace_$sc.log 2>&1
done
for sc in stretch_kitchen_robocasa stretch_kitchen4_sat stretch_scene_sat stretch_kitchen4 stretch_scene; do
  SMJ_LIB_PATH=$PWD/stretch_mujoco_amd/csrc/build/exp/libsmj_bigprof.so timeout 600 python tools/gpu_diag.py $sc 2>&1 | grep -v amdgpu.ids > gpurun_out/stage_cycles_$sc.txt
done
timeout 600 python tools/gpu_diag.py 2>&1 | grep -v amdgpu.ids > gpurun_out/stage_cycles.txt
python tools/gpu_sat_caps.py 2>&1 | grep -v amdgpu > gpurun_out/sat_caps.txt
for sc in stretch_kitchen_robocasa stretch_kitchen4_sat stretch_kitchen4 stretch_scene_sat stretch_scene stretch_kitchen_export_sat; do python tools/gpu_options_probe.py scene=$sc 2>&1 | grep -v amdgpu; done > gpurun_out/scene_probes.txt
# PGS with constraint islands: throughput beside the dense builds, sweep sub-counters (one-wavefront profiling build), step(1)
bash tools/gpu_pgs_sat_probe.sh > gpurun_out/pgs_sat_probe.txt 2>&1
for sc in stretch_kitchen4_sat stretch_kitchen_robocasa; do
  SMJ_LIB_PATH=$PWD/stretch_mujoco_amd/csrc/build/exp/libsmj_bigprof.so timeout 400 python tools/gpu_pgs_diag.py $sc 2>&1 | grep -v amdgpu.ids > gpurun_out/pgs_stage_cycles_$sc.txt
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/trace_pgs_kitchen4_sat -o smj -- python tools/gpu_options_probe.py solver=0 scene=stretch_kitchen4_sat > gpurun_out/prof/trace_pgs_kitchen4_sat.log 2>&1
for o in balance_min=4 balance_min=1; do python tools/gpu_step1_probe.py $o 2>&1 | grep -v amdgpu; done > gpurun_out/step1_probe.txt
bash tools/gpu_step1_trace.sh > gpurun_out/step1_trace.txt 2>&1
python tools/gpu_steplen_probe.py 2>&1 | grep -v amdgpu > gpurun_out/steplen.txt
find gpurun_out/prof -name "*.csv" | wc -l
