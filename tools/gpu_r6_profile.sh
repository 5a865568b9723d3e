#!/bin/bash
# Round-6 rocprofv3 evidence: the bench command (kernel trace + PMC passes, each counter group its own run, never combined with the
# sys / hip / hsa trace domains), the kitchen at Robocasa scale on the satellite builds under Newton and PGS (trace + FETCH / WRITE),
# per-stage cycle tables, soaks with the solver-at-cap counts, step(n) against n.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
CMD="python bench.py --no-second-solver --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/trace -o smj -- $CMD > gpurun_out/prof/bench_trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/prof/pmc_$tag -o smj -- $CMD > gpurun_out/prof/pmc_$tag.log 2>&1
  echo "pmc $tag rc=$?"
done
for SOLV in "" "solver=0"; do
  t=rc${SOLV:+pgs}
  KCMD="python tools/gpu_options_probe.py scene=stretch_kitchen_robocasa $SOLV"
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/${t}trace -o smj -- $KCMD > gpurun_out/prof/${t}_trace.log 2>&1
  for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
    tag=$(echo $C | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/prof/${t}pmc_$tag -o smj -- $KCMD > gpurun_out/prof/${t}pmc_$tag.log 2>&1
    echo "kitchen $t pmc $tag rc=$?"
  done
done
for sc in stretch_kitchen_robocasa stretch_kitchen4_sat; do
  SMJ_NEWTON_TWO_WAVES=0 SMJ_LIB_PATH=$PWD/stretch_mujoco_amd/csrc/build/exp/libsmj_bigprof.so timeout 600 python tools/gpu_diag.py $sc 2>&1 | grep -v amdgpu.ids > gpurun_out/stage_cycles_$sc.txt
  SMJ_LIB_PATH=$PWD/stretch_mujoco_amd/csrc/build/exp/libsmj_bigprof.so timeout 600 python tools/gpu_diag.py $sc 2>&1 | grep -v amdgpu.ids > gpurun_out/stage_cycles_${sc}_two_waves.txt
  SMJ_LIB_PATH=$PWD/stretch_mujoco_amd/csrc/build/exp/libsmj_bigprof.so timeout 400 python tools/gpu_pgs_diag.py $sc 2>&1 | grep -v amdgpu.ids > gpurun_out/pgs_stage_cycles_$sc.txt
done
timeout 600 python tools/gpu_diag.py 2>&1 | grep -v amdgpu.ids > gpurun_out/stage_cycles.txt
for sc in stretch_kitchen_robocasa stretch_kitchen4_sat stretch_kitchen4 stretch_scene_sat stretch_scene stretch_kitchen_standin; do python tools/gpu_options_probe.py scene=$sc 2>&1 | grep -v amdgpu; done > gpurun_out/scene_probes.txt
for sc in stretch_kitchen_robocasa stretch_kitchen4_sat stretch_scene_sat stretch_kitchen_standin stretch_scene stretch_kitchen4; do python tools/gpu_options_probe.py scene=$sc solver=0 2>&1 | grep -v amdgpu; done >> gpurun_out/scene_probes.txt
python tools/gpu_options_probe.py solver=0 2>&1 | grep -v amdgpu >> gpurun_out/scene_probes.txt
python tools/gpu_steplen_probe.py 2>&1 | grep -v amdgpu > gpurun_out/steplen.txt
(python tools/gpu_soak.py 15000 stretch_kitchen_robocasa newton capture=gpurun_out/soak_capture_final.npz; python tools/gpu_soak.py 15000 stretch_kitchen_robocasa newton seed=7; python tools/gpu_soak.py 10000 stretch_kitchen_robocasa pgs; python tools/gpu_soak.py 10000 stretch_kitchen4_sat pgs; python tools/gpu_soak.py 10000 stretch_empty pgs; python tools/gpu_soak.py 20000 stretch_empty newton) 2>&1 | grep -v amdgpu > gpurun_out/soak.txt
# the ray-casting kernels (depth cameras + lidar) in the Robocasa-scale kitchen and the stand-in: per-kernel stats of the same workload
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/rtrace -o smj -- python tools/gpu_render_prof.py 4096 stretch_kitchen_robocasa > gpurun_out/prof/rtrace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/rtrace_standin -o smj -- python tools/gpu_render_prof.py 4096 stretch_kitchen_standin > gpurun_out/prof/rtrace_standin.log 2>&1
find gpurun_out/prof -name "*.csv" | wc -l
