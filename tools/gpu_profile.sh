#!/bin/bash
# rocprofv3 evidence: kernel-trace stats of the bench command, then PMC passes in their own runs (never combined with the
# sys / hip / hsa trace domains); the same for the ray-casting kernels (tools/gpu_render_prof.py).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
CMD="python bench.py --no-second-solver --no-cpu-baseline --no-extra"
$CMD > gpurun_out/prof/bench_plain.log 2>&1; tail -1 gpurun_out/prof/bench_plain.log | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/trace -o smj -- $CMD > gpurun_out/prof/bench_trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/prof/pmc_$tag -o smj -- $CMD > gpurun_out/prof/pmc_$tag.log 2>&1
  echo "pmc $tag rc=$?"
done
RCMD="python tools/gpu_render_prof.py"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/rtrace -o smj -- $RCMD > gpurun_out/prof/render_trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/prof/rpmc_$tag -o smj -- $RCMD > gpurun_out/prof/rpmc_$tag.log 2>&1
  echo "render pmc $tag rc=$?"
done
# the kitchen stand-in (tall variant as the primary kernel) and the PGS path: kernel-trace stats only
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/ktrace -o smj -- python tools/gpu_options_probe.py scene=stretch_kitchen_standin > gpurun_out/prof/kitchen_trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/ptrace -o smj -- python tools/gpu_options_probe.py solver=0 > gpurun_out/prof/pgs_trace.log 2>&1
# the big variant's two-envs-per-CU builds: the reference's own scene (38 dof columns) and the kitchen with four free objects (50):
# kernel-trace stats, then PMC passes (own runs) on the kitchen
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/strace -o smj -- python tools/gpu_options_probe.py scene=stretch_scene > gpurun_out/prof/scene_trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/k4trace -o smj -- python tools/gpu_options_probe.py scene=stretch_kitchen4 > gpurun_out/prof/kitchen4_trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/prof/k4pmc_$tag -o smj -- python tools/gpu_options_probe.py scene=stretch_kitchen4 > gpurun_out/prof/k4pmc_$tag.log 2>&1
  echo "kitchen4 pmc $tag rc=$?"
done
# per-stage cycle tables of the big builds (a copy of the library whose big builds carry the counters: make bigprof)
for sc in stretch_scene stretch_kitchen4; do
  SMJ_LIB_PATH=$PWD/stretch_mujoco_amd/csrc/build/exp/libsmj_bigprof.so timeout 600 python tools/gpu_diag.py $sc 2>&1 | grep -v amdgpu.ids > gpurun_out/stage_cycles_$sc.txt
done
find gpurun_out/prof -name "*.csv" | wc -l
