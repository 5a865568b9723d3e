#!/bin/bash
# PMC passes over the depth kernels (own runs, kernel trace only beside them)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-30)
  rm -rf gpurun_out/prof_r/pmc_$tag
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/prof_r/pmc_$tag -o smj -- python ${1:-tools/gpu_depth_bench.py} 4096 > gpurun_out/prof_r/pmc_$tag.log 2>&1
  f=$(find gpurun_out/prof_r/pmc_$tag -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter(); seen=set()
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")[:32]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if (r["Dispatch_Id"],k) not in seen: seen.add((r["Dispatch_Id"],k)); n[k]+=1
for k,v in agg.items():
    if any(x in k for x in ("meshlet","depth_kernel","lidar")): print(k, "dispatches", n[k], {a: f"{b/n[k]:.3g}" for a,b in v.items()})
PY
done
