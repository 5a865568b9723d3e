#!/bin/bash
# instruction-cache evidence for the step kernel: rocprofv3 PMC passes (their own runs, kernel trace only)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
CMD="python bench.py --no-second-solver --no-cpu-baseline --no-extra --steps 100 --warmup 50"
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"; do
  tag=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/prof/ic_$tag -o smj -- $CMD > gpurun_out/prof/ic_$tag.log 2>&1
  echo "pmc $tag rc=$?"
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/prof/ic_*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "smj_step_kernel" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(f.split("/")[2], {k: (v / max(n[k], 1)) for k, v in acc.items()}, "launches", max(n.values()) if n else 0)
PY
