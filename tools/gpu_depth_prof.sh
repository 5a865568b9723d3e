#!/bin/bash
# per-call durations of the depth kernels (rocprofv3 kernel trace): empty scene (tools/gpu_depth_bench.py) and kitchen stand-in
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_r
for w in "empty:tools/gpu_depth_bench.py" "kitchen:tools/gpu_render_prof.py"; do
tag=${w%%:*}; script=${w#*:}
rm -rf gpurun_out/prof_r/*
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r/trace -o smj -- python $script 4096 > gpurun_out/prof_r/trace.log 2>&1
f=$(find gpurun_out/prof_r/trace -name "*kernel_trace.csv" | head -1)
python - "$f" $tag <<'PY'
import csv,sys,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Kernel_Name"]
    for k in ("meshlet","depth_kernel","fill_kernel","prepass"):
        if k in n: d[k].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
for k,v in d.items(): print(sys.argv[2], k, "calls", len(v), "ms:", " ".join("%.2f"%x for x in v))
PY
done
timeout 900 python -m pytest tests/test_gpu_depth.py tests/test_gpu_kitchen.py -m gpu -q -x 2>&1 | tail -3
