#!/bin/bash
# kernel trace of step(1) calls on the headline workload: what a one-step launch is made of
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/step1 -o smj -- python tools/gpu_step1_probe.py > gpurun_out/prof/step1.log 2>&1
python - <<'PY'
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/prof/step1/smj_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 300 step(1)-sized standard launches: find launches of smj_step_kernel( with small grid
std = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("smj_step_kernel(")]
# step(1) phase = the first 400 standard launches after the 6 warm launches of 50 steps; take a window in the middle
win = std[50:350]
a, b = win[0], win[-1]
seg = rows[a:b + 1]
dur = collections.defaultdict(list)
for r in seg: dur[r["Kernel_Name"][:48]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
n = len(win) - 1
print("window: %d step(1) calls, %.1f us per call wall (GPU timeline)" % (n, span / n))
busy = 0
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print("  %-50s calls/step %.2f  mean %.1f us  total/step %.1f us" % (k, len(v) / n, sum(v) / len(v), sum(v) / n)); busy += sum(v) / n
print("  sum of kernel time per call %.1f us, gaps %.1f us" % (busy, span / n - busy))
PY
