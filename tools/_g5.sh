cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/pmcq; export TMPDIR=/tmp
timeout 300 python tools/_sched.py 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -3
CMD="python bench.py --no-second-solver --no-cpu-baseline --no-extra"
for C in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmcq/$C -o smj -- $CMD > gpurun_out/pmcq/$C.log 2>&1; done
python - <<'PY'
import csv,glob
for C in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob(f"gpurun_out/pmcq/{C}/**/*counter_collection.csv",recursive=True)[0]
    rows=[r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("smj_step_kernel(") and r["Counter_Name"]==C]
    v=[float(r["Counter_Value"]) for r in rows][-10:]
    print(C, "mean KB per launch (last 10):", sum(v)/len(v))
PY
