cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -s > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -E "passed|failed|^FAILED|^ERROR|Error" gpurun_out/pytest_gpu.log | tail -20
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log; tail -3 gpurun_out/smoke.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log > gpurun_out/bench_latest.json; cut -c1-300 gpurun_out/bench_latest.json
