cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log; tail -3 gpurun_out/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline --no-second-solver > gpurun_out/bench_driver.log 2>&1; tail -1 gpurun_out/bench_driver.log | cut -c1-300
timeout 600 python bench.py --envs 16384 --no-extra --no-cpu-baseline --no-second-solver > gpurun_out/bench_16k.log 2>&1; tail -1 gpurun_out/bench_16k.log | cut -c1-200
timeout 600 python bench.py --envs 8192 --no-extra --no-cpu-baseline --no-second-solver > gpurun_out/bench_8k.log 2>&1; tail -1 gpurun_out/bench_8k.log | cut -c1-200
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log > gpurun_out/bench_latest.json; cut -c1-300 gpurun_out/bench_latest.json
python - <<'PY'
import torch
p=torch.cuda.get_device_properties(0)
print(p)
import ctypes
hip=ctypes.CDLL('libamdhip64.so')
v=ctypes.c_int()
for name,a in (('maxSharedMemPerBlock',8),('maxSharedMemPerMultiprocessor', 81)):
    hip.hipDeviceGetAttribute(ctypes.byref(v), a, 0); print(name, v.value)
PY
