cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kitchen.py -m gpu -x -q -s > gpurun_out/pytest_kitchen.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_kitchen.log
grep -E "passed|failed|^FAILED|^ERROR|Error|fault" gpurun_out/pytest_kitchen.log | tail -8
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "escalation or full_batch" > gpurun_out/pytest_esc.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_esc.log
grep -E "passed|failed|^FAILED|^ERROR|Error|fault" gpurun_out/pytest_esc.log | tail -8
