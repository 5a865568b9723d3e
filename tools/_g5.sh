cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python tools/_pipe.py > gpurun_out/pipe.log 2>&1; echo rc=$? >> gpurun_out/pipe.log; cat gpurun_out/pipe.log
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "escalation or full_batch" 2>&1 | tail -5
