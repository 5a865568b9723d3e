cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python tools/_poll.py > gpurun_out/poll.log 2>&1; echo rc=$? >> gpurun_out/poll.log; cat gpurun_out/poll.log
