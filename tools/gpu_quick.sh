#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python tools/gpu_diag.py > gpurun_out/diag.log 2>&1; tail -22 gpurun_out/diag.log
