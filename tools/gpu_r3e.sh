#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/gpu_probe4.sh
bash tools/gpu_r3d.sh
