cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
timeout 300 python tools/_sched.py
for sc in stretch_kitchen_standin stretch_scene stretch_kitchen4; do timeout 600 python tools/_sched.py scene=$sc; done
} 2>&1 | grep -v amdgpu.ids
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3
