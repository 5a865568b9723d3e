cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for sc in stretch_kitchen_standin stretch_scene stretch_kitchen4; do for o in "pipeline=0" "pipeline=5" "pipeline=10"; do timeout 600 python tools/_sched.py scene=$sc $o; done; done
timeout 300 python tools/_sched.py
} > gpurun_out/sched.log 2>&1; grep -v amdgpu.ids gpurun_out/sched.log | tail -60
