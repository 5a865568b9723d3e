"""The flow of the reference's examples/draw_circles.py -- the gripper traces a circle in the arm / lift plane with blocking moves --
for a batch of robots at once, each with its own diameter.

    python examples/draw_circles_batch.py [num_envs] [points]
"""
import math
import sys

import torch

sys.path.insert(0, ".")
from stretch_mujoco_amd import Actuators, StretchBatchSimulator  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 13
sim = StretchBatchSimulator(num_envs=B, device="cuda:0")
sim.start()                                            # allocate, reset, home()
sim.move_to(Actuators.head_tilt, -1.5707)              # as the reference: look at the gripper
sim.move_to(Actuators.head_pan, -0.7853)
sim.move_to(Actuators.wrist_yaw, 1.5707)
sim.move_to(Actuators.gripper, -0.15)
sim.wait_until_at_setpoint(Actuators.wrist_yaw)
st = sim.pull_status()
arm0, lift0 = st.arm.pos.clone(), st.lift.pos.clone()
diameter = torch.linspace(0.08, 0.2, B, device=sim.device)       # draw_circles.py draws 0.2 m
worst = torch.zeros(B, device=sim.device)
for k in range(N):
    t = 2 * math.pi * k / (N - 1)
    sim.move_to(Actuators.arm, arm0 + diameter / 2 * (1 - math.cos(t)))      # (start ON the circle: the first point is where the arm is; the circle lies outward, inside the arm's range)
    sim.move_to(Actuators.lift, lift0 + diameter / 2 * math.sin(t))
    ok = sim.wait_until_at_setpoint(Actuators.arm) & sim.wait_until_at_setpoint(Actuators.lift)   # [B] bool, sim clock, 0.05 tolerance like the reference
    st = sim.pull_status()
    r = torch.sqrt((st.arm.pos - (arm0 + diameter / 2)) ** 2 + (st.lift.pos - lift0) ** 2)
    worst = torch.maximum(worst, (r - diameter / 2).abs())
    if k % 4 == 0:
        print(f"point {k:2d}: reached in {int(ok.sum())} of {B} envs; radius error max {float((r - diameter / 2).abs().max()):.4f} m")
print("largest distance from its circle, per env [m]:", [round(float(v), 4) for v in worst])
sim.home()
sim.stop()
