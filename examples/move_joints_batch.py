"""The flow of the reference's examples/move_joints.py (lift_sequence), for a batch of robots at once.

    python examples/move_joints_batch.py [num_envs]
"""
import sys

import torch

sys.path.insert(0, ".")
from stretch_mujoco_amd import Actuators, StretchBatchSimulator  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sim = StretchBatchSimulator(num_envs=B, device="cuda:0")
sim.start()                                           # allocate, reset, home()
sim.stow()

LIFT_START, MOVE_BY = 0.1, 0.5
sim.move_to(Actuators.lift, LIFT_START)
reached = sim.wait_until_at_setpoint(Actuators.lift)  # [B] bool, sim clock
start = sim.pull_status().lift.pos.clone()
print("reached start:", int(reached.sum()), "of", B, " lift =", start[:4].tolist())

# every env moves by its own amount: commands take [B] tensors (or scalars, or env_ids subsets)
by = torch.linspace(0.1, MOVE_BY, B, device=sim.device)
sim.move_by(Actuators.lift, by)
sim.wait_while_is_moving(Actuators.lift)
now = sim.pull_status().lift.pos
print("moved by:", (now - start)[:4].tolist(), " asked:", by[:4].tolist())

sim.set_base_velocity(v_linear=0.3, omega=0.5, env_ids=[0, 1])
sim.step(500)                                         # 1 s of sim time; the caller owns the clock
x, y, th = sim.get_base_pose()
print("base pose of env 0 / env 2:", [float(x[0]), float(y[0]), float(th[0])], [float(x[2]), float(y[2]), float(th[2])])
print("grasp centre of env 0 (world):", sim.get_ee_pose()[0, :3, 3].tolist())
sim.stop()
