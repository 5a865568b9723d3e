"""The reference's examples/start_pose.py (a start translation and rotation for the robot) with one pose PER ENV, and its
examples/world_frames.py counterpart on the query side: where the base, the gripper and a named link are, per env.

    python examples/start_pose_batch.py [num_envs]
"""
import math
import sys

import torch

sys.path.insert(0, ".")
from stretch_mujoco_amd import StretchBatchSimulator  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
yaw = torch.linspace(-math.pi / 2, math.pi / 2, B)
xy = torch.stack([0.1 * torch.arange(B), -0.2 * torch.ones(B), torch.zeros(B)], 1)                    # start_pose.py: (0.1, -0.2, 0), yaw -90 deg
quat = torch.stack([torch.cos(yaw / 2), torch.zeros(B), torch.zeros(B), torch.sin(yaw / 2)], 1)     # w x y z, like MuJoCo
sim = StretchBatchSimulator(num_envs=B, device="cuda:0", start_translation=xy.tolist(), start_rotation_quat=quat.tolist())
sim.start()
x, y, theta = sim.get_base_pose()                       # three [B] tensors, like the reference's (x, y, theta) tuple
print("base pose per env:", [[round(float(a), 3), round(float(b), 3), round(float(c), 3)] for a, b, c in zip(x, y, theta)])
ee = sim.get_ee_pose()                                  # [B, 4, 4] world pose of the gripper's grasp centre
print("gripper position per env:", [[round(float(v), 3) for v in T[:3, 3]] for T in ee])
head = sim.get_link_pose("link_head_tilt")             # forward kinematics at the status joint positions, as the reference does through the URDF
print("link_head_tilt height per env:", [round(float(T[2, 3]), 3) for T in head])
sim.stop()
