"""Pure helper functions of the glue layer, batched (work on floats, numpy arrays and torch tensors alike).

Mirrors stretch_mujoco/utils.py: diff_drive_fwd_kinematics (:94-114), diff_drive_inv_kinematics (:117-135),
map_between_ranges (:352-360), compute_K (:56-61), limit_depth_distance (:87-91).
"""
from __future__ import annotations

import math

import numpy as np

from . import config


def diff_drive_fwd_kinematics(w_left, w_right):
    R = config.robot_settings["wheel_diameter"] / 2
    L = config.robot_settings["wheel_separation"]
    if R <= 0:
        raise ValueError("Radius must be greater than zero.")
    if L <= 0:
        raise ValueError("Distance between wheels must be greater than zero.")
    V = R * (w_left + w_right) / 2.0
    omega = R * (w_right - w_left) / L
    return (V, omega)


def diff_drive_inv_kinematics(V, omega):
    R = config.robot_settings["wheel_diameter"] / 2
    L = config.robot_settings["wheel_separation"]
    if R <= 0:
        raise ValueError("Radius must be greater than zero.")
    if L <= 0:
        raise ValueError("Distance between wheels must be greater than zero.")
    w_left = (V - (omega * L / 2)) / R
    w_right = (V + (omega * L / 2)) / R
    return (w_left, w_right)


def map_between_ranges(value, from_min_max, to_min_max):
    return (value - from_min_max[0]) * (to_min_max[1] - to_min_max[0]) / (from_min_max[1] - from_min_max[0]) + to_min_max[0]


def compute_K(fovy: float, width: int, height: int) -> np.ndarray:
    f = 0.5 * height / math.tan(fovy * math.pi / 360)
    return np.array(((f, 0, width / 2), (0, f, height / 2), (0, 0, 1)))


def limit_depth_distance(depth_image_meters, max_depth: float):
    """Values strictly greater than max_depth become 0 (works for numpy arrays and torch tensors)."""
    try:
        import torch

        if isinstance(depth_image_meters, torch.Tensor):
            return torch.where(depth_image_meters > max_depth, torch.zeros_like(depth_image_meters), depth_image_meters)
    except ImportError:  # pragma: no cover
        pass
    return np.where(depth_image_meters > max_depth, 0, depth_image_meters)


def to_real_gripper_range(pos):
    """stretch_mujoco/mujoco_server.py:517-525"""
    return map_between_ranges(pos, config.robot_settings["sim_gripper_min_max"], config.robot_settings["gripper_min_max"])


def to_sim_gripper_range(pos):
    """stretch_mujoco/mujoco_server.py:580-588"""
    return map_between_ranges(pos, config.robot_settings["gripper_min_max"], config.robot_settings["sim_gripper_min_max"])
