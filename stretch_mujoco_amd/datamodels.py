"""Batched counterparts of the reference dataclasses (stretch_mujoco/datamodels/*.py).

Field names and units are the reference's; every scalar becomes a tensor with a leading batch dimension [B].
Tensors returned by `pull_*` are fresh copies (the reference returns fresh pickled copies as well).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Optional


@dataclass
class PositionVelocity:  # status_stretch_joints.py:5-12
    pos: Any
    vel: Any


@dataclass
class BaseStatus:  # status_stretch_joints.py:14-24
    x: Any
    y: Any
    theta: Any
    x_vel: Any
    theta_vel: Any


@dataclass
class StatusStretchJoints:  # status_stretch_joints.py:26-74
    time: Any
    fps: float
    sim_to_real_time_ratio_msg: str
    base: BaseStatus
    lift: PositionVelocity
    arm: PositionVelocity
    head_pan: PositionVelocity
    head_tilt: PositionVelocity
    wrist_yaw: PositionVelocity
    wrist_pitch: PositionVelocity
    wrist_roll: PositionVelocity
    gripper: PositionVelocity

    def __getitem__(self, name: str):
        """Backward compatibility: square-bracket access (status_stretch_joints.py:41-43)."""
        return getattr(self, name)


@dataclass
class StatusStretchSensors:  # status_stretch_sensors.py:10-77
    time: Any
    fps: float
    base_gyro: Optional[Any] = None
    base_imu: Optional[Any] = None  # the accelerometer lives in the field named base_imu (status_stretch_sensors.py:53-55)
    lidar: Optional[Any] = None

    def get_data(self, sensor):
        from .enums import StretchSensors

        data = {StretchSensors.base_gyro: self.base_gyro, StretchSensors.base_accel: self.base_imu,
                StretchSensors.base_lidar: self.lidar}[sensor]
        if data is None:
            raise ValueError(f"Tried to get {sensor} data, but it is empty.")
        return data


@dataclass
class StatusStretchCameras:  # status_stretch_camera.py:10-125 (depth only on this path)
    time: Any
    fps: float
    cam_d405_depth: Optional[Any] = None
    cam_d435i_depth: Optional[Any] = None
    cam_d405_K: Optional[Any] = None
    cam_d435i_K: Optional[Any] = None
