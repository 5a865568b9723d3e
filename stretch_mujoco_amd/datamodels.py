"""Batched counterparts of the reference dataclasses (stretch_mujoco/datamodels/*.py).

Field names and units are the reference's; every scalar becomes a tensor with a leading batch dimension [B].
Tensors returned by `pull_*` are fresh copies (the reference returns fresh pickled copies as well).
"""
from __future__ import annotations

from dataclasses import dataclass
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
    """Batched: every image is a torch tensor [B, H, W] (fp32 metres), K is the 3x3 intrinsic matrix shared by all envs.
    The depth tensors are views of simulator-owned buffers that the next pull_camera_data() overwrites: .clone() to keep."""
    time: Any
    fps: float
    cam_d405_rgb: Optional[Any] = None
    cam_d405_depth: Optional[Any] = None
    cam_d405_K: Optional[Any] = None
    cam_d435i_rgb: Optional[Any] = None
    cam_d435i_depth: Optional[Any] = None
    cam_d435i_K: Optional[Any] = None
    cam_nav_rgb: Optional[Any] = None

    def get_camera_data(self, camera, *, auto_rotate: bool = True, auto_correct_rgb: bool = True, **_ignored):
        """status_stretch_camera.py:48-86: the d435i frames come out of the (physically rotated) optical frame and are
        turned upright with rot90(-1) when auto_rotate is set, the nav camera's with rot90(+1); colour images [..., H, W, 3]
        come back in BGR channel order when auto_correct_rgb is set (the reference's cv2.COLOR_RGB2BGR); ValueError when the
        image is empty."""
        from .enums import StretchCameras
        import torch

        data = None
        turn = 0
        if camera == StretchCameras.cam_d405_depth and self.cam_d405_depth is not None:
            data = self.cam_d405_depth
        elif camera == StretchCameras.cam_d435i_depth and self.cam_d435i_depth is not None:
            data = self.cam_d435i_depth
            data = torch.rot90(data, -1, dims=(-2, -1)) if auto_rotate else data
        elif camera in (StretchCameras.cam_d405_rgb, StretchCameras.cam_d435i_rgb, StretchCameras.cam_nav_rgb) and getattr(self, camera.name) is not None:
            data = getattr(self, camera.name)
            turn = {StretchCameras.cam_d405_rgb: 0, StretchCameras.cam_d435i_rgb: -1, StretchCameras.cam_nav_rgb: 1}[camera]
            if auto_rotate and turn:
                data = torch.rot90(data, turn, dims=(-3, -2))
            if auto_correct_rgb:
                data = data.flip(-1)
        if data is None:
            raise ValueError(f"Tried to get {camera} data, but it is empty or not implemented.")
        return data

    def get_all(self, *, auto_rotate: bool = True, **kw) -> dict:
        from .enums import StretchCameras

        data = {}
        for camera in StretchCameras.all():
            try:
                data[camera] = self.get_camera_data(camera, auto_rotate=auto_rotate, **kw)
            except ValueError:
                ...
        return data

    def set_camera_data(self, camera, data):
        from .enums import StretchCameras

        if camera not in list(StretchCameras):
            raise NotImplementedError(f"Camera {camera} is not implemented.")
        setattr(self, camera.name, data)

    @staticmethod
    def default():
        return StatusStretchCameras(time=0, fps=0)
