"""Enums of the API surface (names, values and validation as in the reference's stretch_mujoco/enums/)."""
from __future__ import annotations

from enum import Enum
from functools import lru_cache


class Actuators(Enum):
    """stretch_mujoco/enums/actuators.py:7-25"""

    arm = 0
    gripper = 1
    head_pan = 2
    head_tilt = 3
    lift = 4
    wrist_pitch = 5
    wrist_roll = 6
    wrist_yaw = 7
    base_rotate = 8
    base_translate = 9
    left_wheel_vel = 10
    right_wheel_vel = 11
    gripper_left_finger = 12
    gripper_right_finger = 13

    def get_joint_names_in_mjcf(self) -> list:
        table = {
            Actuators.left_wheel_vel: ["joint_left_wheel"], Actuators.right_wheel_vel: ["joint_right_wheel"],
            Actuators.lift: ["joint_lift"],
            Actuators.arm: ["joint_arm_l0", "joint_arm_l1", "joint_arm_l2", "joint_arm_l3"],
            Actuators.wrist_yaw: ["joint_wrist_yaw"], Actuators.wrist_pitch: ["joint_wrist_pitch"],
            Actuators.wrist_roll: ["joint_wrist_roll"], Actuators.gripper: ["joint_gripper_slide"],
            Actuators.gripper_left_finger: ["joint_gripper_finger_left_open"],
            Actuators.gripper_right_finger: ["joint_gripper_finger_right_open"],
            Actuators.head_pan: ["joint_head_pan"], Actuators.head_tilt: ["joint_head_tilt"],
        }
        if self not in table:
            raise NotImplementedError(f"Joint names for {self} are not defined.")
        return table[self]

    @staticmethod
    @lru_cache(maxsize=None)
    def get_actuator_by_joint_names_in_mjcf(joint_name: str) -> "Actuators":
        """stretch_mujoco/enums/actuators.py:63-120 (same matching order)."""
        if joint_name == "joint_left_wheel":
            return Actuators.left_wheel_vel
        if joint_name == "joint_right_wheel":
            return Actuators.right_wheel_vel
        if joint_name in ("translate_mobile_base", "position"):
            return Actuators.base_translate
        if joint_name == "rotate_mobile_base":
            return Actuators.base_rotate
        if joint_name == "joint_lift":
            return Actuators.lift
        if "joint_arm" in joint_name:
            return Actuators.arm
        if joint_name == "joint_wrist_yaw":
            return Actuators.wrist_yaw
        if joint_name == "joint_wrist_pitch":
            return Actuators.wrist_pitch
        if joint_name == "joint_wrist_roll":
            return Actuators.wrist_roll
        if joint_name in ("joint_gripper_slide", "gripper_aperture"):
            return Actuators.gripper
        if "joint_gripper_finger_left" in joint_name:
            return Actuators.gripper_left_finger
        if "joint_gripper_finger_right" in joint_name:
            return Actuators.gripper_right_finger
        if joint_name == "joint_head_pan":
            return Actuators.head_pan
        if joint_name == "joint_head_tilt":
            return Actuators.head_tilt
        raise NotImplementedError(f"Actuator for {joint_name} is not defined.")

    # status accessors (actuators.py:124-182)
    def _get_status_attribute(self, is_position: bool, status):
        attribute_name = "pos" if is_position else "vel"
        if self in (Actuators.arm, Actuators.gripper, Actuators.head_pan, Actuators.head_tilt, Actuators.lift,
                    Actuators.wrist_pitch, Actuators.wrist_roll, Actuators.wrist_yaw):
            return getattr(getattr(status, self.name), attribute_name)
        raise NotImplementedError(f"Get {'Position' if is_position else 'Velocity'} for {self.name} is not implemented.")

    def _get_base_status_attribute(self, is_position: bool, status):
        x = "x" if is_position else "x_vel"
        y = "y" if is_position else "y_vel"
        theta = "theta" if is_position else "theta_vel"
        if self in (Actuators.base_rotate, Actuators.base_translate):
            return (getattr(status.base, x), getattr(status.base, y), getattr(status.base, theta))
        raise NotImplementedError(f"Get {'Position' if is_position else 'Velocity'}  for {self.name} is not implemented.")

    def get_position(self, status):
        if self in (Actuators.base_rotate, Actuators.base_translate):
            raise Exception(f"Please use `get_position_relative()` for {self.name}")
        return self._get_status_attribute(True, status)

    def get_position_relative(self, status):
        if self not in (Actuators.base_rotate, Actuators.base_translate):
            raise Exception(f"Please use `get_position()` for {self.name}")
        return self._get_base_status_attribute(True, status)

    def get_velocity(self, status):
        if self in (Actuators.base_rotate, Actuators.base_translate):
            raise Exception(f"Please use `get_velocity_relative()` for {self.name}")
        return self._get_status_attribute(False, status)

    def get_velocity_relative(self, status):
        if self not in (Actuators.base_rotate, Actuators.base_translate):
            raise Exception(f"Please use `get_velocity()` for {self.name}")
        return self._get_base_status_attribute(False, status)


# ctrl index of each MJCF actuator (stretch.xml:525-534)
CTRL_INDEX = {"left_wheel_vel": 0, "right_wheel_vel": 1, "lift": 2, "arm": 3, "wrist_yaw": 4, "wrist_pitch": 5,
              "wrist_roll": 6, "gripper": 7, "head_pan": 8, "head_tilt": 9}


class StretchSensors(Enum):
    """stretch_mujoco/enums/stretch_sensors.py:8-40"""

    base_gyro = 0
    base_accel = 1
    base_lidar = 2

    @staticmethod
    def all() -> list:
        return [s for s in StretchSensors]

    @staticmethod
    def none() -> list:
        return []

    @staticmethod
    @lru_cache(maxsize=None)
    def lidar_names(resolution: int = 720):
        num_digits = len(str(resolution))
        return [f"{StretchSensors.base_lidar.name}{str(i).zfill(num_digits)}" for i in range(resolution)]


class StretchCameras(Enum):
    """stretch_mujoco/enums/stretch_cameras.py:10-21"""

    cam_d405_rgb = 0
    cam_d405_depth = 1
    cam_d435i_rgb = 2
    cam_d435i_depth = 3
    cam_nav_rgb = 4

    @staticmethod
    def all() -> list:
        return [c for c in StretchCameras]

    @staticmethod
    def none() -> list:
        return []

    @staticmethod
    def rgb() -> list:
        return [StretchCameras.cam_d405_rgb, StretchCameras.cam_d435i_rgb, StretchCameras.cam_nav_rgb]

    @staticmethod
    def depth() -> list:
        return [StretchCameras.cam_d405_depth, StretchCameras.cam_d435i_depth]

    @property
    def camera_name_in_mjcf(self) -> str:
        return {StretchCameras.cam_d405_rgb: "d405_rgb", StretchCameras.cam_d405_depth: "d405_depth",
                StretchCameras.cam_d435i_rgb: "d435i_camera_rgb", StretchCameras.cam_d435i_depth: "d435i_camera_depth",
                StretchCameras.cam_nav_rgb: "nav_camera_rgb"}[self]

    @property
    def is_depth(self) -> bool:
        return self in (StretchCameras.cam_d405_depth, StretchCameras.cam_d435i_depth)

    @property
    def depth_limit(self) -> float:
        """Metres beyond which post_processing_callback zeroes the depth (enums/stretch_cameras.py:87-102, config.py:8)."""
        from . import config

        if self == StretchCameras.cam_d405_depth:
            return float(config.depth_limits["d405"])
        if self == StretchCameras.cam_d435i_depth:
            return float(config.depth_limits["d435i"])
        raise NotImplementedError(f"Camera {self} has no depth limit")

    @property
    def initial_camera_settings(self) -> "CameraSettings":
        """enums/stretch_cameras.py:105-156: the depth cameras reuse the settings of their RGB twins."""
        if self in (StretchCameras.cam_d405_rgb, StretchCameras.cam_d405_depth):
            return CameraSettings(field_of_view_vertical_in_degrees=58, focal=(242.56, 242.34), width=480, height=270,
                                  sensor_resolution=(1280, 720))
        if self in (StretchCameras.cam_d435i_rgb, StretchCameras.cam_d435i_depth):
            return CameraSettings(field_of_view_vertical_in_degrees=42, focal=(304.24, 304.07), width=424, height=240,
                                  sensor_resolution=(1920, 1080))
        if self == StretchCameras.cam_nav_rgb:
            import math

            fovy = int(abs(math.degrees(2 * math.atan(math.tan(math.radians(70) / 2) * (1280 / 720)))))
            return CameraSettings(field_of_view_vertical_in_degrees=fovy, focal=(0.0, 0.0), width=800, height=600,
                                  sensor_resolution=(1280, 720))
        raise NotImplementedError(f"Camera {self} initial settings are not implemented")


class CameraSettings:
    """enums/stretch_cameras.py:184-206 (the fields the depth path uses)."""

    def __init__(self, field_of_view_vertical_in_degrees, focal, width, height, sensor_resolution=None):
        self.field_of_view_vertical_in_degrees = field_of_view_vertical_in_degrees
        self.focal = focal
        self.width = width
        self.height = height
        self.sensor_resolution = sensor_resolution

    @property
    def sensor_size(self):
        return None   # pixel sizes are commented out in the reference: cam_sensorsize stays 0, pure-fovy pinhole
