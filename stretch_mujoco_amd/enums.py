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
        """MJCF joints driven by this actuator (actuators.py:27-61)."""
        names = [j for j, a in _JOINT_TO_ACTUATOR.items() if a == self.name and j.startswith("joint_") and "*" not in j]
        names += [f"{j[:-1]}_open" for j, a in _JOINT_PREFIX_TO_ACTUATOR if a == self.name and "finger" in j]
        if self is Actuators.arm:
            names = [f"joint_arm_l{k}" for k in range(4)]
        if not names:
            raise NotImplementedError(f"{self.name} drives no MJCF joint")
        return names

    @staticmethod
    @lru_cache(maxsize=None)
    def get_actuator_by_joint_names_in_mjcf(joint_name: str) -> "Actuators":
        """Joint (or tendon / alias) name -> actuator, with the precedence of actuators.py:63-120: exact names first in table
        order, where the two substring rules (arm segments, finger joints) sit at their reference positions."""
        for rule, actuator in _MATCH_ORDER:
            if (rule.endswith("*") and rule[:-1] in joint_name) or rule == joint_name:
                return Actuators[actuator]
        raise NotImplementedError(f"no actuator is mapped to MJCF joint '{joint_name}'")

    # status accessors (actuators.py:124-182): joint actuators read status.<name>.pos / .vel, the two relative base moves
    # read the (x, y, theta) triple of status.base, everything else (wheels, finger pseudo-actuators) has no status entry
    def _read(self, status, want_base: bool, field: str):
        what = "position" if field == "pos" else "velocity"
        if (self.name in _BASE_MOVES) != want_base:   # wrong accessor family: a plain Exception, as in the reference
            right = f"get_{what}()" if want_base else f"get_{what}_relative()"
            raise Exception(f"{self.name}: read it with {right}")
        if want_base:
            sfx = "" if field == "pos" else "_vel"
            return tuple(getattr(status.base, k + sfx) for k in ("x", "y", "theta"))
        if self.name not in _STATUS_JOINTS:
            raise NotImplementedError(f"{self.name} has no {what} entry in the status")
        return getattr(getattr(status, self.name), field)

    def get_position(self, status):
        return self._read(status, False, "pos")

    def get_position_relative(self, status):
        return self._read(status, True, "pos")

    def get_velocity(self, status):
        return self._read(status, False, "vel")

    def get_velocity_relative(self, status):
        return self._read(status, True, "vel")


# joint / tendon / alias name -> actuator.  A trailing "*" marks a substring rule.  ORDER is part of the contract
# (actuators.py:63-120 tests the names top to bottom).
_MATCH_ORDER = (
    ("joint_left_wheel", "left_wheel_vel"), ("joint_right_wheel", "right_wheel_vel"),
    ("translate_mobile_base", "base_translate"), ("position", "base_translate"), ("rotate_mobile_base", "base_rotate"),
    ("joint_lift", "lift"), ("joint_arm*", "arm"),
    ("joint_wrist_yaw", "wrist_yaw"), ("joint_wrist_pitch", "wrist_pitch"), ("joint_wrist_roll", "wrist_roll"),
    ("joint_gripper_slide", "gripper"), ("gripper_aperture", "gripper"),
    ("joint_gripper_finger_left*", "gripper_left_finger"), ("joint_gripper_finger_right*", "gripper_right_finger"),
    ("joint_head_pan", "head_pan"), ("joint_head_tilt", "head_tilt"),
)
_JOINT_TO_ACTUATOR = {rule: act for rule, act in _MATCH_ORDER if not rule.endswith("*")}
_JOINT_PREFIX_TO_ACTUATOR = tuple((rule, act) for rule, act in _MATCH_ORDER if rule.endswith("*"))
_STATUS_JOINTS = ("arm", "gripper", "head_pan", "head_tilt", "lift", "wrist_pitch", "wrist_roll", "wrist_yaw")
_BASE_MOVES = ("base_rotate", "base_translate")


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
        """Sensor names base_lidar<i>, zero-padded to the width of `resolution` (stretch_sensors.py:33-40)."""
        width = len(f"{resolution}")
        return ["%s%0*d" % (StretchSensors.base_lidar.name, width, i) for i in range(resolution)]


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
