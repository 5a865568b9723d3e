"""StretchBatchSimulator -- the reference's StretchMujocoSimulator API with a leading batch dimension.

Reference: stretch_mujoco/stretch_mujoco_simulator.py:34-534 (client) + stretch_mujoco/mujoco_server.py (server).
The server process, proxies, locks and the realtime sleep (mujoco_server.py:381-384) have no place in a batched
throughput simulator: one Python object owns PyTorch-ROCm tensors (batch-major, [dim, B]) and drives the HIP
library through the ctypes C-ABI (lib.py, include/smj.h).  New, without a reference counterpart: `step(n)`,
`reset(env_ids)`.

Ordering contract kept from `_ctrl_callback` (mujoco_server.py:450-463): commands issued between steps are
folded into ctrl before the next physics step; status is the post-step readout.
"""
from __future__ import annotations

import ctypes
import json
import os
from typing import Optional, Sequence

import numpy as np
import torch

from . import lib as _lib
from . import model_blob
from .datamodels import StatusStretchCameras, StatusStretchJoints, StatusStretchSensors
from .enums import Actuators, StretchCameras, StretchSensors
from .glue import Glue

_MODELS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "models")


def _require_connection(fn):
    def wrapper(self, *a, **k):
        if not self.is_running():
            raise ConnectionError("The Stretch Mujoco Simulator is not running. Use the start() method to start it.")
        return fn(self, *a, **k)

    wrapper.__name__ = fn.__name__
    wrapper.__doc__ = fn.__doc__
    return wrapper


class StretchBatchSimulator:
    def __init__(self, num_envs: int = 1, device: str = "cuda:0", scene: str = "stretch_empty",
                 model_blob_bytes: Optional[bytes] = None, sensors_to_use: Sequence[StretchSensors] = (),
                 cameras_to_use: Sequence[StretchCameras] = (), start_translation=None, start_rotation_quat=None,
                 debug: bool = False, solver: str = "newton"):
        self.num_envs = int(num_envs)
        self.device = torch.device(device)
        if model_blob_bytes is None:
            if scene.endswith(".xml"):
                # any MJCF scene that includes stretch.xml (the reference's `scene_xml_path`, stretch_mujoco_simulator.py:47-55):
                # compiled here with the build's own MJCF compiler; needs the mesh assets next to the XML
                from . import mjcf_compiler, model_fuse

                # (a scene beyond the dense builds' 64 dofs / 32 bodies / 128 collision geoms -- a kitchen -- is prepared for the satellite builds)
                model_blob_bytes = model_blob.dumps(model_fuse.prepare_for_kernels(mjcf_compiler.compile_file(scene), satellites="auto"))
            else:
                with open(os.path.join(_MODELS, scene + ".smjb"), "rb") as f:
                    model_blob_bytes = f.read()
        self._blob = model_blob_bytes
        self.model = model_blob.loads(model_blob_bytes)
        self.names = json.loads(model_blob.get_str(self.model, "names_json"))
        self._sensors = list(sensors_to_use)
        self._cameras = list(cameras_to_use)
        # depth cameras: metric depth (smj_render_depth).  RGB cameras: an unlit-albedo STAND-IN (smj_render_rgb: geom / material
        # rgba of the first geom each pixel ray meets, uint8 [B, H, W, 3]) -- MuJoCo's lighting, textures and shadows are not on
        # this path (scope)
        self._start_translation = start_translation
        self._start_rotation_quat = start_rotation_quat
        self._debug = debug
        if solver not in ("pgs", "newton"):
            raise ValueError("solver must be 'pgs' (north_star) or 'newton' (the reference model's default)")
        self.solver = solver
        self._ctx = None
        self._L = None
        self.timestep = float(self.model["opt_timestep"][0])

    # ------------------------------------------------------------------ lifecycle
    def start(self, headless: bool = True, home: bool = True, **_ignored) -> None:
        """Allocate, reset and (like the reference, stretch_mujoco_simulator.py:136) send the robot home."""
        if self._ctx is not None:
            return
        if self.device.type != "cuda":
            raise _lib.SmjError("the physics path runs on the ROCm device only; there is no CPU fallback")
        L = self._L = _lib.load()
        ctx = ctypes.c_void_p()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        rc = L.smj_create(self._blob, len(self._blob), self.num_envs, dev_index, ctypes.byref(ctx))
        if rc != 0:
            msg = L.smj_last_error(ctx).decode() if ctx else "smj_create failed"
            if ctx:
                L.smj_destroy(ctx)
            raise _lib.SmjError(f"smj_create failed ({rc}): {msg}")
        self._ctx = ctx
        dims = (ctypes.c_int * 16)()
        _lib.check(L, ctx, L.smj_dims(ctx, dims), "smj_dims")
        D = _lib.DIM
        self.nq, self.nv, self.nu = dims[D["NQ"]], dims[D["NV"]], dims[D["NU"]]
        self.nlidar = dims[D["NLIDAR"]]
        # capacities of the kernel variant smj_create chose for this model (standard: 32 dofs / 80 rows / 16 contacts; big: 64 / 160 / 48)
        self.nv_max, self.nefc_max, self.ncon_max = dims[D["NV_MAX"]], dims[D["NEFC_MAX"]], dims[D["NCON_MAX"]]
        self.nsat_max = dims[D["NSAT_MAX"]]   # > 0: the model runs on a satellite build of the step kernel (csrc/smj_sat.h)
        self.debug_layout = _lib.debug_layout(self.nv_max, self.ncon_max, self.nsat_max)
        B, f = self.num_envs, dict(dtype=torch.float32, device=self.device)
        i32 = dict(dtype=torch.int32, device=self.device)
        self.qpos = torch.zeros(self.nq, B, **f); self.qvel = torch.zeros(self.nv, B, **f)
        nu1 = max(self.nu, 1)   # (a model without actuators -- a scene of objects only -- still binds every slot: zero-row views of real storage)
        store = {k: torch.zeros(nu1, B, **f) for k in ("ctrl", "actlen", "actvel")}
        self.ctrl = store["ctrl"][: self.nu]; self.qacc_warmstart = torch.zeros(self.nv, B, **f)
        self.nstep = torch.zeros(B, **i32)
        self.actuator_length = store["actlen"][: self.nu]; self.actuator_velocity = store["actvel"][: self.nu]
        self._store = store
        self.base_pose = torch.zeros(3, B, **f)
        self.gyro = torch.zeros(3, B, **f); self.accel = torch.zeros(3, B, **f)
        self.lidar = torch.zeros(max(self.nlidar, 1), B, **f)
        self.info = torch.zeros(4, B, **i32)
        S = _lib.SLOT
        binds = [("QPOS", self.qpos), ("QVEL", self.qvel), ("CTRL", store["ctrl"]), ("WARMSTART", self.qacc_warmstart),
                 ("NSTEP", self.nstep), ("ACT_LENGTH", store["actlen"]), ("ACT_VELOCITY", store["actvel"]),
                 ("BASE_POSE", self.base_pose), ("GYRO", self.gyro), ("ACCEL", self.accel), ("LIDAR", self.lidar),
                 ("INFO", self.info)]
        if self._debug:
            self.debug = torch.zeros(dims[D["DEBUG_FLOATS"]], B, **f)
            self.prof = torch.zeros(48, B, **f)
            binds.append(("DEBUG", self.debug))
            binds.append(("PROF", self.prof))
        for name, t in binds:
            _lib.check(L, ctx, L.smj_bind(ctx, S[name], ctypes.c_void_p(t.data_ptr()), B), f"smj_bind({name})")
        key_ctrl = torch.tensor(np.asarray(self.model["key_ctrl"], np.float32)[:, : self.nu])
        self.glue = Glue(B, self.nu, key_ctrl, self.names["key"], self.device)
        # the relative base moves (BaseController) advance inside the step kernel: its per-env state is the glue's tensor
        _lib.check(L, ctx, L.smj_bind(ctx, S["BASECTL"], ctypes.c_void_p(self.glue.bctl.data_ptr()), B), "smj_bind(BASECTL)")
        self.launches = 0   # smj_step launches so far (tests: a base move in flight must not multiply them)
        self.set_option("solver", {"pgs": 0, "newton": 2}[self.solver])
        self._read_flags = 0
        if StretchSensors.base_gyro in self._sensors or StretchSensors.base_accel in self._sensors:
            self._read_flags |= _lib.READ_IMU
        if StretchSensors.base_lidar in self._sensors:
            self._read_flags |= _lib.READ_LIDAR
        self._depth = {}
        # body poses of the last step: input of the depth renderer and of get_link_pose (240 floats per env, always on)
        self.xpose = torch.zeros(dims[D["NBODY"]] * 12, B, **f)
        _lib.check(L, ctx, L.smj_bind(ctx, S["XPOSE"], ctypes.c_void_p(self.xpose.data_ptr()), B), "smj_bind(XPOSE)")
        self._read_flags |= _lib.READ_POSES
        if self._cameras:
            if dims[D["NCAM"]] == 0:
                raise _lib.SmjError("the model blob carries no render tables: depth cameras are unavailable for this scene")
            for cam in self._cameras:
                st = cam.initial_camera_settings
                if cam.is_depth:
                    self._depth[cam] = torch.zeros(B, st.height, st.width, **f)
                else:
                    self._depth[cam] = torch.zeros(B, st.height, st.width, 3, dtype=torch.uint8, device=self.device)
        self.reset()
        if home:
            self.home()

    def stop(self) -> None:
        if self._ctx is not None:
            torch.cuda.synchronize(self.device)
            self._L.smj_destroy(self._ctx)
            self._ctx = None

    def is_running(self) -> bool:
        return self._ctx is not None

    def __del__(self):
        try:
            self.stop()
        except Exception:
            pass

    def set_option(self, name: str, value: float) -> None:
        """mjOption-style knobs: iterations, tolerance, warmstart, pgs_fixed_iter, max_contacts_per_pair, solver."""
        _lib.check(self._L, self._ctx, self._L.smj_set_option(self._ctx, name.encode(), float(value)), f"smj_set_option({name})")

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ------------------------------------------------------------------ new: reset / step
    @_require_connection
    def reset(self, env_ids=None) -> None:
        """mj_resetData for the selected envs, then the start pose (change_start_pose, mujoco_server.py:206-229)."""
        if env_ids is None:
            mask_ptr = None
            ids = slice(None)
        else:
            ids = torch.as_tensor(env_ids, device=self.device, dtype=torch.long)
            mask = torch.zeros(self.num_envs, dtype=torch.uint8, device=self.device)
            mask[ids] = 1
            mask_ptr = ctypes.c_void_p(mask.data_ptr())
        _lib.check(self._L, self._ctx, self._L.smj_reset(self._ctx, mask_ptr, self._stream()), "smj_reset")
        if self._start_translation is not None:  # [3] for every env or [B,3]
            t = torch.as_tensor(self._start_translation, dtype=torch.float32, device=self.device).reshape(-1, 3)
            self.qpos[0:3, ids] = t.expand(self.num_envs, 3).t()[:, ids]
        if self._start_rotation_quat is not None:  # wxyz, [4] or [B,4]
            q = torch.as_tensor(self._start_rotation_quat, dtype=torch.float32, device=self.device).reshape(-1, 4)
            self.qpos[3:7, ids] = q.expand(self.num_envs, 4).t()[:, ids]
        self.glue.reset(env_ids)
        # readout of the reset state (one zero-length update is not available: refresh by a cheap kinematic fill)
        self.actuator_length[:, ids] = 0
        self.actuator_velocity[:, ids] = 0
        self.base_pose[0:2, ids] = self.qpos[0:2, ids]
        qw, qx, qy, qz = self.qpos[3, ids], self.qpos[4, ids], self.qpos[5, ids], self.qpos[6, ids]
        self.base_pose[2, ids] = torch.atan2(2 * (qx * qy + qw * qz), qw * qw + qx * qx - qy * qy - qz * qz)

    @_require_connection
    def step(self, n: int = 1) -> None:
        """Advance every env by n physics steps (n x `_physics_step`, mujoco_server.py:371-384, without the sleep)."""
        n = int(n)
        if n <= 0:
            return
        self._push_command()
        _lib.check(self._L, self._ctx, self._L.smj_step(self._ctx, n, self._read_flags, self._stream()), "smj_step")
        self.launches += 1

    def _push_command(self) -> None:
        """`_ctrl_callback` -> push_command (mujoco_server.py:450-463, :527-578): commands issued since the last step are folded
        into ctrl / the base-controller state on the device (masked ops, no host synchronisation); when one of them concerns
        the base (or a keyframe overwrote the wheel ctrl) one BaseController.update() runs as a HIP tick -- every later
        update happens inside the step kernel, after each physics step."""
        if self.glue.push_command(self.ctrl, self.actuator_length, self.base_pose, tick_base=False):
            _lib.check(self._L, self._ctx, self._L.smj_base_controller_tick(self._ctx, self._stream()), "smj_base_controller_tick")

    # ------------------------------------------------------------------ reference API
    @_require_connection
    def home(self, env_ids=None, settle: bool = True) -> None:
        """Keyframe 'home' then wait until the lift stops (stretch_mujoco_simulator.py:213-221)."""
        self.glue.set_keyframe("home", env_ids)
        if settle:
            self.wait_while_is_moving(Actuators.lift)

    @_require_connection
    def stow(self, env_ids=None, settle: bool = True) -> None:
        """Keyframe 'stow' then wait on wrist_pitch (stretch_mujoco_simulator.py:224-233)."""
        self.glue.set_keyframe("stow", env_ids)
        if settle:
            self.wait_while_is_moving(Actuators.wrist_pitch)

    @_require_connection
    def move_to(self, actuator, pos, env_ids=None) -> None:
        self.glue.move_to(actuator, pos, env_ids)

    @_require_connection
    def move_by(self, actuator, pos, env_ids=None) -> None:
        self.glue.move_by(actuator, pos, env_ids)

    @_require_connection
    def set_base_velocity(self, v_linear, omega, env_ids=None) -> None:
        self.glue.set_base_velocity(v_linear, omega, env_ids)

    @_require_connection
    def pull_status(self) -> StatusStretchJoints:
        time = self.nstep.to(torch.float64) * self.timestep
        return Glue.pull_status(time, self.actuator_length, self.actuator_velocity, self.base_pose)

    @_require_connection
    def pull_sensor_data(self) -> StatusStretchSensors:
        """gyro [B,3], accel (field base_imu) [B,3], lidar [B,360]; values of the last physics step of the last launch.
        The reference refreshes these at 15 Hz wall clock (mujoco_server.py:271); here the cadence is the caller's step(n)."""
        out = StatusStretchSensors(time=self.nstep.to(torch.float64) * self.timestep, fps=0.0)
        if self._read_flags & _lib.READ_IMU:
            out.base_gyro = self.gyro.t().clone()
            out.base_imu = self.accel.t().clone()
        if self._read_flags & _lib.READ_LIDAR:
            out.lidar = self.lidar[: self.nlidar].t().clone()
        return out

    @_require_connection
    def pull_camera_data(self) -> StatusStretchCameras:
        """Depth images [B, H, W] (fp32 metres) and RGB stand-in images [B, H, W, 3] (uint8 unlit albedo, see __init__) of the
        cameras in cameras_to_use, rendered from the body poses of the last physics step
        (what Renderer.update_scene sees, mujoco_server_camera_manager.py:127-143), limited like
        StretchCameras.post_processing_callback; K as get_camera_params (:168-183: fovy with the SENSOR resolution).
        The reference renders at 30 Hz wall clock (mujoco_server.py:272); here the cadence is the caller's.
        The image tensors are simulator-owned and overwritten by the next call."""
        from .utils import compute_K

        out = StatusStretchCameras(time=self.nstep.to(torch.float64) * self.timestep, fps=0.0)
        names = self.names["camera"]
        for cam in self._cameras:
            st = cam.initial_camera_settings
            img = self._depth[cam]
            if cam.is_depth:
                rc = self._L.smj_render_depth(self._ctx, names.index(cam.camera_name_in_mjcf), st.width, st.height,
                                              float(st.field_of_view_vertical_in_degrees), cam.depth_limit,
                                              ctypes.c_void_p(img.data_ptr()), self._stream())
                _lib.check(self._L, self._ctx, rc, "smj_render_depth")
            else:
                rc = self._L.smj_render_rgb(self._ctx, names.index(cam.camera_name_in_mjcf), st.width, st.height,
                                            float(st.field_of_view_vertical_in_degrees), ctypes.c_void_p(img.data_ptr()), None,
                                            self._stream())
                _lib.check(self._L, self._ctx, rc, "smj_render_rgb")
            out.set_camera_data(cam, img)
        for attr, cam in (("cam_d405_K", StretchCameras.cam_d405_rgb), ("cam_d435i_K", StretchCameras.cam_d435i_rgb)):
            st = cam.initial_camera_settings
            setattr(out, attr, compute_K(st.field_of_view_vertical_in_degrees, st.sensor_resolution[0], st.sensor_resolution[1]))
        return out

    @_require_connection
    def pull_joint_limits(self) -> dict:
        """{Actuators: (lo, hi)} from jnt_range, later joints of one actuator overwrite earlier (mujoco_server.py:281-291)."""
        out = {}
        for j, name in enumerate(self.names["joint"]):
            try:
                act = Actuators.get_actuator_by_joint_names_in_mjcf(name)
            except NotImplementedError:
                continue
            r = self.model["jnt_range"][j]
            out[act] = (float(r[0]), float(r[1]))
        return out

    @_require_connection
    def get_base_pose(self):
        s = self.pull_status()
        return (s.base.x, s.base.y, s.base.theta)

    # ------------------------------------------------------------------ wait helpers as batched predicates
    @_require_connection
    def get_link_pose(self, link_name: str, simulated: bool = False) -> torch.Tensor:
        """World pose [B, 4, 4] of a link (a body name of stretch.xml, e.g. "link_grasp_center").

        Default = the reference's semantics (stretch_mujoco_simulator.py:468-486): forward kinematics of the link relative to
        `base_link` at the STATUS joint positions -- lift, arm (a quarter of the extension on each of the four telescope
        joints, utils.py:209-212), wrist yaw / pitch / roll, head pan / tilt; every other joint (gripper, fingers) at zero --
        placed at the planar base pose Rz(theta), (x, y, 0).  The reference evaluates its URDF for this; the URDF is not in
        the checkout, the same kinematic chain is read from the compiled MJCF model (stretch.xml carries the URDF's link
        names and frames).  simulated=True returns the simulated body pose of the last physics step instead, base roll /
        pitch / height, finger and rubber-tip joints included."""
        names = self.names["body"]
        if link_name not in names:
            raise KeyError(link_name)
        i = names.index(link_name)
        fb = int(self.model["link_fused"][i])
        rp = torch.tensor(np.asarray(self.model["link_relpos"][i], np.float32), device=self.device)
        Rl = self._quat_mat(self.model["link_relquat"][i], self.device)
        B = self.num_envs
        T = torch.zeros(B, 4, 4, dtype=torch.float32, device=self.device)
        T[:, 3, 3] = 1.0
        if simulated:
            P = self.xpose[12 * fb: 12 * fb + 3].t()                       # [B, 3]
            Rb = self.xpose[12 * fb + 3: 12 * fb + 12].t().reshape(-1, 3, 3)
            T[:, :3, :3] = Rb @ Rl
            T[:, :3, 3] = P + (Rb @ rp)
            return T
        st = self.pull_status()
        q = {"joint_lift": st.lift.pos, "joint_wrist_yaw": st.wrist_yaw.pos, "joint_wrist_pitch": st.wrist_pitch.pos,
             "joint_wrist_roll": st.wrist_roll.pos, "joint_head_pan": st.head_pan.pos, "joint_head_tilt": st.head_tilt.pos}
        for k in range(4):
            q[f"joint_arm_l{k}"] = st.arm.pos / 4
        m = self.model
        base = self.names["body"].index("base_link")
        base = int(m["link_fused"][base])
        path, b = [], fb
        while b != base and b > 0:
            path.append(b)
            b = int(m["body_parentid"][b])
        if b != base:
            raise KeyError(f"{link_name} is not part of the robot")
        R = torch.eye(3, device=self.device).expand(B, 3, 3).clone()
        p = torch.zeros(B, 3, device=self.device)
        f32 = lambda a: torch.tensor(np.asarray(a, np.float32), device=self.device)
        for b in reversed(path):     # [MJ] mj_kinematics: body frame in the parent, then the body's joints about their anchors
            p = p + (R @ f32(m["body_pos"][b]))
            R = R @ self._quat_mat(m["body_quat"][b], self.device)
            for j in range(int(m["body_jntadr"][b]), int(m["body_jntadr"][b]) + int(m["body_jntnum"][b])):
                val = q.get(self.names["joint"][j])
                if val is None:
                    continue
                val = val.float() - float(m["qpos0"][int(m["jnt_qposadr"][j])])
                axis, jpos = f32(m["jnt_axis"][j]), f32(m["jnt_pos"][j])
                if int(m["jnt_type"][j]) == 2:       # slide
                    p = p + (R @ axis) * val.unsqueeze(1)
                else:                                 # hinge: rotate about the anchor
                    anchor = p + (R @ jpos)
                    R = R @ self._axis_angle_mat(axis, val)
                    p = anchor - (R @ jpos)
        x, y, th = st.base.x.float(), st.base.y.float(), st.base.theta.float()
        Rz = torch.zeros(B, 3, 3, device=self.device)
        Rz[:, 0, 0] = torch.cos(th); Rz[:, 0, 1] = -torch.sin(th); Rz[:, 1, 0] = torch.sin(th); Rz[:, 1, 1] = torch.cos(th); Rz[:, 2, 2] = 1.0
        Rw = Rz @ R
        pw = (Rz @ p.unsqueeze(2)).squeeze(2)
        pw[:, 0] += x; pw[:, 1] += y
        T[:, :3, :3] = Rw @ Rl
        T[:, :3, 3] = pw + (Rw @ rp)
        return T

    @staticmethod
    def _quat_mat(qt, device) -> torch.Tensor:
        w, x, y, z = [float(v) for v in np.asarray(qt)]
        return torch.tensor([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                             [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                             [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]], dtype=torch.float32, device=device)

    @staticmethod
    def _axis_angle_mat(axis: torch.Tensor, ang: torch.Tensor) -> torch.Tensor:
        """Rotation matrices [B, 3, 3] about a fixed axis by per-env angles (Rodrigues)."""
        a = (axis / axis.norm()).tolist()
        K = torch.tensor([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]], dtype=torch.float32, device=axis.device)
        s, c = torch.sin(ang).view(-1, 1, 1), torch.cos(ang).view(-1, 1, 1)
        return torch.eye(3, device=axis.device) + s * K + (1 - c) * (K @ K)

    @_require_connection
    def get_ee_pose(self) -> torch.Tensor:
        """stretch_mujoco_simulator.py:464-466"""
        return self.get_link_pose("link_grasp_center")

    @_require_connection
    def is_reached_set_position(self, actuator, position_tolerance: float = 0.05) -> torch.Tensor:
        """[B] bool: the joint is within `position_tolerance` of its last `move_to` target; envs whose command holds no
        move_to entry for the actuator count as reached (stretch_mujoco_simulator.py:235-265, which warns and returns
        True).  Base and wheel actuators are not supported, as in the reference."""
        if isinstance(actuator, str):
            actuator = Actuators[actuator]
        if actuator in (Actuators.base_rotate, Actuators.base_translate, Actuators.left_wheel_vel, Actuators.right_wheel_vel):
            raise NotImplementedError(f"Check joint reached is not supported for {actuator}.")
        from .enums import CTRL_INDEX

        i = CTRL_INDEX[actuator.name]
        cur = actuator.get_position(self.pull_status())
        target = self.glue.mt_val[i].to(cur.dtype)
        return (~self.glue.mt_has[i]) | ((cur - target).abs() <= position_tolerance)

    @_require_connection
    def wait_until_at_setpoint(self, actuator, timeout: float = 5.0, position_tolerance: float = 0.05) -> torch.Tensor:
        """Step until every env has reached the actuator's last move_to target or `timeout` SIM-seconds elapse; returns the
        [B] bool mask of envs that got there (stretch_mujoco_simulator.py:267-297: wall clock there, sim clock here)."""
        if isinstance(actuator, str):
            actuator = Actuators[actuator]
        budget = int(round(timeout / self.timestep))
        chunk = max(1, int(round(0.05 / self.timestep)))
        self.step(1)        # fold the pending command into ctrl before the first check
        done_steps = 1
        ok = self.is_reached_set_position(actuator, position_tolerance)
        while not bool(ok.all()) and done_steps < budget:
            self.step(chunk)
            done_steps += chunk
            ok = self.is_reached_set_position(actuator, position_tolerance)
        return ok

    @_require_connection
    def wait_while_is_moving(self, actuator, timeout: Optional[float] = 5.0, check_interval: float = 0.1,
                             position_tolerance: float = 0.0005) -> torch.Tensor:
        """Step until the actuator stops moving in every env (|dpos| <= tol over check_interval of SIM time) or
        `timeout` sim-seconds elapse.  Returns a [B] bool mask of envs that came to rest.
        Reference semantics: stretch_mujoco_simulator.py:299-358 (wall-clock there, sim-clock here)."""
        if isinstance(actuator, str):
            actuator = Actuators[actuator]
        chunk = max(1, int(round(check_interval / self.timestep)))
        budget = int(round((timeout if timeout is not None else 1e9) / self.timestep))

        def position():
            st = self.pull_status()
            if actuator in (Actuators.left_wheel_vel, Actuators.base_translate):
                return st.base.x
            if actuator == Actuators.right_wheel_vel:
                return st.base.y
            if actuator == Actuators.base_rotate:
                return st.base.theta
            return actuator.get_position(st)

        last = position()
        still = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)
        done_steps = 0
        while done_steps < budget:
            self.step(chunk)
            done_steps += chunk
            cur = position()
            still = (cur - last).abs() <= position_tolerance
            last = cur
            if bool(still.all()):
                break
        return still
