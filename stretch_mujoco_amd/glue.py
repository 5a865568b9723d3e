"""Batched command -> ctrl mapping, base move-by controller and status readout.

Restates, with a leading batch dimension, the Python that runs around `mj_step` in the reference:
  * StatusCommand merge rules              stretch_mujoco/datamodels/status_command.py:53-76
  * MujocoServer.push_command              stretch_mujoco/mujoco_server.py:527-578
  * BaseController                         stretch_mujoco/mujoco_server.py:93-176
  * MujocoServer.pull_status               stretch_mujoco/mujoco_server.py:465-515
All state lives in torch tensors on the simulator's device; nothing here touches the physics.
Quirks kept on purpose: no angle wrap in base rotate-by (:160-163); gripper velocity stays in sim units
(:499-501); base x_vel/theta_vel are computed from actuator_velocity = gear*qvel (the x3 wheel quirk).
Nothing in here reads a device value back to the host: commands are folded in with masks.
"""
from __future__ import annotations

import torch

from . import config, utils
from .datamodels import BaseStatus, PositionVelocity, StatusStretchJoints
from .enums import CTRL_INDEX, Actuators

_NO_ABS = (Actuators.left_wheel_vel, Actuators.right_wheel_vel, Actuators.base_rotate, Actuators.base_translate)
MODE_NONE, MODE_TRANSLATE, MODE_ROTATE, MODE_VELOCITY = 0, 1, 2, 3


class Glue:
    def __init__(self, num_envs: int, nu: int, key_ctrl: torch.Tensor, key_names, device, dtype=torch.float32):
        B = self.B = num_envs
        self.nu = nu
        self.device = device
        f = dict(dtype=dtype, device=device)
        b = dict(dtype=torch.bool, device=device)
        self.key_ctrl = key_ctrl.to(**f)          # [nkey, nu]
        self.key_names = list(key_names)
        self.mt_val = torch.zeros(nu, B, **f); self.mt_trig = torch.zeros(nu, B, **b)
        self.mt_has = torch.zeros(nu, B, **b)     # a move_to entry exists in the command (it outlives its trigger)
        self.mb_val = torch.zeros(nu, B, **f); self.mb_trig = torch.zeros(nu, B, **b)
        self.bt_val = torch.zeros(B, **f); self.bt_trig = torch.zeros(B, **b)
        self.br_val = torch.zeros(B, **f); self.br_trig = torch.zeros(B, **b)
        self.bv_v = torch.zeros(B, **f); self.bv_w = torch.zeros(B, **f); self.bv_trig = torch.zeros(B, **b)
        self.kf_id = torch.zeros(B, dtype=torch.long, device=device); self.kf_trig = torch.zeros(B, **b)
        # BaseController state (mujoco_server.py:95-103), one column per env: row 0 mode, rows 1-3 start pose, row 4 increment,
        # rows 5-6 (v, omega).  On the GPU this tensor is bound to SMJ_SLOT_BASECTL: the step kernel runs the controller's
        # update() after every physics step, the host only writes it when a command is pushed.
        self.bctl = torch.zeros(8, B, **f)
        # host-side hints: which command kinds were issued since the last push -- an idle push_command() launches nothing, and
        # nothing here ever reads a device value back (no synchronisation)
        self._h = dict(bt=False, br=False, mb=False, mt=False, bv=False, kf=False)

    # views kept for tests / introspection
    @property
    def bc_mode(self):
        return self.bctl[0].to(torch.int32)

    @property
    def bc_start(self):
        return self.bctl[1:4]

    # ---------------------------------------------------------------- client side (stretch_mujoco_simulator.py)
    def _ids(self, env_ids):
        if env_ids is None:
            return slice(None)
        return torch.as_tensor(env_ids, device=self.device, dtype=torch.long)

    @staticmethod
    def _actuator(actuator) -> Actuators:
        if isinstance(actuator, str):
            actuator = Actuators[actuator]  # KeyError for unknown names, as in the reference
        return actuator

    def move_to(self, actuator, pos, env_ids=None):
        actuator = self._actuator(actuator)
        if actuator in _NO_ABS:
            raise Exception(f"Cannot set an absolute position for a continuous joint {actuator.name}")
        i = CTRL_INDEX[actuator.name]  # KeyError for finger pseudo-actuators: no such MJCF actuator
        ids = self._ids(env_ids)
        self.mt_val[i, ids] = pos
        self.mt_trig[i, ids] = True
        self.mt_has[i, ids] = True
        self._h["mt"] = True
        self.mb_trig[i, ids] = False  # set_move_to pops move_by (status_command.py:53-57)

    def move_by(self, actuator, pos, env_ids=None):
        actuator = self._actuator(actuator)
        if actuator in (Actuators.left_wheel_vel, Actuators.right_wheel_vel):
            raise Exception(f"Cannot set an absolute position for a continuous joint {actuator.name}")
        ids = self._ids(env_ids)
        if actuator == Actuators.base_translate:
            self.bt_val[ids] = pos; self.bt_trig[ids] = True
            self._h["bt"] = True
            return
        if actuator == Actuators.base_rotate:
            self.br_val[ids] = pos; self.br_trig[ids] = True
            self._h["br"] = True
            return
        i = CTRL_INDEX[actuator.name]
        self.mb_val[i, ids] = pos
        self.mb_trig[i, ids] = True
        self._h["mb"] = True
        self.mt_trig[i, ids] = False  # set_move_by pops move_to (status_command.py:59-63)
        self.mt_has[i, ids] = False

    def set_base_velocity(self, v_linear, omega, env_ids=None):
        ids = self._ids(env_ids)
        self.bv_v[ids] = v_linear; self.bv_w[ids] = omega; self.bv_trig[ids] = True
        self._h["bv"] = True
        # pops every base / wheel move (status_command.py:65-76)
        self.bt_trig[ids] = False; self.br_trig[ids] = False

    def set_keyframe(self, name: str, env_ids=None, replace_command: bool = True):
        ids = self._ids(env_ids)
        if replace_command:  # home()/stow() install a brand-new StatusCommand (stretch_mujoco_simulator.py:217-231)
            self.mt_trig[:, ids] = False; self.mb_trig[:, ids] = False
            self.mt_has[:, ids] = False   # the new command holds no move_to entry: is_reached_set_position() is True again
            self.bt_trig[ids] = False; self.br_trig[ids] = False; self.bv_trig[ids] = False
        self.kf_id[ids] = self.key_names.index(name)
        self.kf_trig[ids] = True
        self._h["kf"] = True

    def reset(self, env_ids=None):
        ids = self._ids(env_ids)
        for t in (self.mt_trig, self.mb_trig, self.mt_has):
            t[:, ids] = False
        for t in (self.bt_trig, self.br_trig, self.bv_trig, self.kf_trig):
            t[ids] = False
        self.bctl[:, ids] = 0

    # ---------------------------------------------------------------- server side (mujoco_server.py:527-578)
    def push_command(self, ctrl: torch.Tensor, act_len: torch.Tensor, base_pose: torch.Tensor, tick_base=True) -> bool:
        """Fold pending commands into ctrl [nu,B] and the base-controller state in place.  act_len [nu,B], base_pose [3,B] are
        the status readout of the last step.  Masked tensor ops only: no value is read back from the device.
        tick_base: run BaseController.update() once at the end, as the reference does after every push (mujoco_server.py:576)
        -- True in the host-only replays; the simulator passes False and lets the HIP tick (smj_base_controller_tick) and the
        step kernel do it.  Returns whether a base tick is due (a base command or a keyframe was folded in)."""
        g = CTRL_INDEX["gripper"]
        h = self._h
        need_tick = h["bt"] or h["br"] or h["bv"] or h["kf"]
        # move_by: base first (push to the controller: last_command, start_pose), then joints
        for key, trig, val, mode in (("bt", self.bt_trig, self.bt_val, MODE_TRANSLATE), ("br", self.br_trig, self.br_val, MODE_ROTATE)):
            if h[key]:
                self.bctl[0] = torch.where(trig, torch.full_like(self.bctl[0], float(mode)), self.bctl[0])
                self.bctl[4] = torch.where(trig, val, self.bctl[4])
                self.bctl[1:4] = torch.where(trig.unsqueeze(0), base_pose.to(self.bctl.dtype), self.bctl[1:4])
                trig.zero_()
        if h["mb"]:
            target = act_len + self.mb_val
            target[g] = utils.to_sim_gripper_range(utils.to_real_gripper_range(act_len[g]) + self.mb_val[g])
            ctrl.copy_(torch.where(self.mb_trig, target.to(ctrl.dtype), ctrl))
            self.mb_trig.zero_()
        # move_to
        if h["mt"]:
            target = self.mt_val.clone()
            target[g] = utils.to_sim_gripper_range(self.mt_val[g])
            ctrl.copy_(torch.where(self.mt_trig, target.to(ctrl.dtype), ctrl))
            self.mt_trig.zero_()
        # set_base_velocity
        if h["bv"]:
            t = self.bv_trig
            self.bctl[0] = torch.where(t, torch.full_like(self.bctl[0], float(MODE_VELOCITY)), self.bctl[0])
            self.bctl[5] = torch.where(t, self.bv_v, self.bctl[5]); self.bctl[6] = torch.where(t, self.bv_w, self.bctl[6])
            self.bctl[1:4] = torch.where(t.unsqueeze(0), base_pose.to(self.bctl.dtype), self.bctl[1:4])
            t.zero_()
        # keyframe
        if h["kf"]:
            kc = self.key_ctrl[self.kf_id].t()  # [nu,B]
            ctrl.copy_(torch.where(self.kf_trig.unsqueeze(0), kc.to(ctrl.dtype), ctrl))
            self.kf_trig.zero_()
        for k in h:
            h[k] = False
        if tick_base:
            self.base_controller_update(ctrl, base_pose)
        return need_tick

    def base_controller_update(self, ctrl, pose):
        """BaseController.update() (mujoco_server.py:110-176) in torch -- the host twin of the HIP controller
        (csrc/smj_step_impl.h base_controller, csrc/smj_kernels.hip smj_base_tick_kernel), used by the CPU replays."""
        mode, inc = self.bctl[0], self.bctl[4]
        start = self.bctl[1:4]
        li, ri = CTRL_INDEX["left_wheel_vel"], CTRL_INDEX["right_wheel_vel"]
        one = torch.ones_like(inc)
        sign = torch.where(inc > 0, one, -one)
        # translate (mujoco_server.py:144-154)
        dist = torch.linalg.vector_norm(pose[:2].to(start.dtype) - start[:2], dim=0)
        t_on = mode == MODE_TRANSLATE
        t_done = t_on & ~(dist <= inc.abs())
        # rotate (mujoco_server.py:156-165; no angle wrap)
        r_on = mode == MODE_ROTATE
        r_done = r_on & ~((start[2] - pose[2].to(start.dtype)).abs() <= inc.abs())
        v_on = mode == MODE_VELOCITY
        zero = torch.zeros_like(inc)
        v = torch.where(t_on & ~t_done, config.base_motion["default_x_vel"] * sign, zero)
        v = torch.where(v_on, self.bctl[5], v)
        w = torch.where(r_on & ~r_done, config.base_motion["default_r_vel"] * sign, zero)
        w = torch.where(v_on, self.bctl[6], w)
        wl, wr = utils.diff_drive_inv_kinematics(v, w)
        active = mode != MODE_NONE
        ctrl[li] = torch.where(active, wl.to(ctrl.dtype), ctrl[li])
        ctrl[ri] = torch.where(active, wr.to(ctrl.dtype), ctrl[ri])
        self.bctl[0] = torch.where(t_done | r_done, torch.zeros_like(mode), mode)

    # ---------------------------------------------------------------- status (mujoco_server.py:465-515)
    @staticmethod
    def pull_status(time, act_len, act_vel, base_pose) -> StatusStretchJoints:
        def pv(name):
            i = CTRL_INDEX[name]
            return PositionVelocity(act_len[i].clone(), act_vel[i].clone())

        x_vel, theta_vel = utils.diff_drive_fwd_kinematics(act_vel[CTRL_INDEX["left_wheel_vel"]], act_vel[CTRL_INDEX["right_wheel_vel"]])
        grip = PositionVelocity(utils.to_real_gripper_range(act_len[CTRL_INDEX["gripper"]]), act_vel[CTRL_INDEX["gripper"]].clone())
        return StatusStretchJoints(
            time=time, fps=0.0, sim_to_real_time_ratio_msg="",
            base=BaseStatus(base_pose[0].clone(), base_pose[1].clone(), base_pose[2].clone(), x_vel, theta_vel),
            lift=pv("lift"), arm=pv("arm"), head_pan=pv("head_pan"), head_tilt=pv("head_tilt"), wrist_yaw=pv("wrist_yaw"),
            wrist_pitch=pv("wrist_pitch"), wrist_roll=pv("wrist_roll"), gripper=grip)
