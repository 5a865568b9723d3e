"""Generate tests/golden/glue_golden.json by IMPORTING the reference's Python glue (not copying it).

    python tools/gen_glue_golden.py [/root/reference]

Runs only where the reference is mounted.  Third-party modules the reference imports but this image lacks
(mujoco, cv2, urchin, stretch_urdf) are replaced by empty stub packages in a temp dir; the physics itself is
NOT exercised -- only the pure-Python command/status glue, driven against a fake MjData:
  utils.diff_drive_*/map_between_ranges/compute_K/limit_depth_distance, StretchSensors.lidar_names,
  StatusCommand merge rules, MujocoServer.push_command + BaseController, MujocoServer.pull_status,
  StretchMujocoSimulator.move_to/move_by validation.
The JSON holds inputs and expected outputs only (data, no source).
"""
import json
import os
import sys
import tempfile

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "glue_golden.json")
sys.dont_write_bytecode = True


def make_stubs(d):
    def pkg(name, body=""):
        p = os.path.join(d, *name.split("."))
        os.makedirs(p, exist_ok=True)
        with open(os.path.join(p, "__init__.py"), "w") as f:
            f.write(body)

    pkg("mujoco", "class MjModel: pass\nclass MjData: pass\nclass Renderer: pass\n")
    for sub in ("_functions", "_callbacks", "_render", "viewer"):
        pkg("mujoco." + sub)
    pkg("mujoco._enums", "class mjtGeom: pass\nclass mjtObj: pass\n")
    pkg("mujoco._structs", "class MjModel: pass\nclass MjData: pass\nclass MjvCamera: pass\nclass MjvOption: pass\nclass MjvScene: pass\n")
    pkg("mujoco.glfw", "class GLContext: pass\n")
    pkg("cv2", "COLORMAP_JET = 2\nROTATE_90_CLOCKWISE = 0\n")
    pkg("urchin", "class URDF:\n    @staticmethod\n    def load(*a, **k):\n        return None\n")
    pkg("stretch_urdf")


ACT_NAMES = ["left_wheel_vel", "right_wheel_vel", "lift", "arm", "wrist_yaw", "wrist_pitch", "wrist_roll", "gripper",
             "head_pan", "head_tilt"]
KEY_CTRL = {"home": [0, 0, 0.6, 0.1, 0, 0, 0, 0, 0, 0], "stow": [0, 0, 0.23, 0, 3.14, -0.4, 0, 0, 0, 0]}


class FakeActuator:
    def __init__(self, data, i):
        self._d, self._i = data, i

    @property
    def length(self):
        return np.array([self._d.length[self._i]])

    @property
    def velocity(self):
        return np.array([self._d.velocity[self._i]])

    @property
    def ctrl(self):
        return self._d._ctrl[self._i: self._i + 1]

    @ctrl.setter
    def ctrl(self, v):
        self._d._ctrl[self._i] = float(np.asarray(v).reshape(-1)[0])


class FakeBody:
    def __init__(self, d):
        self.xpos, self.xmat = d.xpos, d.xmat


class FakeData:
    def __init__(self):
        self._ctrl = np.zeros(10)
        self.length = np.zeros(10)
        self.velocity = np.zeros(10)
        self.xpos = np.zeros(3)
        self.xmat = np.eye(3).reshape(9)
        self.time = 1.0

    @property
    def ctrl(self):
        return self._ctrl

    @ctrl.setter
    def ctrl(self, v):
        self._ctrl[:] = np.asarray(v, float)

    def actuator(self, name):
        return FakeActuator(self, ACT_NAMES.index(name))  # ValueError for unknown names

    def body(self, name):
        assert name == "base_link"
        return FakeBody(self)


class FakeKey:
    def __init__(self, ctrl):
        self.ctrl = np.array(ctrl, float)


class FakeModel:
    def keyframe(self, name):
        return FakeKey(KEY_CTRL[name])


class FakeProxies:
    def __init__(self):
        self.command = None
        self.status = None

    def set_command(self, c):
        self.command = c

    def get_command(self):
        return self.command

    def set_status(self, s):
        self.status = s


def main():
    tmp = tempfile.mkdtemp(prefix="smj_stubs_")
    make_stubs(tmp)
    sys.path[:0] = [tmp, REF]
    import stretch_mujoco.utils as utils
    from stretch_mujoco.datamodels.status_command import CommandBaseVelocity, CommandKeyframe, CommandMove, StatusCommand
    from stretch_mujoco.enums.stretch_sensors import StretchSensors
    from stretch_mujoco.enums.stretch_cameras import CameraSettings
    from stretch_mujoco.mujoco_server import BaseController, MujocoServer
    from stretch_mujoco.stretch_mujoco_simulator import StretchMujocoSimulator
    import stretch_mujoco.config as config

    rng = np.random.default_rng(20260928)
    G = {"config": {"robot_settings": {k: list(v) if isinstance(v, tuple) else v for k, v in config.robot_settings.items()},
                    "depth_limits": config.depth_limits, "base_motion": config.base_motion}}

    # ---- pure functions
    inv = [[float(v), float(w)] + [float(x) for x in utils.diff_drive_inv_kinematics(v, w)]
           for v, w in np.concatenate([[[0.3, 0], [0, 1.0], [0.3, -0.1]], rng.uniform(-1, 1, (20, 2))])]
    fwd = [[float(a), float(b)] + [float(x) for x in utils.diff_drive_fwd_kinematics(a, b)]
           for a, b in np.concatenate([[[5.905511811023622, 5.905511811023622]], rng.uniform(-8, 8, (20, 2))])]
    sim_r, real_r = config.robot_settings["sim_gripper_min_max"], config.robot_settings["gripper_min_max"]
    mp = [[float(x), float(utils.map_between_ranges(x, sim_r, real_r)), float(utils.map_between_ranges(x, real_r, sim_r))]
          for x in np.concatenate([[0.0, 0.5], rng.uniform(-0.5, 0.6, 20)])]
    K = [[fovy, w, h, utils.compute_K(fovy, w, h).tolist()] for fovy, w, h in [(58, 1280, 720), (42, 1920, 1080), (102, 800, 600)]]
    depth_in = [0.5, 1.0, 1.0001, 9.0, 10.0, 10.5]
    G["pure"] = dict(inv=inv, fwd=fwd, map=mp, K=K, depth_in=depth_in,
                     depth_out_1=utils.limit_depth_distance(np.array(depth_in), 1).tolist(),
                     depth_out_10=utils.limit_depth_distance(np.array(depth_in), 10).tolist(),
                     lidar_names_360=[StretchSensors.lidar_names(360)[i] for i in (0, 1, 359)],
                     fov_v_from_h=float(CameraSettings.field_of_view_vertical_from_horizontal(70, 1280, 720)))

    # ---- server glue: scripted scenarios
    def new_server():
        s = object.__new__(MujocoServer)
        s.mjdata, s.mjmodel, s.data_proxies = FakeData(), FakeModel(), FakeProxies()
        s.base_controller = BaseController(s)
        return s

    def apply_client(cmd, op):
        kind = op[0]
        if kind == "move_to":
            cmd.set_move_to(CommandMove(actuator_name=op[1], pos=op[2], trigger=True))
        elif kind == "move_by":
            cmd.set_move_by(CommandMove(actuator_name=op[1], pos=op[2], trigger=True))
        elif kind == "base_velocity":
            cmd.set_base_velocity(CommandBaseVelocity(v_linear=op[1], omega=op[2], trigger=True))
        elif kind == "keyframe":
            return StatusCommand(keyframe=CommandKeyframe(name=op[1], trigger=True))
        return cmd

    pos_acts = ["lift", "arm", "wrist_yaw", "wrist_pitch", "wrist_roll", "gripper", "head_pan", "head_tilt"]
    scenarios = []

    def run_scenario(ticks):
        s = new_server()
        cmd = StatusCommand.default()
        rec = []
        for t in ticks:
            d = s.mjdata
            d.length[:] = t["length"]; d.velocity[:] = t["velocity"]
            d.xpos[:] = [t["pose"][0], t["pose"][1], 0.0]
            th = t["pose"][2]
            d.xmat[:] = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]]).reshape(9)
            for op in t["ops"]:
                cmd = apply_client(cmd, op)
            s.push_command(cmd)
            lc = s.base_controller.last_command
            mode = 0 if lc is None else (3 if isinstance(lc, CommandBaseVelocity) else (1 if lc.actuator_name == "base_translate" else 2))
            pend = dict(move_to=sorted(k for k, v in cmd.move_to.items()), move_by=sorted(k for k, v in cmd.move_by.items()))
            rec.append(dict(ctrl=[float(x) for x in d.ctrl], mode=mode, start=[float(x) for x in s.base_controller.start_pose],
                            keys=pend))
        return rec

    def rand_tick(ops, pose=None):
        return dict(length=rng.uniform(-0.5, 1.0, 10).tolist(), velocity=rng.uniform(-1, 1, 10).tolist(),
                    pose=(pose if pose is not None else rng.uniform(-1, 1, 3).tolist()), ops=ops)

    # S0: the SURVEY C.2 cases
    t0 = rand_tick([["keyframe", "home"]])
    scenarios.append([t0])
    t = rand_tick([["move_by", "lift", 0.1], ["move_to", "gripper", 0.5], ["base_velocity", 0.3, -0.1]])
    t["length"][2] = 0.59
    scenarios.append([t])
    scenarios.append([rand_tick([["move_by", "base_translate", 0.07]], [0, 0, 0]), rand_tick([], [0.03, 0, 0]),
                      rand_tick([], [0.08, 0, 0]), rand_tick([], [0.2, 0, 0])])
    scenarios.append([rand_tick([["move_by", "base_rotate", -0.5]], [0, 0, 0.1]), rand_tick([], [0, 0, -0.2]),
                      rand_tick([], [0, 0, -0.45]), rand_tick([], [0, 0, 0.3])])
    # merge rules
    scenarios.append([rand_tick([["move_to", "lift", 0.4], ["move_by", "lift", 0.05]])])
    scenarios.append([rand_tick([["move_by", "arm", 0.05], ["move_to", "arm", 0.2]])])
    scenarios.append([rand_tick([["move_by", "base_translate", 0.3], ["base_velocity", 0.1, 0.2]]), rand_tick([])])
    scenarios.append([rand_tick([["base_velocity", 0.1, 0.2]]), rand_tick([["keyframe", "stow"]]), rand_tick([])])
    scenarios.append([rand_tick([["move_by", "gripper", 0.1]]), rand_tick([["move_by", "gripper", -0.2]])])
    # random sequences
    for _ in range(40):
        ticks = []
        pose = rng.uniform(-1, 1, 3)
        for _k in range(int(rng.integers(2, 8))):
            ops = []
            for _j in range(int(rng.integers(0, 3))):
                r = rng.random()
                a = pos_acts[int(rng.integers(0, len(pos_acts)))]
                if r < 0.3:
                    ops.append(["move_to", a, float(rng.uniform(-0.5, 1.0))])
                elif r < 0.6:
                    ops.append(["move_by", a, float(rng.uniform(-0.2, 0.2))])
                elif r < 0.7:
                    ops.append(["move_by", "base_translate", float(rng.uniform(-0.3, 0.3))])
                elif r < 0.8:
                    ops.append(["move_by", "base_rotate", float(rng.uniform(-0.6, 0.6))])
                elif r < 0.9:
                    ops.append(["base_velocity", float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-1, 1))])
                else:
                    ops.append(["keyframe", ["home", "stow"][int(rng.integers(0, 2))]])
            pose = pose + rng.uniform(-0.15, 0.15, 3)
            ticks.append(rand_tick(ops, pose.tolist()))
        scenarios.append(ticks)
    G["push_command"] = [dict(ticks=tk, expect=run_scenario(tk)) for tk in scenarios]

    # ---- pull_status
    ps = []
    for _ in range(10):
        s = new_server()
        from stretch_mujoco.utils import FpsCounter

        s.physics_fps_counter = FpsCounter()
        d = s.mjdata
        d.length[:] = rng.uniform(-0.5, 1.0, 10); d.velocity[:] = rng.uniform(-6, 6, 10)
        pose = rng.uniform(-2, 2, 3)
        d.xpos[:] = [pose[0], pose[1], 0.0]
        d.xmat[:] = np.array([[np.cos(pose[2]), -np.sin(pose[2]), 0], [np.sin(pose[2]), np.cos(pose[2]), 0], [0, 0, 1]]).reshape(9)
        d.time = float(rng.uniform(0.1, 10))
        s.pull_status()
        st = s.data_proxies.status
        out = {k: [float(getattr(st, k).pos), float(getattr(st, k).vel)] for k in
               ("lift", "arm", "head_pan", "head_tilt", "wrist_yaw", "wrist_pitch", "wrist_roll", "gripper")}
        out["base"] = [float(st.base.x), float(st.base.y), float(st.base.theta), float(st.base.x_vel), float(st.base.theta_vel)]
        out["time"] = float(st.time)
        ps.append(dict(length=d.length.tolist(), velocity=d.velocity.tolist(), pose=pose.tolist(), time=d.time, expect=out))
    G["pull_status"] = ps

    # ---- client-side validation (StretchMujocoSimulator.move_to / move_by)
    class DummyLock:
        def __enter__(self): return self
        def __exit__(self, *a): return False

    cl = object.__new__(StretchMujocoSimulator)
    cl.data_proxies = FakeProxies(); cl.data_proxies.command = StatusCommand.default()
    cl._command_lock = DummyLock()
    cl.is_running = lambda: True
    val = []
    names = ["arm", "gripper", "head_pan", "head_tilt", "lift", "wrist_pitch", "wrist_roll", "wrist_yaw", "base_rotate",
             "base_translate", "left_wheel_vel", "right_wheel_vel", "gripper_left_finger", "gripper_right_finger", "not_an_actuator"]
    for meth in ("move_to", "move_by"):
        for n in names:
            cl.data_proxies.command = StatusCommand.default()
            try:
                getattr(cl, meth)(n, 0.1)
                c = cl.data_proxies.command
                val.append(dict(method=meth, actuator=n, ok=True, move_to=sorted(c.move_to), move_by=sorted(c.move_by)))
            except BaseException as e:  # noqa: BLE001
                val.append(dict(method=meth, actuator=n, ok=False, exc=type(e).__name__, msg=str(e)))
    cl2 = object.__new__(StretchMujocoSimulator)
    cl2.is_running = lambda: False
    try:
        cl2.pull_status()
        nr = "none"
    except BaseException as e:  # noqa: BLE001
        nr = type(e).__name__
    G["validation"] = dict(cases=val, not_running_exc=nr)

    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(G, f, indent=1)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
