"""TEST INFRASTRUCTURE ONLY -- ctypes wrapper of tests/emul/libsmj_emul.so (CPU lane emulator of the HIP kernel)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = {}

SLOTS = dict(qpos=0, qvel=1, ctrl=2, warm=3, nstep=4, act_len=5, act_vel=6, base=7, gyro=8, accel=9, lidar=10, info=11, debug=12, bctl=15)


def lib(variant: str = "standard"):
    big = variant   # cache key
    if big not in _LIB:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
        L = ctypes.CDLL(os.environ.get("SMJ_EMUL_LIB_" + variant.upper()) or os.path.join(_HERE, {"standard": "libsmj_emul.so", "tall": "libsmj_emul_tall.so", "mid": "libsmj_emul_mid.so", "big": "libsmj_emul_big.so", "big38": "libsmj_emul_big38.so", "big50": "libsmj_emul_big50.so", "poison": "libsmj_emul_poison.so", "sat": "libsmj_emul_sat.so", "sat32": "libsmj_emul_sat32.so"}[variant]))   # (SMJ_EMUL_LIB_<VARIANT>: an experimental build of that variant, tools only)
        L.emul_create.restype = ctypes.c_void_p
        L.emul_create.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
        L.emul_bind.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_long]
        L.emul_step.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint]
        L.emul_set_option.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_double]
        L.emul_destroy.argtypes = [ctypes.c_void_p]
        _LIB[big] = L
    return _LIB[big]


class Emul:
    def __init__(self, blob: bytes, dims: dict, num_envs: int = 1, debug: bool = True, big: bool | None = None, variant: str | None = None):
        """variant: "standard" (32 dofs / 80 rows / 16 contacts), "tall" (32 / 160 / 48), "mid" (32 / 128 / 44: the build of the tall
        variant that smj_create runs as the primary kernel; the emulator has no escalation, so the default stays "tall") or "big"
        (64 / 160 / 48); default: chosen
        like smj_create does, by the model's size and the blob's capacity hint.  `big=True` is shorthand for variant="big"."""
        if variant is None:
            import stretch_mujoco_amd.model_blob as mb0

            ns = mb0.loads(blob).get("k_nsat")
            if ns is not None and int(np.asarray(ns).ravel()[0]) > 0:
                variant = "sat" if int(np.asarray(ns).ravel()[0]) <= 16 else "sat32"
        if variant is None:
            if big or dims["nv"] > 32:   # the big variant is built for 38 / 50 / 64 dof columns (smj_model.h); big=True with a small model: 64
                variant = "big38" if 32 < dims["nv"] <= 38 else "big50" if 38 < dims["nv"] <= 50 else "big"
            else:
                import stretch_mujoco_amd.model_blob as mb

                hint = mb.loads(blob).get("k_capacity_hint")
                variant = "tall" if hint is not None and int(np.asarray(hint).ravel()[0]) > 0 else "standard"
        self.variant = variant
        self.big = variant.startswith("big")
        self.L = lib(variant)
        self.nvp, self.ncon_max = self.L.emul_nvp(), self.L.emul_ncon_max()
        self.nsat_max = {"sat": 16, "sat32": 32}.get(variant, 0)
        self.B = B = num_envs
        self.c = self.L.emul_create(blob, len(blob), B)
        if not self.c:
            raise ValueError("emul_create failed")
        nq, nv, nu, nl = dims["nq"], dims["nv"], dims["nu"], dims["nlidar"]
        f = np.float32
        self.buf = dict(qpos=np.zeros((nq, B), f), qvel=np.zeros((nv, B), f), ctrl=np.zeros((nu, B), f),
                        warm=np.zeros((nv, B), f), nstep=np.zeros(B, np.int32), act_len=np.zeros((nu, B), f),
                        act_vel=np.zeros((nu, B), f), base=np.zeros((3, B), f), gyro=np.zeros((3, B), f),
                        accel=np.zeros((3, B), f), lidar=np.zeros((max(nl, 1), B), f), info=np.zeros((4, B), np.int32),
                        bctl=np.zeros((8, B), f))
        if debug:
            self.buf["debug"] = np.zeros((self.L.emul_debug_floats(), B), f)
        for k, a in self.buf.items():
            self.L.emul_bind(self.c, SLOTS[k], a.ctypes.data_as(ctypes.c_void_p), B)
        # PGS: the emulator starts the sweeps the way MuJoCo does unless a test asks for the kernels' default (a second start from the
        # previous step's forces, option pgs_dual_warmstart = 1) -- most PGS tests compare iterate for iterate with the unmodified oracle
        self.set_option("pgs_dual_warmstart", 0)

    def set_option(self, name, v):
        assert self.L.emul_set_option(self.c, name.encode(), float(v)) == 0

    def clear_caches(self):
        """What smj_reset drops beside the state: kept manifolds, separating directions, the PGS second start."""
        self.L.emul_clear_caches.argtypes = [ctypes.c_void_p]
        self.L.emul_clear_caches(self.c)

    def set_poison(self, byte):
        """Fill the emulated LDS with `byte` before every launch (-1: leave whatever the previous launch left)."""
        self.L.emul_set_poison(int(byte))

    def step(self, n=1, read_flags=0):
        self.L.emul_step(self.c, n, read_flags)

    def __getattr__(self, k):
        if k in self.__dict__.get("buf", {}):
            return self.buf[k]
        raise AttributeError(k)
