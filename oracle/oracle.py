"""TEST INFRASTRUCTURE ONLY -- ctypes wrapper of oracle/libsmj_oracle.so (fp64 CPU restatement).

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
PARITY UNPINNED: see smj_oracle.h.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build() -> str:
    path = os.path.join(_HERE, "libsmj_oracle.so")
    src = os.path.join(_HERE, "smj_oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return path


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.smjo_load.restype = ctypes.c_void_p
        L.smjo_load.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
        L.smjo_make_data.restype = ctypes.c_void_p
        L.smjo_make_data.argtypes = [ctypes.c_void_p]
        for f in ("smjo_reset", "smjo_forward", "smjo_step"):
            getattr(L, f).argtypes = [ctypes.c_void_p, ctypes.c_void_p]
            getattr(L, f).restype = None
        L.smjo_step_n.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.smjo_step_n.restype = None
        L.smjo_sensors.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.smjo_sensors.restype = None
        L.smjo_render_depth.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_double, ctypes.c_double, ctypes.c_void_p]
        L.smjo_render_depth.restype = ctypes.c_int
        L.smjo_render_geomid.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                         ctypes.c_void_p, ctypes.c_void_p]
        L.smjo_render_geomid.restype = ctypes.c_int
        L.smjo_set_option.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_double]
        L.smjo_get.restype = ctypes.POINTER(ctypes.c_double)
        L.smjo_get.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
        L.smjo_get_int.restype = ctypes.POINTER(ctypes.c_int)
        L.smjo_get_int.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)]
        L.smjo_dim.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        L.smjo_flops.argtypes = [ctypes.c_int]
        L.smjo_flops.restype = ctypes.c_longlong
        L.smjo_set_contacts.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.smjo_set_contacts.restype = None
        for f in ("smjo_free_data", "smjo_free_model"):
            getattr(L, f).argtypes = [ctypes.c_void_p]
            getattr(L, f).restype = None
        L.smjo_mc_words.restype = ctypes.c_int
        for f in ("smjo_mc_get", "smjo_mc_set"):
            getattr(L, f).argtypes = [ctypes.c_void_p, ctypes.c_void_p]
            getattr(L, f).restype = None
        _LIB = L
    return _LIB


_SIZES = {"qpos": "nq", "ctrl": "nu", "xpos": ("nbody", 3), "xquat": ("nbody", 4), "xmat": ("nbody", 9),
          "xipos": ("nbody", 3), "ximat": ("nbody", 9), "xanchor": ("njnt", 3), "xaxis": ("njnt", 3),
          "geom_xpos": ("ngeom", 3), "geom_xmat": ("ngeom", 9), "site_xpos": ("nsite", 3), "site_xmat": ("nsite", 9),
          "cam_xpos": ("ncam", 3), "cam_xmat": ("ncam", 9),
          "subtree_com": ("nbody", 3), "cinert": ("nbody", 10), "crb": ("nbody", 10), "cvel": ("nbody", 6),
          "cacc": ("nbody", 6), "ten_length": 1, "actuator_length": "nu", "actuator_velocity": "nu",
          "actuator_force": "nu", "actuator_moment": ("nu", "nv"), "lidar": "nlidar"}

CONTACT_FIELDS = 29  # doubles per contact_t (27 doubles + 4 ints)


class Oracle:
    """One environment, fp64.  Arrays are numpy views into the C data (no copies)."""

    def __init__(self, blob: bytes):
        self.L = lib()
        self._blob = blob
        self.m = self.L.smjo_load(blob, len(blob))
        if not self.m:
            raise ValueError("bad model blob")
        self.d = self.L.smjo_make_data(self.m)

    def close(self):
        """Free the C side.  Explicit, not __del__: arr() hands out numpy VIEWS of the C arrays, and tests keep such views after the
        Oracle object is gone.  The rollout harness, which makes thousands of short-lived oracles (one per perturbed evaluation), closes them."""
        if getattr(self, "d", None):
            self.L.smjo_free_data(self.d); self.d = None
        if getattr(self, "m", None):
            self.L.smjo_free_model(self.m); self.m = None

    def dim(self, name: str) -> int:
        return self.L.smjo_dim(self.m, name.encode())

    def set_option(self, name: str, value: float):
        if self.L.smjo_set_option(self.m, name.encode(), float(value)) != 0:
            raise KeyError(name)

    def arr(self, name: str) -> np.ndarray:
        n = ctypes.c_int(0)
        p = self.L.smjo_get(self.d, name.encode(), ctypes.byref(n))
        if not p:
            raise KeyError(name)
        cnt = n.value
        shape = None
        if cnt < 0:
            s = _SIZES[name]
            if isinstance(s, tuple):
                shape = tuple(self.dim(x) if isinstance(x, str) else x for x in s)
                cnt = int(np.prod(shape))
            else:
                cnt = self.dim(s) if isinstance(s, str) else s
        a = np.ctypeslib.as_array(p, shape=(cnt,))
        return a.reshape(shape) if shape else a

    def iarr(self, name: str) -> np.ndarray:
        n = ctypes.c_int(0)
        p = self.L.smjo_get_int(self.d, name.encode(), ctypes.byref(n))
        if not p:
            raise KeyError(name)
        return np.ctypeslib.as_array(p, shape=(n.value,))

    @property
    def time(self) -> float:
        return float(self.arr("time")[0])

    @property
    def nefc(self) -> int:
        return int(self.iarr("nefc")[0])

    @property
    def ncon(self) -> int:
        return int(self.iarr("ncon")[0])

    def reset(self):
        self.L.smjo_reset(self.m, self.d)

    def forward(self):
        self.L.smjo_forward(self.m, self.d)

    def step(self, n: int = 1):
        self.L.smjo_step_n(self.m, self.d, n)

    def set_contacts(self, con: np.ndarray):
        """The next forward()/step() uses these contacts [n, 9] = (dist, pos3, normal3, geom1, geom2) instead of its own
        collision stage (one-shot): dynamics on identical contacts."""
        con = np.ascontiguousarray(con, np.float64).reshape(-1, 9)
        self.L.smjo_set_contacts(self.d, len(con), con.ctypes.data_as(ctypes.c_void_p))

    def mc_export(self) -> np.ndarray:
        """The kept contact manifolds (option manifold_keep, the twin of the kernels' manifold cache) as one buffer."""
        buf = np.zeros(self.L.smjo_mc_words(), np.float64)
        self.L.smjo_mc_get(self.d, buf.ctypes.data_as(ctypes.c_void_p))
        return buf

    def mc_import(self, buf: np.ndarray):
        buf = np.ascontiguousarray(buf, np.float64)
        assert buf.size == self.L.smjo_mc_words()
        self.L.smjo_mc_set(self.d, buf.ctypes.data_as(ctypes.c_void_p))

    def render_depth(self, cam: int, width: int, height: int, fovy_deg: float, max_depth: float = 0.0) -> np.ndarray:
        """Depth image [height, width] (fp32) of camera `cam` from the poses of the last forward()/step()."""
        out = np.zeros((height, width), np.float32)
        rc = self.L.smjo_render_depth(self.m, self.d, int(cam), int(width), int(height), float(fovy_deg), float(max_depth),
                                      out.ctypes.data_as(ctypes.c_void_p))
        if rc != 0:
            raise ValueError("model blob has no render tables, or bad camera id")
        return out

    def render_geomid(self, cam: int, width: int, height: int, fovy_deg: float):
        """(geom id image [H, W] int32 with -1 = nothing hit, albedo image [H, W, 3] uint8) of camera `cam`: the RGB stand-in."""
        gid = np.zeros((height, width), np.int32)
        rgb = np.zeros((height, width, 3), np.uint8)
        rc = self.L.smjo_render_geomid(self.m, self.d, int(cam), int(width), int(height), float(fovy_deg),
                                       gid.ctypes.data_as(ctypes.c_void_p), rgb.ctypes.data_as(ctypes.c_void_p))
        if rc != 0:
            raise ValueError("model blob has no render tables, or bad camera id")
        return gid, rgb

    def sensors(self, with_lidar: bool = True):
        self.L.smjo_sensors(self.m, self.d, int(with_lidar))
