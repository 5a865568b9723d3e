"""ctypes loader of libsmj.so -- the thin C-ABI binding north_star asks for (include/smj.h).

There is no CPU fallback: if the HIP library is missing this raises, loudly.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SMJ_LIB_PATH") or os.path.join(_HERE, "libsmj.so")   # SMJ_LIB_PATH: another build of the same library (compiler-flag experiments)

SLOT = dict(QPOS=0, QVEL=1, CTRL=2, WARMSTART=3, NSTEP=4, ACT_LENGTH=5, ACT_VELOCITY=6, BASE_POSE=7, GYRO=8, ACCEL=9,
            LIDAR=10, INFO=11, DEBUG=12, PROF=13, XPOSE=14, BASECTL=15)
DIM = dict(NQ=0, NV=1, NU=2, NBODY=3, NLIDAR=4, NKEY=5, NUM_ENVS=6, DEBUG_FLOATS=7, NEFC_MAX=8, NCON_MAX=9, NCAM=10, NV_MAX=11, NSAT_MAX=12)
READ_IMU, READ_LIDAR, READ_POSES = 1, 2, 4
EXPORTS = ("smj_create", "smj_destroy", "smj_bind", "smj_dims", "smj_reset", "smj_step", "smj_set_option",
           "smj_last_error", "smj_version", "smj_render_depth", "smj_render_rgb", "smj_comm_init", "smj_allgather_returns", "smj_comm_destroy",
           "smj_base_controller_tick")

_lib = None


def debug_layout(nvp: int = 32, ncon: int = 16, nsat: int = 0) -> dict:
    """Offsets of the optional debug dump (SMJ_SLOT_DEBUG) for a kernel variant with `nvp` dof lanes and `ncon` contact slots
    (csrc/smj_model.h, smj_debug_layout): standard variant 32 / 16, big variant 64 / 48; `nsat`: satellite capacity of the
    satellite builds (their qacc follows the classic layout, 6 per satellite)."""
    L, o = {}, 0
    for name, n in (("qm", nvp * nvp), ("g", nvp), ("qacc", nvp), ("efc_force", 64), ("efc_b", 64), ("efc_r", 64), ("efc_aref", 64),
                    ("ar_diag", 64), ("xpos", 96), ("qfrc_bias", nvp), ("qfrc_passive", nvp), ("qfrc_act", nvp), ("con", 8 * ncon),
                    ("ar", 64 * 64), ("satqacc", 6 * nsat)):
        L[name] = o
        o += n
    L["floats"] = o
    return L


def full_qacc(debug, layout: dict, model: dict):
    """qacc of the whole model [nv, B] from a debug dump: the main tree's dofs from the classic slot, the satellites' from theirs."""
    import numpy as np

    nsat = int(np.asarray(model.get("k_nsat", [0])).ravel()[0])
    if nsat == 0:
        nv = int(model["dims"][1])
        return debug[layout["qacc"]:layout["qacc"] + nv]
    nvm = int(model["k_main_dims"][1])
    parts = [debug[layout["qacc"]:layout["qacc"] + nvm]]
    si = np.asarray(model["k_sat_i"]).reshape(nsat, -1)
    for k in range(nsat):
        parts.append(debug[layout["satqacc"] + 6 * k:layout["satqacc"] + 6 * k + int(si[k, 4])])
    cat = np.concatenate if isinstance(parts[0], np.ndarray) else __import__("torch").cat
    return cat(parts, 0)


class SmjError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SmjError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950). There is no CPU fallback for the physics path.")
    L = ctypes.CDLL(LIB_PATH)
    vp, ci, cl, cu = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_uint
    L.smj_create.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ci, ci, ctypes.POINTER(vp)]
    L.smj_destroy.argtypes = [vp]
    L.smj_bind.argtypes = [vp, ci, vp, cl]
    L.smj_dims.argtypes = [vp, ctypes.POINTER(ci)]
    L.smj_reset.argtypes = [vp, vp, vp]
    L.smj_step.argtypes = [vp, ci, cu, vp]
    L.smj_render_depth.argtypes = [vp, ci, ci, ci, ctypes.c_float, ctypes.c_float, vp, vp]
    L.smj_render_rgb.argtypes = [vp, ci, ci, ci, ctypes.c_float, vp, vp, vp]
    L.smj_set_option.argtypes = [vp, ctypes.c_char_p, ctypes.c_double]
    L.smj_base_controller_tick.argtypes = [vp, vp]
    L.smj_comm_init.argtypes = [vp, ci, ci, ctypes.c_char_p, ctypes.c_double]
    L.smj_allgather_returns.argtypes = [vp, vp, vp, ci, vp]
    L.smj_comm_destroy.argtypes = [vp]
    L.smj_last_error.argtypes = [vp]
    L.smj_last_error.restype = ctypes.c_char_p
    L.smj_version.restype = ctypes.c_char_p
    _lib = L
    return L


def check(L, ctx, rc: int, what: str):
    if rc != 0:
        msg = L.smj_last_error(ctx).decode() if ctx else "no context"
        raise SmjError(f"{what} failed ({rc}): {msg}")
