"""Compiled-model container ("SMJB"): a flat table of named numeric arrays.

This is the on-disk / in-memory format handed to the C-ABI (`smj_create`,
include/smj.h) and to the CPU oracle.  It plays the role MuJoCo's compiled
`MjModel` plays in the reference (built at `stretch_mujoco/mujoco_server.py:252`).

Layout (little endian):
    char[8]  magic  "SMJB0001"
    u32      n_entries
    u32      reserved
    entry[n]: char name[48]; u32 dtype (0=f64, 1=i32, 2=u8, 3=f32); u32 ndim;
              u32 shape[4]; u64 offset (from file start); u64 nbytes
    payload, every array 16-byte aligned
"""
from __future__ import annotations

import struct
from typing import Dict

import numpy as np

MAGIC = b"SMJB0001"
_ENTRY = struct.Struct("<48sII4IQQ")
_DTYPES = {0: np.float64, 1: np.int32, 2: np.uint8, 3: np.float32}
_CODES = {np.dtype(np.float64): 0, np.dtype(np.int32): 1, np.dtype(np.uint8): 2, np.dtype(np.float32): 3}


def _canon(a) -> np.ndarray:
    a = np.asarray(a)
    if a.dtype.kind == "f" and a.dtype != np.float32:   # float32 is kept as stored (bulk render-mesh vertices)
        a = a.astype(np.float64)
    elif a.dtype.kind in "iub" and a.dtype != np.uint8:
        a = a.astype(np.int32)
    elif a.dtype.kind in "SU":
        a = np.frombuffer(str(a).encode("utf-8"), dtype=np.uint8)
    if a.ndim == 0:
        a = a.reshape(1)
    if a.ndim > 4:
        raise ValueError("at most 4 dims")
    return np.ascontiguousarray(a)


def dumps(arrays: Dict[str, np.ndarray]) -> bytes:
    items = [(k, _canon(v)) for k, v in arrays.items()]
    head = 16 + _ENTRY.size * len(items)
    off = (head + 15) // 16 * 16
    table = bytearray()
    payload = bytearray()
    base = off
    for name, a in items:
        nb = a.nbytes
        shape = list(a.shape) + [1] * (4 - a.ndim)
        bname = name.encode("ascii")
        if len(bname) > 47:
            raise ValueError(f"name too long: {name}")
        table += _ENTRY.pack(bname, _CODES[a.dtype], a.ndim, *shape, base + len(payload), nb)
        payload += a.tobytes()
        payload += b"\0" * ((-len(payload)) % 16)
    out = bytearray(MAGIC) + struct.pack("<II", len(items), 0) + table
    out += b"\0" * (off - len(out))
    out += payload
    return bytes(out)


def loads(buf: bytes) -> Dict[str, np.ndarray]:
    if buf[:8] != MAGIC:
        raise ValueError("not an SMJB model blob")
    n, _ = struct.unpack_from("<II", buf, 8)
    out: Dict[str, np.ndarray] = {}
    for i in range(n):
        name, code, ndim, s0, s1, s2, s3, off, nb = _ENTRY.unpack_from(buf, 16 + i * _ENTRY.size)
        name = name.rstrip(b"\0").decode("ascii")
        shape = (s0, s1, s2, s3)[:ndim]
        dt = _DTYPES[code]
        out[name] = np.frombuffer(buf, dtype=dt, count=nb // np.dtype(dt).itemsize, offset=off).reshape(shape).copy()
    return out


def save(path: str, arrays: Dict[str, np.ndarray]) -> None:
    with open(path, "wb") as f:
        f.write(dumps(arrays))


def load(path: str) -> Dict[str, np.ndarray]:
    with open(path, "rb") as f:
        return loads(f.read())


def get_str(arrays: Dict[str, np.ndarray], key: str) -> str:
    return bytes(arrays[key]).decode("utf-8")
