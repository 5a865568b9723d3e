"""The C-ABI library loads on a box without a GPU and exports every symbol include/smj.h declares."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from stretch_mujoco_amd import lib


def _declared():
    with open(os.path.join(ROOT, "include", "smj.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(smj_[a-z_]+)\s*\(", text)))


def test_header_and_loader_agree():
    assert _declared() == sorted(lib.EXPORTS)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(lib.LIB_PATH):
        pytest.fail(f"{lib.LIB_PATH} not built: run __graft_entry__.build()")
    L = lib.load()
    for sym in _declared():
        assert hasattr(L, sym), sym
    assert b"gfx950" in L.smj_version()


def test_slot_and_dim_enums_match_header():
    with open(os.path.join(ROOT, "include", "smj.h")) as f:
        text = f.read()
    for name, val in lib.SLOT.items():
        assert re.search(rf"SMJ_SLOT_{name}\w*\s*=\s*{val}\b", text), name
    for name, val in lib.DIM.items():
        assert re.search(rf"SMJ_DIM_{name}\s*=\s*{val}\b", text), name


def test_bad_blob_is_rejected_without_touching_the_gpu():
    L = lib.load()
    ctx = ctypes.c_void_p()
    assert L.smj_create(b"garbage-garbage-garbage", 23, 4, 0, ctypes.byref(ctx)) != 0
    assert not ctx


def test_no_cpu_fallback_in_the_product():
    """The physics path must fail loudly without the ROCm device, and the package must not reach into oracle/."""
    from stretch_mujoco_amd import StretchBatchSimulator

    sim = StretchBatchSimulator(num_envs=2, device="cpu")
    with pytest.raises(lib.SmjError):
        sim.start()
    with pytest.raises(ConnectionError):
        sim.pull_status()
    pkg = os.path.join(ROOT, "stretch_mujoco_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            path = os.path.join(dirpath, fn)
            if fn.endswith(".py"):
                with open(path) as f:
                    for line in f:
                        code = line.split("#")[0]
                        assert not re.search(r"\b(import|from)\s+(oracle|tests)\b", code), (fn, line)
                        assert "libsmj_oracle" not in code and "libsmj_emul" not in code, (fn, line)
            elif fn.endswith((".h", ".hip", ".cpp")):
                with open(path) as f:
                    for line in f:
                        if line.lstrip().startswith("#include"):
                            assert "oracle" not in line and "emul" not in line, (fn, line)


def test_entry_points_and_scripts_compile():
    """bench.py, __graft_entry__.py, tools/ and examples/ are only ever executed on the GPU box: at least their syntax is checked here."""
    import glob

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")] + sorted(glob.glob(os.path.join(root, "tools", "*.py")))
    files += sorted(glob.glob(os.path.join(root, "examples", "*.py")))
    assert len(files) > 20
    for f in files:
        with open(f) as fh:
            compile(fh.read(), f, "exec")
