import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MODELS = os.path.join(ROOT, "stretch_mujoco_amd", "models")
GOLDEN = os.path.join(ROOT, "tests", "golden")
HOME_CTRL = [0, 0, 0.6, 0.1, 0, 0, 0, 0, 0, 0]
MIX_CTRL = [2, -1, 0.6, 0.1, 1, -0.4, 0.5, 0.02, 0.3, -0.2]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def blob_fused() -> bytes:
    with open(os.path.join(MODELS, "stretch_empty.smjb"), "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def blob_full() -> bytes:
    with open(os.path.join(MODELS, "stretch_empty_full.smjb"), "rb") as f:
        return f.read()


def home_qpos(qpos0):
    """qpos0 with the lift at 0.6 and the arm at 0.1 (the 'home' keyframe targets, stretch.xml:544).  At qpos0 itself the
    lift is fully down and the wrist sits 5 cm inside the base hull: a legitimate MuJoCo start (the reference homes the
    robot right after start), but the contact normal of such a deep, degenerate penetration is round-off sensitive, so
    trajectory-parity tests start from this clear pose instead."""
    import numpy as np

    q = np.array(qpos0, dtype=np.float64).copy()
    q[9] = 0.6
    q[10:14] = 0.025
    return q


@pytest.fixture(scope="session")
def blob_kitchen():
    with open(os.path.join(MODELS, "stretch_kitchen_standin.smjb"), "rb") as f:
        return f.read()
