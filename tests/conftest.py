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
