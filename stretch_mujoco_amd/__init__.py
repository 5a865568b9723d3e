"""MI355X-native batched Stretch simulator."""
from .enums import Actuators, StretchCameras, StretchSensors  # noqa: F401

__all__ = ["Actuators", "StretchCameras", "StretchSensors", "StretchBatchSimulator"]


def __getattr__(name):
    if name == "StretchBatchSimulator":
        from .simulator import StretchBatchSimulator

        return StretchBatchSimulator
    raise AttributeError(name)
