"""Lidar sectors (as in the reference's examples/laser_scan.py), depth images and the colour cameras' stand-in images for a batch of robots in the kitchen stand-in.

    python examples/sensors_batch.py [num_envs]
"""
import sys

import torch

sys.path.insert(0, ".")
from stretch_mujoco_amd import StretchBatchSimulator, StretchCameras, StretchSensors  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sim = StretchBatchSimulator(num_envs=B, device="cuda:0", scene="stretch_kitchen_standin",
                            sensors_to_use=StretchSensors.all(), cameras_to_use=StretchCameras.all())
sim.start()
sim.step(33)                                          # 15 Hz lidar cadence in sim time
scan = sim.pull_sensor_data().lidar                   # [B, 360]; index = degrees, 0 = rear, 180 = front (laser_scan.py:32-39)
deg = torch.arange(360, device=sim.device)
sectors = {"front": (deg >= 150) & (deg <= 210), "back": (deg >= 330) | (deg <= 30),
           "right": (deg >= 60) & (deg <= 120), "left": (deg >= 240) & (deg <= 300)}
for name, m in sectors.items():
    d = scan[:, m]
    d = torch.where((d >= 0.2) & (d <= 5), d, torch.full_like(d, float("inf")))
    print(f"{name:5s}: nearest return per env [m]", [round(float(v), 3) for v in d.min(dim=1).values])

cams = sim.pull_camera_data()                         # depth [B, H, W] metres, 0 beyond the camera's limit
for cam in StretchCameras.depth():
    img = cams.get_camera_data(cam)                   # d435i frames are turned upright, as in the reference
    valid = img[img > 0]
    print(cam.name, tuple(img.shape), "valid pixels", f"{float((img > 0).float().mean()):.2f}",
          "median depth", round(float(valid.median()), 3) if valid.numel() else None)
for cam in StretchCameras.rgb():                      # unlit-albedo stand-in for the colour cameras: uint8 [B, H, W, 3], BGR like the reference
    img = cams.get_camera_data(cam)
    print(cam.name, tuple(img.shape), str(img.dtype), "sky pixels", f"{float((img == torch.tensor([255, 224, 169], dtype=torch.uint8, device=img.device)).all(-1).float().mean()):.2f}")
print("K (d435i):", cams.cam_d435i_K.tolist())
sim.stop()
