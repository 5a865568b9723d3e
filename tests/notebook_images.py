"""Shared by tests/test_notebook_images.py (oracle) and tests/test_gpu_depth.py (HIP renderer): the camera images the reference's
notebook stores (tests/golden/notebook_images.npz, decoded by tools/gen_notebook_images_golden.py), the display pipeline they went
through, and the scene state they were taken in.  Test infrastructure."""
import math
import os

import numpy as np

from conftest import GOLDEN, HOME_CTRL

# docs/getting_started.ipynb cell 14: the frames were 640 x 480 with these focal lengths in pixels (cam_d405_K, cam_d435i_K)
NB_W, NB_H = 640, 480
NB_F_D405, NB_F_D435I = 514.682, 399.427
# cell 20: pull_status() at t = 8.26 s prints the base pose (x, y, theta); the base settles within 250 steps of start() and never
# moves again while the notebook only moves the head, so this is the pose of every image of cells 15 and 23
NB_BASE_POSE = (-0.0122, 0.0044, -0.0650)
# camera ids in stretch.xml order
CAM_D405_RGB, CAM_D405_DEPTH, CAM_D435I_RGB, CAM_D435I_DEPTH, CAM_NAV = range(5)


def fovy_of(f_px, height=NB_H):
    return 2 * math.degrees(math.atan(0.5 * height / f_px))


def golden():
    return np.load(os.path.join(GOLDEN, "notebook_images.npz"))


def display(img01, size_wh):
    """mediapy.show_images(..., vmin=0, vmax=1, height=h) as far as the stored pixels depend on it: clip to 0..1, Lanczos resampling to
    the display size, 8-bit.  img01: [H, W] or [H, W, C] floats."""
    from PIL import Image

    x = np.clip(np.asarray(img01, np.float32), 0, 1)
    if x.ndim == 2:
        r = np.asarray(Image.fromarray(x, mode="F").resize(size_wh, Image.LANCZOS))
    else:
        r = np.stack([np.asarray(Image.fromarray(np.ascontiguousarray(x[..., c]), mode="F").resize(size_wh, Image.LANCZOS))
                      for c in range(x.shape[-1])], -1)
    return np.clip(r * 255 + 0.5, 0, 255).astype(np.uint8)


def display_mask(mask, size_wh):
    return display(mask.astype(np.float32), size_wh) > 127


def hsv(rgb8):
    """[..., 3] uint8 -> hue in degrees, saturation, value."""
    c = rgb8.astype(np.float64) / 255
    v = c.max(-1)
    d = v - c.min(-1)
    s = np.where(v > 0, d / np.maximum(v, 1e-12), 0)
    r, g, b = c[..., 0], c[..., 1], c[..., 2]
    dd = np.maximum(d, 1e-12)
    h = np.where(v == r, ((g - b) / dd) % 6, np.where(v == g, (b - r) / dd + 2, (r - g) / dd + 4)) * 60
    return np.where(d > 0, h, 0), s, v


def colour_classes(rgb8):
    """What can be told apart in MuJoCo's lit image without its renderer: the skybox, the wood texture of the table, the red cylinder
    and the blue box of models/scene.xml."""
    h, s, v = hsv(rgb8)
    return dict(sky=(h > 185) & (h < 215) & (v > 0.6) & (s > 0.2),
                wood=(h > 15) & (h < 50) & (s > 0.25) & (v > 0.12),
                red=((h < 12) | (h > 348)) & (s > 0.6) & (v > 0.25),
                blue=(h > 200) & (h < 260) & (s > 0.5) & (v > 0.2))


def iou(a, b):
    return float((a & b).sum()) / max(1, int((a | b).sum()))


def yaw_of(quat):
    w, x, y, z = quat
    return math.atan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z))


def place_base(qpos, x, y, theta):
    """The robot moved rigidly in the floor plane to the base pose (x, y, theta): its own state, and what rests on the table, do not
    change.  (Free joint of base_link: qpos[0:7].)"""
    q = np.array(qpos, np.float64).copy()
    dth = theta - yaw_of(q[3:7])
    w2, z2 = math.cos(dth / 2), math.sin(dth / 2)
    w, xx, yy, zz = q[3:7]
    q[3:7] = [w2 * w - z2 * zz, w2 * xx - z2 * yy, w2 * yy + z2 * xx, w2 * zz + z2 * w]
    q[0], q[1] = x, y
    return q


def scene_geoms(model):
    """Geom ids of scene.xml's table (the static box), blue box and red cylinder, by what the compiled model says about them."""
    t = np.asarray(model["geom_type"])
    rgba = np.asarray(model["geom_rgba"], float).reshape(-1, 4)
    body = np.asarray(model["geom_bodyid"])
    table = [i for i in range(len(t)) if t[i] == 6 and body[i] == 0]
    blue = [i for i in range(len(t)) if t[i] == 6 and body[i] != 0 and rgba[i][2] > 0.4 and rgba[i][0] < 0.3 and np.asarray(model["geom_group"])[i] == 0]
    red = [i for i in range(len(t)) if t[i] == 5 and rgba[i][0] > 0.7 and rgba[i][1] < 0.3]
    assert len(table) == 1 and len(blue) == 1 and len(red) == 1, (table, blue, red)
    return table[0], blue[0], red[0]


def oracle_at_cell15(blob):
    """The state of cell 15's images: default scene, start() (home keyframe targets), t = 3.2 s; base at the printed pose."""
    from oracle.oracle import Oracle

    o = Oracle(blob)
    o.set_option("solver", 2)
    o.arr("ctrl")[:] = HOME_CTRL
    o.step(1601)
    return o


def nav_display_fovy(fovy_final, w=533, h=400):
    """The notebook's nav frames are landscape with the arm pointing right: today's camera (stretch.xml:468) turned by rot90(+1), which
    is what StatusStretchCameras applies (status_stretch_camera.py:60-80).  Raw render: h wide, w high; its vertical field of view is
    the horizontal one of the displayed frame."""
    return 2 * math.degrees(math.atan(math.tan(math.radians(fovy_final / 2)) * w / h))
