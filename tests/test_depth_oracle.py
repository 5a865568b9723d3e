"""Depth-camera restatement (oracle/smj_oracle.c smjo_render_depth) checked against closed forms and a brute-force ray
caster, and the host-side camera data model against the reference's conventions.  CPU only."""
import math

import numpy as np
import pytest

from conftest import home_qpos
from oracle.oracle import Oracle
from stretch_mujoco_amd import model_blob
from stretch_mujoco_amd.datamodels import StatusStretchCameras
from stretch_mujoco_amd.enums import StretchCameras
from stretch_mujoco_amd.utils import compute_K

D405, D435 = 1, 3   # camera ids in stretch.xml order (d405_rgb, d405_depth, d435i_camera_rgb, d435i_camera_depth, nav)


def quat2mat(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


@pytest.fixture(scope="module")
def posed(blob_fused):
    m = model_blob.loads(blob_fused)
    o = Oracle(blob_fused)
    q = home_qpos(m["qpos0"])
    o.arr("qpos")[:] = q
    o.forward()
    return m, o


def pixel_rays(cam_mat, W, H, fovy):
    th = math.tan(fovy * math.pi / 360)
    u, v = np.meshgrid(np.arange(W), np.arange(H))
    dc = np.stack([((u + 0.5) / W * 2 - 1) * th * W / H, (1 - (v + 0.5) / H * 2) * th, -np.ones_like(u, float)], -1)
    return dc @ cam_mat.T


def test_names_order(posed):
    m, _ = posed
    import json
    names = json.loads(model_blob.get_str(m, "names_json"))["camera"]
    assert names == ["d405_rgb", "d405_depth", "d435i_camera_rgb", "d435i_camera_depth", "nav_camera_rgb"]
    for cam in StretchCameras.depth():
        assert names.index(cam.camera_name_in_mjcf) in (D405, D435)


def test_plane_closed_form(posed):
    """Wherever a ray of the head camera reaches the floor unobstructed the depth is the closed form -c_z / d_z, and no
    pixel is ever deeper than the floor."""
    m, o = posed
    W, H, fovy = 106, 60, 42.0
    img = o.render_depth(D435, W, H, fovy, 0.0)
    cp = o.arr("cam_xpos").reshape(-1, 3)[D435]
    cm = o.arr("cam_xmat").reshape(-1, 3, 3)[D435]
    d = pixel_rays(cm, W, H, fovy)
    zfar = m["vis_znear_zfar_extent"][1] * m["vis_znear_zfar_extent"][2]
    with np.errstate(divide="ignore"):
        floor = np.where(d[..., 2] < 0, -cp[2] / d[..., 2], np.inf)
    floor = np.where(floor > zfar, zfar, floor)
    assert np.all(img <= floor * (1 + 1e-6) + 1e-6)
    on_floor = np.isclose(img, floor, rtol=1e-6, atol=1e-6) & np.isfinite(floor) & (floor < zfar)
    assert on_floor.mean() > 0.2
    sky = d[..., 2] >= 0
    assert sky.any() and np.all(img[sky & (img == img.max())] == np.float32(zfar))


def test_limit_semantics(posed):
    """utils.limit_depth_distance (utils.py:87-91): strictly beyond the limit -> 0, the far plane included."""
    _, o = posed
    raw = o.render_depth(D435, 53, 30, 42.0, 0.0)
    lim = o.render_depth(D435, 53, 30, 42.0, 10.0)
    assert np.array_equal(lim, np.where(raw > 10.0, 0, raw))
    assert (lim == 0).any() and (lim > 0).any()


def test_bvh_matches_brute_force(posed):
    """The median-split tree walk returns exactly what testing every front-facing triangle of every visible mesh does."""
    m, o = posed
    W, H, fovy = 24, 14, 58.0
    img = o.render_depth(D405, W, H, fovy, 0.0)
    cp = o.arr("cam_xpos").reshape(-1, 3)[D405]
    cm = o.arr("cam_xmat").reshape(-1, 3, 3)[D405]
    gx = o.arr("geom_xpos").reshape(-1, 3)
    gm = o.arr("geom_xmat").reshape(-1, 3, 3)
    rays = pixel_rays(cm, W, H, fovy).reshape(-1, 3)
    zn, zf, ext = m["vis_znear_zfar_extent"]
    near, far = zn * ext, zf * ext
    best = np.full(len(rays), far * (1 + 1e-6))
    V = m["rmesh_vert"].astype(np.float64)
    for g in range(len(m["geom_type"])):
        rm = m["geom_rmeshid"][g]
        if m["geom_group"][g] > 2 or m["geom_rgba"][g][3] == 0:
            continue
        if m["geom_type"][g] == 0:   # plane
            with np.errstate(divide="ignore"):
                t = np.where(rays[:, 2] < 0, -cp[2] / rays[:, 2], np.inf)
            best = np.where((t >= near) & (t < best), t, best)
            continue
        if rm < 0:
            continue
        v = V[m["rmesh_vertadr"][rm]: m["rmesh_vertadr"][rm] + m["rmesh_vertnum"][rm]]
        f = m["rmesh_face"][m["rmesh_faceadr"][rm]: m["rmesh_faceadr"][rm] + m["rmesh_facenum"][rm]]
        lo = (cp - gx[g]) @ gm[g]
        ld = rays @ gm[g]
        a, e1, e2 = v[f[:, 0]], v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]]
        p = np.cross(ld[:, None, :], e2[None])                 # [R, T, 3]
        det = np.einsum("tk,rtk->rt", e1, p)
        tv = lo - a                                            # [T, 3]
        u = np.einsum("tk,rtk->rt", tv, p)
        q = np.cross(tv, e1)                                   # [T, 3]
        w = ld @ q.T
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (e2 * q).sum(-1)[None] / det
        ok = (det > 1e-30) & (u >= 0) & (w >= 0) & (u + w <= det) & (t >= near)
        t = np.where(ok, t, np.inf).min(axis=1)
        best = np.minimum(best, t)
    best = np.where(best > far, far, best).reshape(H, W)
    assert np.allclose(img, best, rtol=1e-6, atol=1e-7)
    assert (img < 0.5).sum() > 10      # the gripper is in view of the wrist camera


def test_camera_intrinsics_and_rotation():
    """get_camera_params (mujoco_server_camera_manager.py:168-183) builds K from fovy and the SENSOR resolution; the d435i
    frames are turned upright by rot90(-1) (status_stretch_camera.py:73-76)."""
    import torch

    st = StretchCameras.cam_d405_depth.initial_camera_settings
    assert (st.width, st.height, st.field_of_view_vertical_in_degrees) == (480, 270, 58)
    K = compute_K(st.field_of_view_vertical_in_degrees, *st.sensor_resolution)
    assert np.allclose(K, [[0.5 * 720 / math.tan(math.radians(29)), 0, 640], [0, 0.5 * 720 / math.tan(math.radians(29)), 360], [0, 0, 1]])
    st = StretchCameras.cam_d435i_depth.initial_camera_settings
    assert (st.width, st.height, st.field_of_view_vertical_in_degrees, st.sensor_resolution) == (424, 240, 42, (1920, 1080))
    assert StretchCameras.cam_d405_depth.depth_limit == 1 and StretchCameras.cam_d435i_depth.depth_limit == 10
    img = torch.arange(2 * 3 * 4, dtype=torch.float32).reshape(2, 3, 4)
    s = StatusStretchCameras.default()
    with pytest.raises(ValueError):
        s.get_camera_data(StretchCameras.cam_d435i_depth)
    s.set_camera_data(StretchCameras.cam_d435i_depth, img)
    s.set_camera_data(StretchCameras.cam_d405_depth, img)
    up = s.get_camera_data(StretchCameras.cam_d435i_depth)
    assert up.shape == (2, 4, 3)
    assert np.array_equal(up[1].numpy(), np.rot90(img[1].numpy(), -1))
    assert s.get_camera_data(StretchCameras.cam_d435i_depth, auto_rotate=False) is img
    assert s.get_camera_data(StretchCameras.cam_d405_depth) is img
    assert set(s.get_all()) == {StretchCameras.cam_d405_depth, StretchCameras.cam_d435i_depth}
    with pytest.raises(ValueError):
        s.get_camera_data(StretchCameras.cam_nav_rgb)
    # colour images [B, H, W, 3]: status_stretch_camera.py:60-80 -- d405 as is, d435i turned by rot90(-1), nav by rot90(+1), each
    # in BGR order unless auto_correct_rgb is off
    import torch
    rgb = torch.arange(2 * 3 * 4 * 3, dtype=torch.uint8).reshape(2, 3, 4, 3)
    for cam, k in ((StretchCameras.cam_d405_rgb, 0), (StretchCameras.cam_d435i_rgb, -1), (StretchCameras.cam_nav_rgb, 1)):
        s.set_camera_data(cam, rgb)
        got = s.get_camera_data(cam)
        assert np.array_equal(got[1].numpy(), np.rot90(rgb[1].numpy(), k)[..., ::-1])
        assert np.array_equal(s.get_camera_data(cam, auto_correct_rgb=False)[0].numpy(), np.rot90(rgb[0].numpy(), k))
        assert s.get_camera_data(cam, auto_rotate=False, auto_correct_rgb=False) is rgb
    assert set(s.get_all()) == set(StretchCameras.all())


NOTEBOOK_D435I_F = 399.427   # docs/getting_started.ipynb cell 14: cam_d435i_K at 640 x 480
NOTEBOOK_D405_F = 514.682


def notebook_pose(blob):
    """The state cell 14 was printed in: `StretchMujocoSimulator()` -- the default scene, models/scene.xml: floor, table, two
    objects -- `sim.start()` (qpos0, then home(): the home keyframe's servo targets, stretch_mujoco_simulator.py:136) and
    t = 3.2 s of simulated time."""
    from conftest import HOME_CTRL
    o = Oracle(blob)
    o.set_option("solver", 2)
    o.arr("ctrl")[:] = HOME_CTRL
    o.step(1601)
    return o


@pytest.fixture(scope="module")
def notebook_scene():
    import os
    from conftest import MODELS
    with open(os.path.join(MODELS, "stretch_scene.smjb"), "rb") as f:
        return notebook_pose(f.read())


NB_D405_BOTTOM = np.array([[0.445, 0.445, 0.445, 0.449, 0.449, 0.449],     # rows -3, -2, -1; columns 0, 1, 2, -3, -2, -1
                           [0.444, 0.444, 0.444, 0.448, 0.448, 0.448],
                           [0.442, 0.442, 0.442, 0.447, 0.447, 0.447]])
NB_D435_TOP, NB_D435_BOT = np.array([1.649, 1.643, 1.638]), np.array([1.647, 1.641, 1.636])
COLS = [0, 1, 2, -3, -2, -1]


def test_wrist_depth_image_against_the_values_printed_in_the_reference_notebook(notebook_scene):
    """docs/getting_started.ipynb cell 14 (640 x 480, cam_d405_K f = 514.682 -> fovy 50 deg): cam_d405_depth is 0 in its top rows
    (nothing within the 1 m limit) and reads 0.445 / 0.444 / 0.442 m in the first columns of its last three rows, 0.449 / 0.448 /
    0.447 m in the last columns -- the wrist camera looking at the TABLE of the default scene from 3.2 s after start().  Those
    numbers are MuJoCo's; the fp64 ray caster on the build's compiled scene reproduces them to the printed precision (+- 1 in the
    third decimal).  Pins, against MuJoCo: the wrist camera's pose through the whole arm chain, the table, "depth = metres along
    the optical axis", the pixel-centre convention, the 1 m limit."""
    o = notebook_scene
    d = o.render_depth(D405, 640, 480, 2 * math.degrees(math.atan(240 / NOTEBOOK_D405_F)), 1.0)
    assert np.all(d[:3][:, COLS] == 0)
    assert np.abs(d[-3:][:, COLS] - NB_D405_BOTTOM).max() < 1.6e-3


def test_head_depth_image_against_the_values_printed_in_the_reference_notebook(notebook_scene):
    """Same cell, cam_d435i_depth (f = 399.427 -> fovy 62 deg; the camera is mounted sideways): first columns 0, last three
    columns 1.649 / 1.643 / 1.638 m in the top rows and 1.647 / 1.641 / 1.636 m in the bottom rows -- the floor.  Reproduced to the
    printed precision from the position of `d435i_camera_rgb`; from `d435i_camera_depth` of today's stretch.xml, which sits 15 mm
    lower (stretch.xml:461-462: the colour camera has pos 0 0.015 0, the depth camera none), every value is 0.019 m smaller --
    the notebook was evidently recorded when both cameras shared the colour camera's position.  Both facts are asserted."""
    o = notebook_scene
    fovy = 2 * math.degrees(math.atan(240 / NOTEBOOK_D435I_F))
    at_rgb = o.render_depth(D435 - 1, 640, 480, fovy, 10.0)
    assert np.all(at_rgb[:3, :3] == 0) and np.all(at_rgb[-3:, :3] == 0)
    assert np.abs(at_rgb[:3, -3:] - NB_D435_TOP).max() < 1.6e-3 and np.abs(at_rgb[-3:, -3:] - NB_D435_BOT).max() < 1.6e-3
    at_depth = o.render_depth(D435, 640, 480, fovy, 10.0)
    assert np.abs((at_rgb[0, -3:] - at_depth[0, -3:]) - 0.019).max() < 1.5e-3
    # ray length instead of axial depth would read sqrt(1 + x^2 + y^2) times more: ruled out by a wide margin
    x, y = (639.5 - 320) / NOTEBOOK_D435I_F, (0.5 - 240) / NOTEBOOK_D435I_F
    assert at_rgb[0, -1] * math.sqrt(1 + x * x + y * y) > 1.25 * NB_D435_TOP[-1]


def test_rgb_stand_in_is_the_albedo_of_the_nearest_geom(posed):
    """smjo_render_geomid / smj_render_rgb: the id image agrees with the depth image about what is sky, the floor is the plane
    geom in its own colour, the colour image is the rgba table entry of the id, and the ids seen are camera-visible geoms."""
    m, o = posed
    fovy = 100.0     # the wrist camera: gripper, floor and sky
    depth = o.render_depth(D405, 53, 30, fovy, 0.0)
    gid, rgb = o.render_geomid(D405, 53, 30, fovy)
    zfar = float(m["vis_znear_zfar_extent"][1] * m["vis_znear_zfar_extent"][2])
    assert np.array_equal(gid < 0, depth >= zfar * (1 - 1e-9))
    rgba = np.asarray(m["geom_rgba"], float).reshape(-1, 4)
    seen = np.unique(gid[gid >= 0])
    assert len(seen) >= 2 and (rgba[seen, 3] > 0).all()
    plane = int(np.where(np.asarray(m["geom_type"]) == 0)[0][0])
    assert plane in seen
    want = np.where(gid[..., None] >= 0, (np.clip(rgba[np.maximum(gid, 0), :3], 0, 1) * 255 + 0.5).astype(np.uint8), np.array([169, 224, 255], np.uint8))
    assert np.array_equal(rgb, want)
    # the colour twin of the wrist depth camera sits at the same place (stretch.xml:383-384): the same image
    gid2, _ = o.render_geomid(D405 - 1, 53, 30, fovy)
    assert np.array_equal(gid2, gid)
