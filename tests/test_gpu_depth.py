"""Depth cameras on the HIP path (smj_render_depth through the C-ABI) against the fp64 ray-casting restatement on the same
poses.  Tolerance: a pixel agrees when |dz| <= 1e-4 + 1e-4 * z (fp32 ray/triangle arithmetic at <= 10 m); rays that graze a
silhouette or a shared triangle edge may fall on different sides in fp32 and fp64, so at most 0.5 % of the pixels may
disagree.  [B, H, W] layout, limits and the API conventions are checked exactly."""
import numpy as np
import pytest
import torch

from conftest import home_qpos
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu


def _sim(B, cams):
    from stretch_mujoco_amd import StretchBatchSimulator

    sim = StretchBatchSimulator(num_envs=B, device="cuda:0", cameras_to_use=cams, solver="newton")
    sim.start(home=False)
    return sim


def _agree(gpu, ref):
    ok = np.abs(gpu - ref) <= 1e-4 + 1e-4 * np.abs(ref)
    return ok, 1.0 - ok.mean()


def test_depth_images_match_oracle_per_env():
    from stretch_mujoco_amd.enums import StretchCameras

    cams = StretchCameras.depth()
    sim = _sim(3, cams)
    o = Oracle(sim._blob)
    q0 = home_qpos(o.arr("qpos").copy())
    # three different poses: home; head turned to look at the arm, wrist bent; base moved and yawed, lift low
    poses = [q0.copy(), q0.copy(), q0.copy()]
    names = {n: i for i, n in enumerate(__import__("json").loads(bytes(sim.model["names_json"]).decode())["joint"])}
    adr = sim.model["jnt_qposadr"]
    poses[1][adr[names["joint_head_pan"]]] = -1.2
    poses[1][adr[names["joint_head_tilt"]]] = -0.8
    poses[1][adr[names["joint_wrist_pitch"]]] = -0.6
    poses[2][0:2] = [0.7, -0.4]
    poses[2][3:7] = [np.cos(0.4), 0, 0, np.sin(0.4)]
    poses[2][adr[names["joint_lift"]]] = 0.35
    poses[2][adr[names["joint_head_tilt"]]] = -1.0
    sim.qpos[:] = torch.tensor(np.stack(poses, 1), dtype=torch.float32, device=sim.device)
    sim.step(1)                       # xpose <- kinematics of the state just written
    imgs = sim.pull_camera_data()
    torch.cuda.synchronize()
    cam_names = __import__("json").loads(bytes(sim.model["names_json"]).decode())["camera"]
    worst = 0.0
    for cam in cams:
        st = cam.initial_camera_settings
        g = getattr(imgs, cam.name)
        assert g.shape == (3, st.height, st.width) and g.dtype == torch.float32
        g = g.cpu().numpy()
        for e in range(3):
            o.arr("qpos")[:] = np.asarray(sim_q(poses[e]))
            o.forward()
            ref = o.render_depth(cam_names.index(cam.camera_name_in_mjcf), st.width, st.height,
                                 st.field_of_view_vertical_in_degrees, cam.depth_limit)
            ok, bad = _agree(g[e], ref)
            worst = max(worst, bad)
            assert bad < 5e-3, (cam, e, bad)
            assert (g[e] <= cam.depth_limit).all() and (g[e] >= 0).all()
            assert ((g[e] > 0) & (ref > 0)).mean() > 0.02      # something is in range in every view
    print("worst disagreeing pixel fraction", worst)
    assert imgs.cam_d405_K.shape == (3, 3) and imgs.cam_d435i_K[0, 2] == 960


def test_depth_in_the_kitchen_at_robocasa_scale_matches_the_oracle():
    """Config 5's scene: both depth cameras in the generated kitchen at Robocasa scale (123 camera-visible geoms, 72 meshes; the
    satellite build runs the physics), settled state after 50 steps, against the fp64 ray caster on the same pose."""
    from stretch_mujoco_amd import StretchBatchSimulator
    from stretch_mujoco_amd.enums import StretchCameras

    cams = StretchCameras.depth()
    sim = StretchBatchSimulator(num_envs=2, device="cuda:0", cameras_to_use=cams, solver="newton", scene="stretch_kitchen_robocasa")
    sim.start(home=False)
    sim.step(50)
    q = sim.qpos[:, 0].cpu().numpy().astype(np.float64)
    sim.step(1)   # xpose <- kinematics of q
    imgs = sim.pull_camera_data()
    torch.cuda.synchronize()
    o = Oracle(sim._blob)
    o.arr("qpos")[:] = q
    o.forward()
    cam_names = __import__("json").loads(bytes(sim.model["names_json"]).decode())["camera"]
    for cam in cams:
        st = cam.initial_camera_settings
        g = getattr(imgs, cam.name)[0].cpu().numpy()
        ref = o.render_depth(cam_names.index(cam.camera_name_in_mjcf), st.width, st.height, st.field_of_view_vertical_in_degrees, cam.depth_limit)
        ok, bad = _agree(g, ref)
        print(cam.name, "disagreeing pixel fraction", bad, "pixels in range", float((ref > 0).mean()))
        assert bad < 5e-3, (cam, bad)
        assert (ref > 0).mean() > 0.02
    sim.stop()


def sim_q(q):
    """fp32 round trip: the oracle must see the pose the GPU saw."""
    return np.asarray(q, np.float32).astype(np.float64)


def test_raw_render_and_errors():
    import ctypes

    from stretch_mujoco_amd import lib
    from stretch_mujoco_amd.enums import StretchCameras

    sim = _sim(2, [StretchCameras.cam_d435i_depth])
    sim.pull_camera_data()        # before any step: garbage poses must not poison the cached camera-static layer
    sim.qpos[:] = torch.tensor(sim_q(home_qpos(sim.model["qpos0"])), dtype=torch.float32, device=sim.device).unsqueeze(1)
    sim.step(1)
    L = lib.load()
    raw = torch.zeros(2, 60, 106, dtype=torch.float32, device=sim.device)
    rc = L.smj_render_depth(sim._ctx, 3, 106, 60, 42.0, 0.0, ctypes.c_void_p(raw.data_ptr()), sim._stream())
    assert rc == 0
    torch.cuda.synchronize()
    zfar = float(sim.model["vis_znear_zfar_extent"][1] * sim.model["vis_znear_zfar_extent"][2])
    r = raw.cpu().numpy()
    assert np.isclose(r.max(), zfar, rtol=1e-6) and (r == r.max()).mean() > 0.1     # sky = far plane in the raw render
    assert torch.equal(raw[0], raw[1])                                              # identical envs, identical images
    o = Oracle(sim._blob)
    o.arr("qpos")[:] = sim_q(home_qpos(sim.model["qpos0"])); o.forward()
    ref = o.render_depth(3, 106, 60, 42.0, 0.0)
    assert (np.abs(r[0] - ref) <= 1e-4 + 1e-4 * np.abs(ref)).mean() > 0.995
    assert L.smj_render_depth(sim._ctx, 9, 106, 60, 42.0, 0.0, ctypes.c_void_p(raw.data_ptr()), sim._stream()) != 0
    assert b"camera id" in L.smj_last_error(sim._ctx)
    assert L.smj_render_depth(sim._ctx, 3, 106, 60, 42.0, 0.0, None, sim._stream()) != 0


def test_depth_full_batch_properties():
    """4096 envs: per-env images depend only on that env's pose (two envs with equal qpos give equal images, a yawed base
    leaves the wrist camera's view of the gripper unchanged)."""
    from stretch_mujoco_amd.enums import StretchCameras

    B = 4096
    sim = _sim(B, [StretchCameras.cam_d405_depth])
    q = torch.tensor(sim_q(home_qpos(sim.model["qpos0"])), dtype=torch.float32, device=sim.device).unsqueeze(1).repeat(1, B)
    yaw = torch.linspace(-3.0, 3.0, B, device=sim.device)
    q[3] = torch.cos(yaw / 2); q[6] = torch.sin(yaw / 2)
    q[0] = torch.linspace(-5, 5, B, device=sim.device)
    sim.qpos[:] = q
    sim.step(1)
    img = sim.pull_camera_data().cam_d405_depth
    torch.cuda.synchronize()
    assert img.shape == (B, 270, 480)
    near = (img > 0) & (img < 0.3)                       # the gripper in front of the wrist camera
    cnt = near.sum(dim=(1, 2)).float()
    assert cnt.min() > 1000 and (cnt.max() - cnt.min()) / cnt.mean() < 0.02
    ref = img[0]
    d = (img - ref).abs()
    assert (d[near & near[0:1]] < 2e-4).float().mean() > 0.995


def test_culling_does_not_change_the_image():
    """The staging pass drops geoms per env (near / far / layer) and the tile kernel per screen rectangle: both are superset tests,
    so the image with the culling switched off (SMJ_DEPTH_NOCULL=1, read once per process -> child process) is the same, bit
    for bit.  64 envs at random poses in the kitchen stand-in, both cameras (tools/gpu_depth_cullcheck.py)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_depth_cullcheck.py"), "64", "stretch_kitchen_standin"],
                         cwd=root, capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if "differing" in l]
    assert out.returncode == 0 and len(lines) == 2, out.stdout + out.stderr
    assert all(l.rstrip().endswith("differing 0") for l in lines), out.stdout


def test_rasterised_meshes_equal_the_ray_cast_meshes():
    """The meshes and box fixtures go into the depth image by rasterisation (smj_meshlet_kernel: one wave per meshlet, atomicMin)
    and the per-pixel kernel resolves the other primitives against it; option depth_raster = 0 casts a ray per pixel through the
    mesh BVHs instead (the round-2 path).  Same pixel-centre sampling, front faces only, depth along the optical axis: the two
    images agree up to fp32 rounding (the rasteriser interpolates 1 / depth over the screen triangle, and takes a box as twelve
    triangles instead of the ray / box closed form, 1-ulp reciprocals) -- within 1e-4 relative on all but the odd silhouette pixel
    (< 2 in 10^5).  64 envs at random poses, kitchen stand-in, both cameras."""
    from stretch_mujoco_amd import StretchBatchSimulator
    from stretch_mujoco_amd.enums import StretchCameras

    B = 64
    sim = StretchBatchSimulator(num_envs=B, device="cuda:0", cameras_to_use=StretchCameras.depth(), scene="stretch_kitchen_standin")
    sim.start(home=True)
    g = torch.Generator(device=sim.device).manual_seed(99)
    lo = torch.tensor(np.asarray(sim.model["actuator_ctrlrange"], np.float32)[:, 0], device=sim.device)
    hi = torch.tensor(np.asarray(sim.model["actuator_ctrlrange"], np.float32)[:, 1], device=sim.device)
    sim.ctrl[:] = lo[:, None] + (hi - lo)[:, None] * torch.rand(sim.nu, B, generator=g, device=sim.device)
    sim.step(300)
    a = {c.name: getattr(sim.pull_camera_data(), c.name).clone() for c in StretchCameras.depth()}
    sim.set_option("depth_raster", 0)
    b = {c.name: getattr(sim.pull_camera_data(), c.name).clone() for c in StretchCameras.depth()}
    torch.cuda.synchronize()
    for k in a:
        same = (a[k] == b[k]).float().mean().item()
        close = ((a[k] - b[k]).abs() <= 1e-4 * b[k].abs()).float().mean().item()
        assert close > 0.99998, (k, same, close)
        assert float((a[k] > 0).float().mean()) > 0.3
    sim.stop()


def test_notebook_corner_values_through_the_raw_entry():
    """docs/getting_started.ipynb cell 14, 3.2 s after start() in the default scene (640 x 480): cam_d405_depth reads 0.445 / 0.444
    / 0.442 m in the first and 0.449 / 0.448 / 0.447 m in the last columns of its last three rows (the table), 0 in its top rows;
    cam_d435i_depth reads 1.649 / 1.643 / 1.638 m (top rows) and 1.647 / 1.641 / 1.636 m (bottom rows) in its last three columns
    from the colour camera's position (tests/test_depth_oracle.py has the story of the 15 mm).  MuJoCo's printed numbers through
    start() / step / smj_render_depth(width, height, fovy): +- 2 in the third decimal."""
    import ctypes
    import math

    from stretch_mujoco_amd import StretchBatchSimulator, lib
    from stretch_mujoco_amd.enums import StretchCameras

    sim = StretchBatchSimulator(num_envs=2, device="cuda:0", cameras_to_use=StretchCameras.depth(), solver="newton", scene="stretch_scene")
    sim.start(home=False)
    sim.home(settle=False)            # the reference's start(): home keyframe targets
    sim.step(1601)
    L = lib.load()
    img = torch.zeros(2, 480, 640, dtype=torch.float32, device=sim.device)
    cols = [0, 1, 2, -3, -2, -1]
    assert L.smj_render_depth(sim._ctx, 1, 640, 480, 2 * math.degrees(math.atan(240 / 514.682)), 1.0, ctypes.c_void_p(img.data_ptr()), sim._stream()) == 0
    torch.cuda.synchronize()
    d = img[0].cpu().numpy()
    want = np.array([[0.445, 0.445, 0.445, 0.449, 0.449, 0.449], [0.444, 0.444, 0.444, 0.448, 0.448, 0.448], [0.442, 0.442, 0.442, 0.447, 0.447, 0.447]])
    assert np.all(d[:3][:, cols] == 0) and np.abs(d[-3:][:, cols] - want).max() < 2.1e-3
    assert L.smj_render_depth(sim._ctx, 2, 640, 480, 2 * math.degrees(math.atan(240 / 399.427)), 10.0, ctypes.c_void_p(img.data_ptr()), sim._stream()) == 0
    torch.cuda.synchronize()
    d = img[0].cpu().numpy()
    assert np.all(d[:3, :3] == 0) and np.all(d[-3:, :3] == 0)
    assert np.abs(d[:3, -3:] - np.array([1.649, 1.643, 1.638])).max() < 2.1e-3 and np.abs(d[-3:, -3:] - np.array([1.647, 1.641, 1.636])).max() < 2.1e-3
    sim.stop()


def test_rgb_stand_in_matches_the_oracle_ray_caster():
    """RGB cameras (smj_render_rgb): per pixel the unlit albedo of the first geom the ray meets -- a stand-in for MuJoCo's
    OpenGL image, defined in oracle/smj_oracle.c smjo_render_geomid.  Geom ids through the raw entry and colours through
    pull_camera_data() against the fp64 ray caster at three poses, all three colour cameras at the reference's resolutions:
    ids equal on >= 99.5 % of the pixels (fp32 / fp64 differ on silhouette pixels, as for depth), colours = the table entry of
    the id on every pixel, shape / dtype [B, H, W, 3] uint8."""
    import ctypes

    from stretch_mujoco_amd import lib
    from stretch_mujoco_amd.enums import StretchCameras

    cams = StretchCameras.rgb()
    sim = _sim(3, StretchCameras.all())
    o = Oracle(sim._blob)
    q0 = home_qpos(o.arr("qpos").copy())
    poses = [q0.copy(), q0.copy(), q0.copy()]
    names = {n: i for i, n in enumerate(__import__("json").loads(bytes(sim.model["names_json"]).decode())["joint"])}
    adr = sim.model["jnt_qposadr"]
    poses[1][adr[names["joint_head_pan"]]] = -1.2
    poses[1][adr[names["joint_head_tilt"]]] = -0.8
    poses[1][adr[names["joint_wrist_pitch"]]] = -0.6
    poses[2][0:2] = [0.7, -0.4]
    poses[2][3:7] = [np.cos(0.4), 0, 0, np.sin(0.4)]
    poses[2][adr[names["joint_lift"]]] = 0.35
    sim.qpos[:] = torch.tensor(np.stack(poses, 1), dtype=torch.float32, device=sim.device)
    sim.step(1)
    imgs = sim.pull_camera_data()
    torch.cuda.synchronize()
    cam_names = __import__("json").loads(bytes(sim.model["names_json"]).decode())["camera"]
    table = np.clip(np.asarray(sim.model["geom_rgba"], float).reshape(-1, 4)[:, :3], 0, 1)
    L = lib.load()
    for cam in cams:
        st = cam.initial_camera_settings
        got = getattr(imgs, cam.name)
        assert got.dtype == torch.uint8 and tuple(got.shape) == (3, st.height, st.width, 3)
        gid = torch.full((3, st.height, st.width), -7, dtype=torch.int32, device=sim.device)
        rgb = torch.zeros(3, st.height, st.width, 3, dtype=torch.uint8, device=sim.device)
        ci = cam_names.index(cam.camera_name_in_mjcf)
        assert L.smj_render_rgb(sim._ctx, ci, st.width, st.height, float(st.field_of_view_vertical_in_degrees),
                                ctypes.c_void_p(rgb.data_ptr()), ctypes.c_void_p(gid.data_ptr()), sim._stream()) == 0
        torch.cuda.synchronize()
        assert torch.equal(rgb, got)
        g, c = gid.cpu().numpy(), rgb.cpu().numpy()
        want = np.where(g[..., None] >= 0, (table[np.maximum(g, 0)] * 255 + 0.5).astype(np.uint8), np.array([169, 224, 255], np.uint8))
        assert np.array_equal(c, want)
        for e in range(3):
            o.arr("qpos")[:] = poses[e]; o.forward()
            og, oc = o.render_geomid(ci, st.width, st.height, st.field_of_view_vertical_in_degrees)
            assert (og == g[e]).mean() >= 0.995, (cam, e, (og == g[e]).mean())
            assert (oc == c[e]).all(-1).mean() >= 0.995
    assert L.smj_render_rgb(sim._ctx, 9, 8, 8, 42.0, ctypes.c_void_p(rgb.data_ptr()), None, sim._stream()) != 0
    sim.stop()


def test_notebook_images_of_cells_15_and_23_through_the_raw_entries():
    """The camera images the reference's notebook stores (tests/golden/notebook_images.npz; tests/test_notebook_images.py has the story
    and the oracle's numbers) against the HIP renderer: start() / home / 1601 steps in the default scene, the base placed at the pose
    cell 20 prints, then (i) cam_d405_depth through smj_render_depth(640, 480, fovy 50): MuJoCo's 53 400 stored pixels to 0.3 grey
    levels on average, fingers' silhouette IoU > 0.975, table-top depths within 1 level; the yaw scan has its minimum at the printed
    -0.065 rad; (ii) after the head has run into its tilt stop, cam_nav_rgb's geom ids through smj_render_rgb: the table's outline
    against the wood MuJoCo drew (IoU > 0.94), the red cylinder and the blue box where it drew them."""
    import ctypes

    import notebook_images as nbi
    from scipy.ndimage import binary_dilation, binary_erosion
    from stretch_mujoco_amd import StretchBatchSimulator, lib
    from stretch_mujoco_amd.enums import StretchCameras

    pytest.importorskip("PIL")
    G = nbi.golden()
    yaws = np.round(np.arange(-0.105, 0.016, 0.01), 3)
    B = len(yaws)
    sim = StretchBatchSimulator(num_envs=B, device="cuda:0", cameras_to_use=StretchCameras.all(), solver="newton", scene="stretch_scene")
    sim.start(home=False)
    sim.home(settle=False)
    sim.step(1601)
    L = lib.load()
    x, y, th = nbi.NB_BASE_POSE

    def place(thetas):
        q = sim.qpos.t().cpu().numpy().astype(np.float64)
        sim.qpos[:] = torch.tensor(np.stack([nbi.place_base(q[e], x, y, thetas[e]) for e in range(B)], 1), dtype=torch.float32, device=sim.device)
        sim.qvel[:6] = 0
        sim.step(1)

    place(yaws)
    img = torch.zeros(B, 480, 640, dtype=torch.float32, device=sim.device)
    assert L.smj_render_depth(sim._ctx, nbi.CAM_D405_DEPTH, 640, 480, nbi.fovy_of(nbi.NB_F_D405), 1.0, ctypes.c_void_p(img.data_ptr()), sim._stream()) == 0
    torch.cuda.synchronize()
    g = G["cell15_cam_d405_depth"].astype(int)
    shown = [nbi.display(img[e].cpu().numpy(), (267, 200)).astype(int) for e in range(B)]
    cost = np.array([np.abs(r - g).mean() for r in shown])
    print(dict(zip(yaws.tolist(), np.round(cost, 3).tolist())))
    k = int(np.argmin(cost))
    assert abs(yaws[k] - th) < 1e-9 and cost[k] < 0.45 and np.sort(cost)[1] > 1.3 * cost[k]
    r = shown[k]
    diff = np.abs(r - g)
    assert (diff <= 2).mean() > 0.988 and ((r > 0) != (g > 0)).sum() < 90
    assert nbi.iou((g > 0) & (g < 80), (r > 0) & (r < 80)) > 0.975
    top = binary_erosion((g > 100) & (r > 100), iterations=3)
    assert top.sum() > 10000 and diff[top].mean() < 0.6 and np.percentile(diff[top], 99.5) <= 2
    # cell 23: head into its tilt stop, every env at the printed pose
    sim.move_to("head_tilt", -2.0)
    sim.step(2500)
    assert abs(float(sim.pull_status().head_tilt.pos[0]) - (-1.522573472981672)) < 2e-5
    place(np.full(B, th))
    table, blue, red = nbi.scene_geoms(sim.model)
    cls = nbi.colour_classes(G["cell23_cam_nav_rgb"])
    gid = torch.full((B, 533, 400), -7, dtype=torch.int32, device=sim.device)
    rgb = torch.zeros(B, 533, 400, 3, dtype=torch.uint8, device=sim.device)
    fits = {}
    for fv in (64.0, 69.0, 74.0):
        assert L.smj_render_rgb(sim._ctx, nbi.CAM_NAV, 400, 533, nbi.nav_display_fovy(fv), ctypes.c_void_p(rgb.data_ptr()),
                                ctypes.c_void_p(gid.data_ptr()), sim._stream()) == 0
        torch.cuda.synchronize()
        ids = np.rot90(gid[0].cpu().numpy(), 1)
        fits[fv] = nbi.iou(ids == table, cls["wood"])
        if fv == 69.0:
            keep = ids
    print(fits)
    assert fits[69.0] > 0.94 and fits[64.0] < 0.9 and fits[74.0] < 0.9

    def centroid(mask):
        ys, xs = np.nonzero(mask)
        return np.array([xs.mean(), ys.mean()])

    for mask, gold in ((keep == red, cls["red"]), (keep == blue, cls["blue"])):
        assert gold.sum() > 150 and mask.sum() > 150
        assert (gold & binary_dilation(mask, iterations=3)).sum() / gold.sum() > 0.9
        assert np.abs(centroid(mask & binary_dilation(gold, iterations=6)) - centroid(gold)).max() < 6.0
    sim.stop()
