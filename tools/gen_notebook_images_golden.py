"""Golden vectors from the reference's notebook: the camera IMAGES it keeps as outputs.

docs/getting_started.ipynb cell 15 shows all five camera images of `pull_camera_data()` 3.2 s after `start()` in the default scene
(`media.show_images(images, vmin=0.0, vmax=1.0, border=True, height=200)`: 640 x 480 frames, the two depth maps mapped 0..1 m ->
grey, every frame resampled to a height of 200 and stored as an 8-bit RGB PNG); cell 23 shows `cam_nav_rgb` after
`move_to('head_tilt', -2.0)` ran into the joint's stop (`media.show_image(..., height=400)`: 533 x 400).  Those PNGs are MuJoCo
output: tens of thousands of reference pixels.  This script decodes them and writes the pixel arrays, nothing else, to
tests/golden/notebook_images.npz.

Run in the build container (needs PIL and /root/reference); the tests read the .npz only.
"""
import base64
import io
import json
import os
import re

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NB = "/root/reference/docs/getting_started.ipynb"


def pngs(cell):
    html = next("".join(o["data"]["text/html"]) for o in cell["outputs"] if "data" in o and "text/html" in o["data"])
    names = re.findall(r"<div>(cam_\w+)</div>", html)
    data = re.findall(r'src="data:image/png;base64,([^"]+)"', html)
    return names, [np.array(Image.open(io.BytesIO(base64.b64decode(b))).convert("RGB")) for b in data]


def main():
    nb = json.load(open(NB))
    code = [c for c in nb["cells"] if c["cell_type"] == "code"]
    c15 = next(c for c in code if "media.show_images(images" in "".join(c["source"]))
    c23 = next(c for c in code if "move_to('head_tilt', -2.0)" in "".join(c["source"]))
    out = {}
    names, imgs = pngs(c15)
    assert names == ["cam_d405_rgb", "cam_d405_depth", "cam_d435i_rgb", "cam_d435i_depth", "cam_nav_rgb"], names
    for n, im in zip(names, imgs):
        if n.endswith("depth"):
            assert (im[..., 0] == im[..., 1]).all() and (im[..., 0] == im[..., 2]).all()
            im = im[..., 0]
        out["cell15_" + n] = im
    _, imgs = pngs(c23)
    out["cell23_cam_nav_rgb"] = imgs[0]
    path = os.path.join(ROOT, "tests", "golden", "notebook_images.npz")
    np.savez_compressed(path, **out)
    for k, v in out.items():
        print(k, v.shape, v.dtype)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
