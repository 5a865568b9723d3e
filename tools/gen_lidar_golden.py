"""Golden vector from the reference's notebook: the lidar scan of docs/getting_started.ipynb cell 18, which the notebook keeps
only as a FIGURE (image/png output of `plt.scatter(x, y, color="r", s=5)` with x = -scan cos(i deg), y = -scan sin(i deg),
scan < 0 set to 0, axes limits +-(max + 1)).  This script decodes the stored PNG and writes what can be read off it as data:

  * the axes box in pixels and its data limits (max |x| = max |y| = 10 = the rangefinder cutoff -> limits +-11),
  * the red pixels, run-length encoded (the measured scan as drawn, 1 degree = 3.9 px at r = 10 m),
  * the ray indices whose point at r = cutoff is drawn (a red pixel within 1.6 px of (-10 cos i, -10 sin i)).

Run in the build container (needs PIL and /root/reference); the tests read tests/golden/lidar_figure.json only.
"""
import base64
import io
import json
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NB = "/root/reference/docs/getting_started.ipynb"


def main():
    nb = json.load(open(NB))
    cell = next(c for c in nb["cells"] if c["cell_type"] == "code" and "StretchSensors.base_lidar" in "".join(c["source"]))
    png = next(o["data"]["image/png"] for o in cell["outputs"] if "data" in o and "image/png" in o["data"])
    im = np.array(Image.open(io.BytesIO(base64.b64decode(png))).convert("RGB")).astype(int)
    H, W, _ = im.shape
    dark = im.sum(2) < 150
    cols = np.where(dark.sum(0) > 0.5 * H)[0]
    rows = np.where(dark.sum(1) > 0.5 * W)[0]
    x0, x1, y0, y1 = int(cols.min()), int(cols.max()), int(rows.min()), int(rows.max())   # the axes frame
    red = (im[:, :, 0] > 180) & (im[:, :, 1] < 100) & (im[:, :, 2] < 100)
    runs = []   # (row, first col, length)
    for r in range(H):
        c = np.nonzero(red[r])[0]
        if len(c) == 0:
            continue
        start = prev = int(c[0])
        for v in c[1:]:
            v = int(v)
            if v != prev + 1:
                runs.append([r, start, prev - start + 1]); start = v
            prev = v
        runs.append([r, start, prev - start + 1])
    ys, xs = np.nonzero(red)
    lim = 11.0
    at_cutoff = []
    for i in range(360):
        a = np.radians(i)
        px = x0 + (-10 * np.cos(a) + lim) / (2 * lim) * (x1 - x0)
        py = y1 - (-10 * np.sin(a) + lim) / (2 * lim) * (y1 - y0)
        if np.sqrt((xs - px) ** 2 + (ys - py) ** 2).min() < 1.6:
            at_cutoff.append(i)
    out = dict(source="docs/getting_started.ipynb cell 18, image/png output", width=W, height=H, axes_px=[x0, x1, y0, y1], lim=lim,
               cutoff=10.0, red_runs=runs, rays_at_cutoff=at_cutoff, n_red=int(red.sum()))
    path = os.path.join(ROOT, "tests", "golden", "lidar_figure.json")
    json.dump(out, open(path, "w"), separators=(",", ":"))
    print(f"{path}: {len(runs)} runs, {int(red.sum())} red pixels, rays at cutoff {at_cutoff[0]}..{at_cutoff[-1]} ({len(at_cutoff)})")


if __name__ == "__main__":
    sys.exit(main())
