"""Reads the joint-angle plot that the reference keeps as a stored cell output in src/quadruped_model.ipynb (the only recorded
output of the real engine in the repository) and prints the statistics tests/test_oracle_physics.py::
test_envelope_of_the_reference_notebook_run quotes: per-curve min / max and the fastest sustained motion over ~0.1 s.
(The grey curve, knee_3, also collects anti-aliased pixels of other lines: its rate is an artefact.)
Runs only where the reference checkout is present (this container); nothing of the notebook is copied into the repo.
usage: python tools/digitize_notebook_plot.py [/root/reference/src/quadruped_model.ipynb]"""
import base64
import io
import json
import sys

import numpy as np
from PIL import Image

path = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src/quadruped_model.ipynb"
nb = json.load(open(path))
png = next(o["data"]["image/png"] for c in nb["cells"] for o in c.get("outputs", []) if "image/png" in o.get("data", {}))
im = np.array(Image.open(io.BytesIO(base64.b64decode(png))).convert("RGB")).astype(int)
# axes: horizontal gridlines at 1.5 .. -1.5 rad = rows 44 .. 484, vertical ones at 0 .. 10 s = columns 115 .. 961 (measured)
grid = np.abs(im - 176).sum(2) < 12
rows = [r for r in range(30, 493) if grid[r, 74:1003].sum() > 300]
cols = [c for c in range(73, 1004) if grid[31:492, c].sum() > 150]
y15, ym15, x0, x10 = min(rows), max(rows), min(cols), max(cols)
rad_per_px, px_per_s = 3.0 / (ym15 - y15), (x10 - x0) / 10.0
CURVES = {"ankle_1": (0x2c, 0xa0, 0x2c), "hip_2": (0xd6, 0x27, 0x28), "knee_2": (0x94, 0x67, 0xbd), "ankle_2": (0x8c, 0x56, 0x4b),
          "hip_3": (0xe3, 0x77, 0xc2), "knee_3": (0x7f, 0x7f, 0x7f), "ankle_3": (0xbc, 0xbd, 0x22), "hip_4": (0x17, 0xbe, 0xcf)}
print(f"calibration: {rad_per_px * 1e3:.1f} mrad / px, {1e3 / px_per_s:.1f} ms / px")
for name, c in CURVES.items():
    m = np.abs(im - np.array(c)).sum(2) < 25
    m[30:300, 830:1003] = False                              # legend
    tr = {}
    for x in range(x0, x10 + 1):
        ys = np.where(m[31:492, x])[0] + 31
        if len(ys) and len(ys) <= 16 and (np.diff(ys) <= 1).all():
            tr[x] = 1.5 - (ys.mean() - y15) * rad_per_px
    v = np.array(list(tr.values()))
    best = 0.0
    for x in tr:
        for w in (8, 9):
            mid = [tr[u] for u in range(x, x + w + 1) if u in tr]
            if x + w in tr and len(mid) >= w - 1 and ((np.diff(mid) >= -0.01).all() or (np.diff(mid) <= 0.01).all()):
                best = max(best, abs(tr[x + w] - tr[x]) / (w / px_per_s))
    print(f"{name:8s} visible {len(tr) / (x10 - x0 + 1):.0%}  min {v.min():+.3f}  max {v.max():+.3f}  fastest 0.1 s motion {best:.2f} rad/s")
