"""How long single rows of the lln visibility matrix take (each painted alone by one workgroup)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import _lib, scenarios  # noqa: E402
from raytracing_amd import geometry as geom  # noqa: E402

place, tx, rx = scenarios.lln_grid(10)
polygons = place.get_polygons_list()
scene = _lib.Scene(geom.scene_arrays_of(polygons))
scene.visibility_rows([5])
nv = np.array([len(p.points) for p in polygons])
rng = np.random.default_rng(0)
ids = [0] + list(np.argsort(-nv)[:6]) + list(rng.choice(len(polygons), 24, replace=False))
for i in ids:
    t = time.perf_counter()
    row = scene.visibility_rows([int(i)])
    dt = time.perf_counter() - t
    print(f"row {int(i):5d} verts {nv[i]:3d} sees {int(row.sum()):5d}  {1e3 * dt:8.2f} ms", flush=True)
for n in (64, 256, 257, 1024):
    sel = rng.choice(len(polygons), n, replace=False)
    t = time.perf_counter()
    scene.visibility_rows(sel)
    print(f"{n} random rows together: {1e3 * (time.perf_counter() - t):8.2f} ms", flush=True)
scene.close()
