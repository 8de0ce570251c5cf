"""Timing probe of the field step on a coverage-map-sized path set (not a test):
python scripts/fields_probe.py place_saint_jean 71 16"""
import os
import sys
import time

import numpy as np
import torch

torch.cuda.init()
sys.path.insert(0, ".")
from raytracing_amd import _lib, geometry as geom, scenarios  # noqa: E402

name = sys.argv[1]
n_side = int(sys.argv[2]) if len(sys.argv) > 2 else 71
n_freq = int(sys.argv[3]) if len(sys.argv) > 3 else 16
path = f"data/{name}.geojson"
if not os.path.exists(path):
    path = f"data/{name}/{name}.geojson"
place = geom.load_place(path)
lo, hi = place.get_domain()
tx = scenarios.emitter_outside(place, (lo[0] + hi[0]) / 2, (lo[1] + hi[1]) / 2, 30.0)
rx = scenarios.lattice_receivers(place, n_side, 1.5)
polygons = place.get_polygons_list()
scene = _lib.Scene(geom.scene_arrays_of(polygons))
pairs, edges = scene.sharp_edges()
vis = scene.visibility_matrix(False)
cube = geom.Cube.by_point_and_side_length(tx, 0.2)
ev = np.logical_or.reduce(scene.faces_visibility([(s.points, s.get_parametric(), -1) for s in cube.polygons], False))
cand = scene.enumerate(vis, ev, pairs, edges, 2, with_diffraction=True)
res = scene.solve_paths(cand, tx, rx)
n = len(res["rx"])
print(name, "polygons", len(polygons), "rx", len(rx), "paths", n, flush=True)
t = time.perf_counter()
keep = res["group"] >= 0
kind = np.zeros(n, dtype=np.int32)
chain = np.zeros((n, _lib.RT_MAX_ORDER), dtype=np.int32)
epolys = np.zeros((n, 2), dtype=np.int32)
epts = np.zeros((n, 6))
k2, c2, p2, e2 = cand.describe(res["group"][keep], res["cand"][keep])
kind[keep], chain[keep], epolys[keep], epts[keep] = k2, c2, p2, e2
print("describe", f"{time.perf_counter()-t:.3f}s", flush=True)
freqs = np.linspace(3.5e9, 60e9, n_freq)
for rep in range(3):
    t = time.perf_counter()
    out = scene.fields(tx, rx, res["rx"], kind, chain, epolys, epts, res["points"], freqs)
    dt = time.perf_counter() - t
    print(f"fields (host arrays) {dt*1e3:.2f} ms  {n*n_freq/dt:.3e} path-frequencies/s  status ok {(out['status']==0).mean():.3f}",
          "finite dB receivers", int(np.isfinite(out["power_db"][:, 0]).sum()), flush=True)
