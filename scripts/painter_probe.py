"""Time of single rows of the visibility matrix (not a test): python scripts/painter_probe.py lln"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from raytracing_amd import _lib, geometry as geom  # noqa: E402

name = sys.argv[1]
path = f"data/{name}.geojson"
if not os.path.exists(path):
    path = f"data/{name}/{name}.geojson"
place = geom.load_place(path)
polygons = place.get_polygons_list()
scene = _lib.Scene(geom.scene_arrays_of(polygons))
scene.faces_visibility([(polygons[1].points, polygons[1].get_parametric(), 1)], False)  # warm-up
n = len(polygons)
for i in [0, 1, 2, 3, n // 3, n // 2, n - 1]:
    t = time.perf_counter()
    v = scene.faces_visibility([(polygons[i].points, polygons[i].get_parametric(), i)], False)
    print(f"row {i}: {1e3*(time.perf_counter()-t):8.2f} ms, sees {int(v.sum())}, vertices {len(polygons[i].points)}", flush=True)
import ctypes
def stamps(label):
    st = (ctypes.c_ulonglong * 8)()
    _lib.load().rt_debug_visibility_stats(st)
    c = max(1, st[3])
    if hasattr(_lib.load(), "rt_debug_visibility_phases"):
        ph = (ctypes.c_ulonglong * 16)()
        _lib.load().rt_debug_visibility_phases(ph)
        names = ["samples", "inside-one", "B edges", "collect", "canvas edges"]
        print("   phases (us per visit, visits): " + ", ".join(f"{nm} {ph[2*i]/max(1,ph[2*i+1])/100:.1f} x{ph[2*i+1]}" for i, nm in enumerate(names)), flush=True)
    print(f"   {label}: candidates {st[3]}, visible {st[4]}; per candidate us: setup {st[0]/c/100:.2f} decide {st[1]/c/100:.2f} tail {st[2]/c/100:.2f}", flush=True)
stamps("single rows")
for lo_, hi_ in [(0, 1), (1, n), (0, n)]:
    faces = [(polygons[i].points, polygons[i].get_parametric(), i) for i in range(lo_, hi_)]
    t = time.perf_counter()
    v = scene.faces_visibility(faces[:65535], False)
    print(f"rows [{lo_},{min(hi_, lo_ + 65535)}): {1e3*(time.perf_counter()-t):8.2f} ms", flush=True)
stamps("batches")
from raytracing_amd import scenarios  # noqa: E402
lo, hi = place.get_domain()
tx = scenarios.emitter_outside(place, (lo[0] + hi[0]) / 2, (lo[1] + hi[1]) / 2, 30.0)
cube = geom.Cube.by_point_and_side_length(tx, 0.2)
for k, sface in enumerate(cube.polygons):
    t = time.perf_counter()
    v = scene.faces_visibility([(sface.points, sface.get_parametric(), -1)], False)
    print(f"screen {k}: {1e3*(time.perf_counter()-t):8.2f} ms, sees {int(v.sum())}", flush=True)
    stamps(f"screen {k}")
