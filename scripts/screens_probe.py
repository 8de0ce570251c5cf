"""Times the painting of the emitter's six screens on a scene, one by one and together (not a test)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from raytracing_amd import _lib, geometry as geom, scenarios  # noqa: E402

place, tx, rx = scenarios.lln_grid(10) if len(sys.argv) < 2 or sys.argv[1] == "lln" else scenarios.levant_grid(8)
polygons = place.get_polygons_list()
scene = _lib.Scene(geom.scene_arrays_of(polygons))
cube = geom.Cube.by_point_and_side_length(tx, 0.2)
faces = [(s.points, s.get_parametric(), -1) for s in cube.polygons]
scene.faces_visibility(faces[:1], False)
for k, face in enumerate(faces):
    t = time.perf_counter()
    v = scene.faces_visibility([face], False)[0]
    print(f"screen {k}: {time.perf_counter() - t:.3f} s, sees {int(v.sum())} of {len(polygons)}", flush=True)
t = time.perf_counter()
vs = scene.faces_visibility(faces, False)
print(f"all six: {time.perf_counter() - t:.3f} s, union {int(np.logical_or.reduce(vs).sum())}")
