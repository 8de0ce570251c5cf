"""Order 4 on the lln map without diffraction (not a test): the deepest level is not stored and holds more than 2^31
chains.  python scripts/wide_index_probe.py [n_receivers]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from raytracing_amd import _lib, geometry as geom, scenarios  # noqa: E402

n_rx = int(sys.argv[1]) if len(sys.argv) > 1 else 16
place, tx, rx = scenarios.lln_grid(100)
rx = rx[:: max(1, len(rx) // n_rx)][:n_rx]
scene = _lib.Scene(geom.scene_arrays_of(place.get_polygons_list()))
cube = geom.Cube.by_point_and_side_length(tx, 0.2)
t = time.perf_counter()
vis, screens = scene.visibility_with_faces([(s.points, s.get_parametric(), -1) for s in cube.polygons])
ev = np.logical_or.reduce(screens)
print(f"visibility {time.perf_counter() - t:.2f} s", flush=True)
t = time.perf_counter()
cand = scene.enumerate(vis, ev, np.zeros((0, 2), np.int32), np.zeros((0, 2, 3)), 4, with_diffraction=False)
print(f"enumeration {time.perf_counter() - t:.2f} s, candidates per group {cand.counts.tolist()}, deepest level / 2^31 = "
      f"{cand.counts[3] / 2**31:.3f}", flush=True)
for rep in range(2):
    t = time.perf_counter()
    res = scene.solve_paths(cand, tx, rx)
    dt = time.perf_counter() - t
    print(f"solve {dt:.3f} s, {res['n_attempted']:.3e} path-solves, {res['n_attempted'] / dt:.3e}/s; paths per group "
          f"{np.bincount(res['group'] + 1, minlength=6).tolist()}, largest candidate index {int(res['cand'].max())}", flush=True)
wide = res["cand"] >= 2**31
print("paths with an index of 2^31 or more:", int(wide.sum()), flush=True)
