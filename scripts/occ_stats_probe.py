"""Work counters of the grid obstruction kernel over one lln coverage-map solve (needs a build with -DRT_OCC_STATS)."""
import ctypes
import sys

import numpy as np
import torch

torch.cuda.init()
sys.path.insert(0, ".")
from raytracing_amd import _lib, geometry as geom, scenarios  # noqa: E402

place, tx, rx = scenarios.lln_grid(100)
polygons = place.get_polygons_list()
scene = _lib.Scene(geom.scene_arrays_of(polygons))
pairs, edges = scene.sharp_edges()
vis = scene.visibility_matrix(False)
cube = geom.Cube.by_point_and_side_length(tx, 0.2)
ev = np.logical_or.reduce(scene.faces_visibility([(s.points, s.get_parametric(), -1) for s in cube.polygons], False))
cand = scene.enumerate(vis, ev, pairs, edges, 2, with_diffraction=True)
lib = _lib.load()
out = (ctypes.c_ulonglong * 8)()
scene.solve_paths(cand, tx, rx)
lib.rt_debug_occlusion_stats(out)
scene.solve_paths(cand, tx, rx)
lib.rt_debug_occlusion_stats(out)
a, c, r, e, it, w, h = [int(v) for v in out][:7]
print(f"active lanes {a:.3e} in {w:.3e} waves ({a / w:.1f} per wave), cells {c / a:.2f}, records {r / a:.2f}, exact tests "
      f"{e / a:.2f} per active lane, hits {h / a:.3f}; wave trips {it / w:.1f} per wave -> lane efficiency "
      f"{r / (it * 64):.3f}")
