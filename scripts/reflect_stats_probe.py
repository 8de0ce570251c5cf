"""Work counters of the tiled reflection kernel on the lln coverage map (needs a build with -DRT_REFLECT_STATS)."""
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
cand = scene.enumerate(vis, ev, pairs, edges, 2, with_diffraction=False)
lib = _lib.load()
out = (ctypes.c_ulonglong * 4)()
lib.rt_debug_reflect_stats(out)
res = scene.solve_paths(cand, tx, rx)
lib.rt_debug_reflect_stats(out)
tests, kept, passed, undecided = [int(v) for v in out]
print(f"tile tests {tests:.3e}, kept {kept:.3e} ({kept / tests:.3%}), undecided {undecided:.3e}, receiver pre-tests passed "
      f"{passed:.3e} ({passed / max(kept * 16, 1):.3%} of the receivers of kept tiles), paths {len(res['rx'])}")
