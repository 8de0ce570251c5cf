"""How the obstruction walk's segments of the lln map split between reflection and diffraction paths (not a test):
the solve with and without the diffraction groups, segments walked and kernel times of each."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from raytracing_amd import _lib, scenarios  # noqa: E402
from raytracing_amd import device_pipeline as dp  # noqa: E402
from raytracing_amd import geometry as geom  # noqa: E402

place, tx, rx = scenarios.lln_grid(100)
scene = _lib.Scene(geom.scene_arrays_of(place.get_polygons_list()))
pairs, edges = scene.sharp_edges()
cube = geom.Cube.by_point_and_side_length(tx, 0.2)
vis, screens = scene.visibility_with_faces([(s.points, s.get_parametric(), -1) for s in cube.polygons])
d_rx = torch.tensor(rx, dtype=torch.float64, device="cuda:0")
scene.set_timing(True)
for with_d in (False, True):
    cand = scene.enumerate(vis, np.logical_or.reduce(screens), pairs, edges, 2, with_diffraction=with_d)
    pipe = dp.DevicePipeline(scene, cand, tx, capacity=40 * len(rx))
    for _ in range(4):
        block = pipe.run(d_rx, [3.6e9], fields=False)
    tm = block.solve_timing
    print("diffraction" if with_d else "reflections only", "paths", block.n_paths, "segments walked", tm["segments"],
          {k: round(float(v), 2) for k, v in tm["ms"].items()}, flush=True)
    cand.close()
