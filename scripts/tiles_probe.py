"""lln.geojson: visibility matrix and emitter screens with the painter's sectors off and with several sector sizes:
device times of the kernels, and how many entries differ from the faces painted whole."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from raytracing_amd import _lib, scenarios  # noqa: E402
from raytracing_amd import geometry as geom  # noqa: E402

place, tx, rx = scenarios.lln_grid(10)
polygons = place.get_polygons_list()
scene = _lib.Scene(geom.scene_arrays_of(polygons))
cube = geom.Cube.by_point_and_side_length(tx, 0.2)
screens = [(s.points, s.get_parametric(), -1) for s in cube.polygons]
base = None
settings = [(0, 0), (1024, 192), (1024, 384), (2048, 256), (4096, 512), (1024, 96)]
if len(sys.argv) > 1:
    settings = [(0, 0)] + [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for min_c, per in settings:
    scene.set_visibility_tiling(min_c, per)
    t0 = time.perf_counter()
    m, v = scene.visibility_with_faces(screens)
    wall = time.perf_counter() - t0
    tm = scene.visibility_timing()
    if base is None:
        base = (m, v)
    print(f"min {min_c:5d} per_tile {per:4d}: wall {wall:.3f} s  rows prepare {tm['rows']['prepare']:.0f} sort {tm['rows']['sort']:.0f} "
          f"paint {tm['rows']['paint']:.0f} ms | screens paint {tm['faces']['paint']:.0f} ms | differing: matrix "
          f"{int((m != base[0]).sum())}, screens {int((v != base[1]).sum())}", flush=True)
    # the screens alone (nothing beside them)
    t0 = time.perf_counter()
    v2 = scene.faces_visibility(screens)
    print(f"      screens alone: wall {time.perf_counter() - t0:.3f} s, paint {scene.visibility_timing()['faces']['paint']:.0f} ms, "
          f"equal {bool((v2 == base[1]).all())}", flush=True)
scene.close()
