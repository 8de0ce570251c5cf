import sys, time
sys.path.insert(0, ".")
import numpy as np
from raytracing_amd import _lib, geometry as geom, scenarios
loaded, tx = scenarios.lln_scene()
scene = _lib.Scene(loaded.arrays)
scene.set_timing(True)
cube = geom.Cube.by_point_and_side_length(np.asarray(tx, dtype=float).reshape(3), 0.2)
screens = [(s.points, s.get_parametric(), -1) for s in cube.polygons]
ref = scene.faces_visibility(screens, False)
for per in (24, 32, 48, 64, 96, 128, 192):
    scene.set_visibility_tiling(512, per)
    for rep in range(2):
        t = time.perf_counter(); v = scene.faces_visibility(screens, False); dt = time.perf_counter() - t
    assert np.array_equal(v, ref)
    print(per, f"{1e3*dt:.1f} ms", scene.visibility_timing()['faces'], flush=True)
