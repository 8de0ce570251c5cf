"""Painting time of a scene's whole matrix under forced tile settings (not a test):
python scripts/tiling_sweep.py lln|ny8 min,size [min,size ...]   (-1,0 = the library's own rule, always run first)"""
import sys

sys.path.insert(0, ".")
import numpy as np

from raytracing_amd import _lib, geometry as geom  # noqa: E402

name = sys.argv[1]
if name == "ny8":
    from raytracing_amd import scenarios

    scene = _lib.Scene(geom.scene_arrays_of(scenarios.ny_tiled(8)[0].get_polygons_list()))
else:
    scene = _lib.Scene(geom.load_scene(f"data/{name}.geojson").arrays)
want = scene.visibility_matrix(False).copy()
for setting in [(-1, 0)] + [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]]:
    scene.set_visibility_tiling(*setting)
    ts = []
    for rep in range(4):
        got = scene.visibility_matrix(False)
        ts.append(scene.visibility_timing()["rows"]["paint"])
    print(f"{name} {setting}: paint {np.mean(ts):.1f} ms mean, {np.min(ts):.1f} min; same matrix {bool((got == want).all())}", flush=True)
