"""Painting time of a scene's visibility matrix over several paintings (not a test): python scripts/paint_times.py lln [n]"""
import sys

sys.path.insert(0, ".")
import numpy as np

from raytracing_amd import _lib, geometry as geom  # noqa: E402

name = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
scene = _lib.Scene(geom.load_scene(f"data/{name}.geojson").arrays)
scene.visibility_matrix(False)
ts = []
for rep in range(reps):
    scene.visibility_matrix(False)
    ts.append(scene.visibility_timing()["rows"]["paint"])
print(f"{name}: paint {np.mean(ts):.1f} ms mean, {np.min(ts):.1f} min, {np.max(ts):.1f} max over {reps}: " + " ".join(f"{t:.0f}" for t in ts), flush=True)
