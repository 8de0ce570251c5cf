"""SHA-256 of a scene's visibility matrix, to hold two builds of the library against each other (not a test):
python scripts/matrix_hash.py lln|ny8|<file under data/ without .geojson> ..."""
import hashlib
import sys

sys.path.insert(0, ".")
import numpy as np

from raytracing_amd import _lib, geometry as geom  # noqa: E402

for name in sys.argv[1:]:
    if name == "ny8":
        from raytracing_amd import scenarios

        scene = _lib.Scene(geom.scene_arrays_of(scenarios.ny_tiled(8)[0].get_polygons_list()))
    else:
        scene = _lib.Scene(geom.load_scene(f"data/{name}.geojson").arrays)
    m = np.ascontiguousarray(scene.visibility_matrix(False))
    print(name, scene.n, int(m.sum()), hashlib.sha256(m.tobytes()).hexdigest()[:24], flush=True)
