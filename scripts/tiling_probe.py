"""The visibility matrix with its longest faces painted in azimuth sectors (not a test):
python scripts/tiling_probe.py lln 6000:1024 6500:1024 ..."""
import sys
import time

sys.path.insert(0, ".")
from raytracing_amd import _lib, geometry as geom  # noqa: E402

loaded = geom.load_scene(f"data/{sys.argv[1]}.geojson")
scene = _lib.Scene(loaded.arrays)
ref = scene.visibility_matrix(False)
for spec in ["-1:0"] + sys.argv[2:]:
    lo, per = (int(v) for v in spec.split(":"))
    scene.set_visibility_tiling(lo, per)
    best = 1e9
    for rep in range(2):
        t = time.perf_counter()
        vis = scene.visibility_matrix(False)
        best = min(best, time.perf_counter() - t)
    print(f"faces with >= {lo} candidates in sectors of {per}: matrix {1e3 * best:.1f} ms, paint "
          f"{scene.visibility_timing()['rows']['paint']:.1f} ms, same matrix {bool((vis == ref).all())}", flush=True)
