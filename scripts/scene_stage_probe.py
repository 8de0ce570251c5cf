"""Host stages in front of the visibility call on the lln scene (not a test): wall time of each, three times over."""
import sys
import time

sys.path.insert(0, ".")
import torch

from raytracing_amd import _lib, scenarios
from raytracing_amd import geometry as geom

torch.zeros(1, device="cuda:0").add_(1)
for k in range(3):
    t0 = time.perf_counter()
    place, tx = scenarios.lln_place()
    t1 = time.perf_counter()
    polygons = place.get_polygons_list()
    t2 = time.perf_counter()
    arrays = geom.SceneArrays(polygons)
    t3 = time.perf_counter()
    scene = _lib.Scene(arrays, device=0)
    t4 = time.perf_counter()
    pairs, edges = scene.sharp_edges()
    t5 = time.perf_counter()
    print(f"place {1e3 * (t1 - t0):.1f}  polygon list {1e3 * (t2 - t1):.1f}  arrays {1e3 * (t3 - t2):.1f}  "
          f"device scene {1e3 * (t4 - t3):.1f}  sharp edges {1e3 * (t5 - t4):.1f} ms ({len(pairs)} edges)", flush=True)
    scene.close()
