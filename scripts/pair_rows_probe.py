"""Single rows of a scene's visibility matrix, alone (painted in sectors) and in a batch of many faces (painted whole):
python scripts/pair_rows_probe.py c3_lln 5496 6713   (not a test)"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import parity_configs as pc
from raytracing_amd import _lib, geometry as geom
w = pc.workload(sys.argv[1])
a, b = int(sys.argv[2]), int(sys.argv[3])
polygons = w["place"].get_polygons_list()
scene = _lib.Scene(geom.scene_arrays_of(polygons))
face = lambda i: (polygons[i].points, polygons[i].get_parametric(), i)
for x, y in [(a, b), (b, a)]:
    alone = scene.faces_visibility([face(x)], False)[0]
    lo = max(0, min(x - 150, len(polygons) - 300))
    batch = scene.faces_visibility([face(i) for i in range(lo, lo + 300)], False)[x - lo]
    print(f"row {x} -> {y}: alone {bool(alone[y])} (sees {int(alone.sum())}), in a batch of 300 {bool(batch[y])} (sees {int(batch.sum())})", flush=True)
    np.save(f"gpurun_out/row_{x}.npy", batch)
