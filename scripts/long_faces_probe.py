"""Are the longest faces of the lln matrix painted faster by the 16-wave workgroups with the large canvas store (what a
batch of up to 256 faces gets) than by the 4-wave ones (not a test)?  Rows of the K faces that see most, painted whole
in a call of their own, and the rest in another, against all rows in one call."""
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

from raytracing_amd import _lib, scenarios
from raytracing_amd import geometry as geom

torch.zeros(1, device="cuda:0").add_(1)
place, tx = scenarios.lln_place()
scene = _lib.Scene(geom.scene_arrays_of(place.get_polygons_list()))
n = scene.n
ids = np.arange(n, dtype=np.int32)
scene.visibility_rows(ids[:300])


def clock(fn):
    t = time.perf_counter()
    out = fn()
    return out, 1e3 * (time.perf_counter() - t)


rows, ms = clock(lambda: scene.visibility_rows(ids))
print(f"all rows in one call: {ms:.0f} ms, kernels {scene.visibility_timing()['rows']}", flush=True)
seen = rows.sum(axis=1)
order = np.argsort(-seen, kind="stable")
print("most seen:", seen[order[:8]].tolist(), "median", int(np.median(seen)))
scene.set_visibility_tiling(0, 0)  # faces painted whole, also in a small batch
for k in [int(a) for a in sys.argv[1:]]:
    long_ids = np.sort(order[:k]).astype(np.int32)
    rest_ids = np.sort(order[k:]).astype(np.int32)
    a, ms_long = clock(lambda: scene.visibility_rows(long_ids))
    t_long = scene.visibility_timing()["rows"]["paint"]
    b, ms_rest = clock(lambda: scene.visibility_rows(rest_ids))
    t_rest = scene.visibility_timing()["rows"]["paint"]
    ok = np.array_equal(a, rows[long_ids]) and np.array_equal(b, rows[rest_ids])
    print(f"K={k}: long faces {ms_long:.0f} ms (paint {t_long:.0f}), rest {ms_rest:.0f} ms (paint {t_rest:.0f}), same rows: {ok}", flush=True)
