"""Timing probe on larger scenes (not a test): precompute phases and one solve, with progress lines."""
import os
import sys
import time

import numpy as np
import torch

torch.cuda.init()

sys.path.insert(0, ".")
from raytracing_amd import _lib, geometry as geom, scenarios  # noqa: E402

name = sys.argv[1]
order = int(sys.argv[2]) if len(sys.argv) > 2 else 2
n_side = int(sys.argv[3]) if len(sys.argv) > 3 else 32
diff = int(sys.argv[4]) if len(sys.argv) > 4 else 0
path = f"data/{name}.geojson"
if not os.path.exists(path):
    path = f"data/{name}/{name}.geojson"
place = geom.load_place(path)
lo, hi = place.get_domain()
tx = scenarios.emitter_outside(place, (lo[0] + hi[0]) / 2, (lo[1] + hi[1]) / 2, 30.0)
rx = scenarios.lattice_receivers(place, n_side, 1.5)
if os.environ.get("RT_TILE_RX"):
    # receivers in 8x8 tiles instead of rows: a wave's 64 receivers are then neighbours in both directions
    idx = np.arange(n_side * n_side).reshape(n_side, n_side)
    t = int(os.environ["RT_TILE_RX"])
    order_ = np.concatenate([idx[a:a + t, b:b + t].ravel() for a in range(0, n_side, t) for b in range(0, n_side, t)])
    rx = rx[order_]
polygons = place.get_polygons_list()
print(name, "polygons", len(polygons), "rx", len(rx), flush=True)
scene = _lib.Scene(geom.scene_arrays_of(polygons))
scene.set_occlusion_mode(int(os.environ.get("RT_OCC_MODE", "0")))
t = time.time(); pairs, edges = scene.sharp_edges(); print("sharp edges", len(pairs), f"{time.time()-t:.3f}s", flush=True)
t = time.time(); vis = scene.visibility_matrix(False); print("visibility", int(vis.sum()), f"{time.time()-t:.3f}s", flush=True)
cube = geom.Cube.by_point_and_side_length(tx, 0.2)
ev = np.zeros(len(polygons), dtype=bool)
t = time.time()
import ctypes
st = (ctypes.c_ulonglong * 8)()
_lib.load().rt_debug_visibility_stats(st)
print("  matrix stats [cand,list,work,splits,shortcut,visible,canvas_edge_calls,gathers]", list(st), flush=True)
for v in scene.faces_visibility([(s.points, s.get_parametric(), -1) for s in cube.polygons], False):
    ev |= v
_lib.load().rt_debug_visibility_stats(st)
print("  screens stats", list(st), flush=True)
print("emitter visibility", int(ev.sum()), f"{time.time()-t:.3f}s", flush=True)
t = time.time(); cand = scene.enumerate(vis, ev, pairs, edges, order, with_diffraction=bool(diff)); print("candidates", cand.counts.tolist(), f"{time.time()-t:.3f}s", flush=True)
t = time.time(); res = scene.solve_paths(cand, tx, rx); dt = time.time() - t
print("solve", f"{dt:.3f}s", "paths", len(res["rx"]), "attempted", res["n_attempted"], f"{res['n_attempted']/dt:.3e} solves/s", scene.last_timing(), flush=True)
t = time.time(); res = scene.solve_paths(cand, tx, rx); dt = time.time() - t
print("solve again", f"{dt:.3f}s", f"{res['n_attempted']/dt:.3e} solves/s", flush=True)

d_rx = torch.tensor(rx, dtype=torch.float64, device="cuda")
cap = max(1 << 20, 4 * len(res["rx"]))
o_rx = torch.empty(cap, dtype=torch.int32, device="cuda"); o_group = torch.empty_like(o_rx); o_cand = torch.empty_like(o_rx)
o_pts = torch.empty((cap, _lib.RT_MAX_ORDER, 3), dtype=torch.float64, device="cuda")
for rep in range(3):
    torch.cuda.synchronize(); t = time.time()
    found, attempted = scene.solve_paths_device(cand, tx, d_rx.data_ptr(), len(rx), o_rx.data_ptr(), o_group.data_ptr(), o_cand.data_ptr(), o_pts.data_ptr(), cap)
    dt = time.time() - t
    print("device-resident solve", f"{dt*1e3:.3f} ms", found, f"{attempted/dt:.3e} solves/s", scene.last_timing()["ms"], scene.last_timing()["segments"], flush=True)
