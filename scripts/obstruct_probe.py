"""Timing probe of the obstruction kernel alone on path-like segments (not a test)."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from raytracing_amd import _lib, geometry as geom, scenarios  # noqa: E402

name = sys.argv[1]
n_side = int(sys.argv[2]) if len(sys.argv) > 2 else 100
n_edges = int(sys.argv[3]) if len(sys.argv) > 3 else 300
path = f"data/{name}.geojson"
if not os.path.exists(path):
    path = f"data/{name}/{name}.geojson"
place = geom.load_place(path)
lo, hi = place.get_domain()
tx = scenarios.emitter_outside(place, (lo[0] + hi[0]) / 2, (lo[1] + hi[1]) / 2, 30.0)
rx = scenarios.lattice_receivers(place, n_side, 1.5)
polygons = place.get_polygons_list()
scene = _lib.Scene(geom.scene_arrays_of(polygons))
pairs, edges = scene.sharp_edges()
rng = np.random.default_rng(0)
pick = rng.choice(len(edges), size=min(n_edges, len(edges)), replace=False)
mid = edges[pick].mean(axis=1)  # a point on each chosen edge
# edge-major: consecutive segments share the edge point and go to consecutive receivers
seg = np.concatenate([np.repeat(mid, len(rx), axis=0), np.tile(rx, (len(mid), 1))], axis=1)
print(name, "polygons", len(polygons), "segments", len(seg), flush=True)
lib = _lib.load()
for label, order in (("edge-major", np.arange(len(seg))), ("shuffled", rng.permutation(len(seg)))):
    d_seg = torch.tensor(seg[order], dtype=torch.float64, device="cuda")
    d_out = torch.zeros(len(seg), dtype=torch.uint8, device="cuda")
    for mode in (1, 2):
        scene.set_occlusion_mode(mode)
        for rep in range(3):
            torch.cuda.synchronize()
            t = time.perf_counter()
            rc = lib.rt_obstruct_segments(scene.handle, ctypes.c_void_p(d_seg.data_ptr()), len(seg), None,
                                          ctypes.c_void_p(d_out.data_ptr()), 1)
            dt = time.perf_counter() - t
            assert rc == 0
        hits = int(d_out.sum().item())
        if hasattr(lib, "rt_debug_occlusion_stamps") and mode == 1:
            st = (ctypes.c_ulonglong * 8)()
            lib.rt_debug_occlusion_stamps(st)
            w = max(1, st[4])
            print("    per wave, us: loads", st[0] / w / 100, "staging", st[1] / w / 100, "record loop", st[2] / w / 100,
                  "drains", st[3] / w / 100, "total", st[6] / w / 100, "drains per wave", st[5] / w, "waves", st[4], "core clock MHz", 100.0 * st[7] / max(1, st[6]), flush=True)
        print(f"  {label:10s} mode={mode} {dt*1e3:8.3f} ms  hit fraction {hits/len(seg):.3f}  "
              f"{len(seg)*len(polygons)/dt:.3e} seg*poly/s", flush=True)
