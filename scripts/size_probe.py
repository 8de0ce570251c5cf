"""Step time of the lln map for several receiver counts (not a test): python scripts/size_probe.py [reference] 60 71 80 100/8/3"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from raytracing_amd import _lib, scenarios  # noqa: E402
from raytracing_amd import device_pipeline as dp  # noqa: E402
from raytracing_amd import geometry as geom  # noqa: E402

place, tx, _ = scenarios.lln_grid(10)
scene = _lib.Scene(geom.scene_arrays_of(place.get_polygons_list()))
pairs, edges = scene.sharp_edges()
cube = geom.Cube.by_point_and_side_length(tx, 0.2)
vis, screens = scene.visibility_with_faces([(s.points, s.get_parametric(), -1) for s in cube.polygons])
cand = scene.enumerate(vis, np.logical_or.reduce(screens), pairs, edges, 2, with_diffraction=True)
scene.set_timing(True)
args = sys.argv[1:]
if args and args[0] == "reference":  # the reference-compatible mode (every found path re-judged by the reference's solver)
    scene.set_reference_solver(True)
    args = args[1:]
for arg in args:
    # "71": a 71 x 71 map; "100/8": the first of eight contiguous blocks of the 100 x 100 map (one rank of eight)
    parts = arg.split("/")
    n_side = int(parts[0])
    _, _, rx = scenarios.lln_grid(n_side)
    if len(parts) > 1:
        per = (len(rx) + int(parts[1]) - 1) // int(parts[1])
        k = int(parts[2]) if len(parts) > 2 else 0
        rx = rx[k * per:(k + 1) * per]
    d_rx = torch.tensor(rx, dtype=torch.float64, device="cuda:0")
    pipe = dp.DevicePipeline(scene, cand, tx, capacity=40 * len(rx))
    out = []
    for _ in range(5):
        t0 = time.perf_counter()
        block = pipe.run(d_rx, [3.6e9], fields=False)
        out.append((round(1e3 * (time.perf_counter() - t0), 2),) + tuple(round(float(block.solve_timing["ms"][k]), 2)
                                                                         for k in ("reflect", "diffract", "obstruct", "finalize")) +
                   (round(float((block.solve_timing.get("rejudge") or {}).get("ms", 0.0)), 2),))
    print(arg, len(rx), "paths", block.n_paths, "(wall, reflect, diffract, obstruct, finalize, rejudge ms) per step:", out[1:], flush=True)
