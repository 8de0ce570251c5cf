"""Config 5's field step alone (not a test): place_saint_jean, five emitters, 5 000 receivers, seven frequencies; the
solve in exact mode, then `steps` passes of description + fields + records per emitter with the library's events on.
python scripts/fields_sweep_probe.py [steps]      (under rocprofv3 --kernel-trace --stats for the kernels' own times)"""
import sys
import time

import numpy as np
import torch

torch.cuda.init()
sys.path.insert(0, ".")
from raytracing_amd import _lib, device_pipeline as dp, geometry as geom, scenarios  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
place, emitters, rx, freqs = scenarios.saint_jean_sweep()
polygons = place.get_polygons_list()
scene = _lib.Scene(geom.scene_arrays_of(polygons), device=0)
pairs, edges = scene.sharp_edges()
vis = scene.visibility_matrix(False)
d_rx = torch.tensor(rx, dtype=torch.float64, device="cuda:0")
pipes = []
for tx in emitters:
    cube = geom.Cube.by_point_and_side_length(tx, 0.2)
    ev = np.logical_or.reduce(scene.faces_visibility([(s.points, s.get_parametric(), -1) for s in cube.polygons], False))
    cand = scene.enumerate(vis, ev, pairs, edges, 2, with_diffraction=True)
    pipes.append(dp.DevicePipeline(scene, cand, tx, capacity=64 * len(rx)))
for timing in (False, True):
    scene.set_timing(timing)
    for _ in range(2):
        for pipe in pipes:
            pipe.run(d_rx, freqs, records=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernel_ms = call_ms = 0.0
    n_paths = 0
    per_emitter = []
    for _ in range(steps):
        for pipe in pipes:
            block = pipe.run(d_rx, freqs, records=True)
            n_paths += block.n_paths
            if timing:
                kernel_ms += block.fields_timing["kernel"]
                call_ms += block.fields_timing["call"]
                per_emitter.append((block.n_paths, round(block.fields_timing["kernel"], 4)))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    pf = n_paths / steps * len(freqs)
    print(f"timing={timing}: step {dt*1e3:.2f} ms, {pf:.0f} path-frequencies", flush=True)
    if timing:
        print(f"  field kernels {kernel_ms/steps:.3f} ms per step = {pf/(kernel_ms/steps*1e-3):.3e} path-frequencies/s; "
              f"field call {call_ms/steps:.3f} ms; per emitter (paths, kernel ms): {per_emitter[:5]}", flush=True)
