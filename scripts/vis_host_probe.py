"""Where the host time of the visibility stage goes: the bench's own call (matrix left on the device + the emitter's
screens), three times, wall clock next to the kernels' event times.  Run under rocprofv3 --kernel-trace --hip-trace for
the timeline."""
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

from raytracing_amd import _lib, scenarios
from raytracing_amd import geometry as geom
from raytracing_amd import multithread_solve as mts

torch.zeros(1, device="cuda:0").add_(1)
place, tx = scenarios.lln_place()
polygons = place.get_polygons_list()
scene = _lib.Scene(geom.scene_arrays_of(polygons))
cube = geom.Cube.by_point_and_side_length(tx, 0.2)
screens = [(s.points, s.get_parametric(), -1) for s in cube.polygons]
for k in range(3):
    torch.cuda.synchronize()
    t = time.perf_counter()
    vis, screen_vis = mts.sharded_visibility(scene, screens, on_device=True)
    wall = 1e3 * (time.perf_counter() - t)
    tm = scene.visibility_timing()
    print(f"call {k}: wall {wall:.1f} ms, kernels {tm}", flush=True)
print(int(vis.sum().item()), int(np.asarray(screen_vis).sum()))
