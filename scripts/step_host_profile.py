"""Host-side cost of a map step (not a test): cProfile over 300 steps of a small block of the lln map, where the host's
share of a step is largest."""
import cProfile
import pstats
import sys

sys.path.insert(0, ".")
import numpy as np
import torch

from raytracing_amd import _lib, scenarios
from raytracing_amd import device_pipeline as dp
from raytracing_amd import geometry as geom
from raytracing_amd import multithread_solve as mts

n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 36
place, tx = scenarios.lln_place()
scene = _lib.Scene(geom.scene_arrays_of(place.get_polygons_list()))
pairs, edges = scene.sharp_edges()
cube = geom.Cube.by_point_and_side_length(tx, 0.2)
vis, screens = mts.sharded_visibility(scene, [(s.points, s.get_parametric(), -1) for s in cube.polygons], on_device=True)
cand = scene.enumerate(vis, np.logical_or.reduce(screens), pairs, edges, 2, with_diffraction=True)
rx = scenarios.lattice_receivers(place, n_side, 1.5)
d_rx = torch.tensor(rx, dtype=torch.float64, device="cuda:0")
pipe = dp.DevicePipeline(scene, cand, tx, capacity=40 * len(rx))
scene.set_timing(False)


def steps(k):
    for _ in range(k):
        block = pipe.run(d_rx, [12.5e9], records=True)
        dp.all_gather_records(dp.receiver_records(block), len(rx), None)


steps(20)
cProfile.run("steps(300)", "/tmp/step.prof")
pstats.Stats("/tmp/step.prof").sort_stats("tottime").print_stats(14)
