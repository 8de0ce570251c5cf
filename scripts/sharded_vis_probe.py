"""Where the time of the sharded visibility route goes on one GPU (not a test): a rank's block painted on the device for
worlds of 1 and 8, then the steps that follow the all-gather (order, symmetrise), each timed."""
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
n = scene.n
cube = geom.Cube.by_point_and_side_length(tx, 0.2)
screens = [(s.points, s.get_parametric(), -1) for s in cube.polygons]


def clock(label, fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    print(f"  {label}: {1e3 * (time.perf_counter() - t):.1f} ms", flush=True)
    return out


for world in (1, 8, 1, 8):
    print("world", world)
    block = clock("paint rank 0's block", lambda: mts.paint_dealt_block_on_device(scene, screens, False, world, 0))
    print("   kernels", scene.visibility_timing())
    gathered = clock("stand-in for the gather (copies)", lambda: torch.stack([block] * world))
    rows = clock("put in order", lambda: mts.dealt_rows_in_order(gathered, n + len(screens)))
    matrix = clock("symmetrise", lambda: (rows[:n] | rows[:n].T).contiguous())
    clock("screens to host", lambda: rows[n:].cpu().numpy().astype(bool))
whole = clock("single call, matrix on device", lambda: mts.sharded_visibility(scene, screens, on_device=True))
