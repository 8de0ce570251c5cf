"""Tile settings for the rows of one rank's share of the lln matrix, without the screens (not a test):
python scripts/shard_rows_sweep.py <world> <rank> min,per [min,per ...]   (-1,0 = the library's own choice)"""
import sys
import time

sys.path.insert(0, ".")
import torch

from raytracing_amd import _lib, scenarios
from raytracing_amd import geometry as geom
from raytracing_amd import multithread_solve as mts

world, rank = int(sys.argv[1]), int(sys.argv[2])
torch.zeros(1, device="cuda:0").add_(1)
place, tx = scenarios.lln_place()
scene = _lib.Scene(geom.scene_arrays_of(place.get_polygons_list()))
scene.set_timing(True)
want = None
for setting in [(-1, 0)] + [tuple(int(v) for v in a.split(",")) for a in sys.argv[3:]]:
    scene.set_visibility_tiling(*setting)
    for rep in range(2):
        torch.cuda.synchronize()
        t = time.perf_counter()
        block = mts.paint_dealt_block_on_device(scene, [], False, world, rank)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
    if want is None:
        want = block.clone()
    assert torch.equal(block, want), "different result"
    tm = scene.visibility_timing()["rows"]
    print(setting, f"{1e3 * dt:.0f} ms; mapping {tm['prepare']:.1f}, sorting and tiles {tm['sort']:.1f}, painting {tm['paint']:.1f}", flush=True)
