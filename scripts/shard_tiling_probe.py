"""Sector settings for the largest faces of a shard of the lln matrix (not a test): painting time of rank 0's block in
worlds of 1, 2, 4 and 8 for several (minimum candidates, sector size) settings; -1 0 = the library's own choice."""
import sys
import time

sys.path.insert(0, ".")
import torch

from raytracing_amd import _lib, scenarios
from raytracing_amd import geometry as geom
from raytracing_amd import multithread_solve as mts

torch.zeros(1, device="cuda:0").add_(1)
place, tx = scenarios.lln_place()
scene = _lib.Scene(geom.scene_arrays_of(place.get_polygons_list()))
scene.set_timing(True)
cube = geom.Cube.by_point_and_side_length(tx, 0.2)
screens = [(s.points, s.get_parametric(), -1) for s in cube.polygons]
mts.paint_dealt_block_on_device(scene, screens, False, 8, 3)
want = None
for setting in [(-1, 0)] + [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]:
    scene.set_visibility_tiling(*setting)
    line = []
    for world in (1, 2, 4, 8):
        for rank in ((0, 1) if world == 8 else (0,)):
            torch.cuda.synchronize()
            t = time.perf_counter()
            block = mts.paint_dealt_block_on_device(scene, screens, False, world, rank)
            torch.cuda.synchronize()
            tm = scene.visibility_timing()
            line.append(f"W{world}r{rank} {1e3 * (time.perf_counter() - t):.0f} ms (rows: mapping {tm['rows']['prepare']:.0f}, sorting {tm['rows']['sort']:.0f}, "
                        f"painting {tm['rows']['paint']:.0f}; screens {tm['faces']['total']:.0f})")
            if world == 1:
                if want is None:
                    want = block.clone()
                assert torch.equal(block, want), "different result"
    print(setting, "  ".join(line), flush=True)
