"""Every rank's share of the lln rows, painted with the library's own tile rules, against the rows painted whole
(not a test): python scripts/shard_parity_probe.py [repeats]"""
import sys

sys.path.insert(0, ".")
import numpy as np
import torch

from raytracing_amd import _lib, scenarios
from raytracing_amd import geometry as geom
from raytracing_amd import multithread_solve as mts

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
torch.zeros(1, device="cuda:0").add_(1)
place, tx = scenarios.lln_place()
scene = _lib.Scene(geom.scene_arrays_of(place.get_polygons_list()))
cube = geom.Cube.by_point_and_side_length(tx, 0.2)
screens = [(s.points, s.get_parametric(), -1) for s in cube.polygons]
scene.set_visibility_tiling(0, 0)  # every face whole
whole = mts.paint_dealt_block_on_device(scene, screens, False, 1, 0).cpu().numpy()
scene.set_visibility_tiling(-1, 0)
items = np.arange(scene.n + len(screens))
bad = 0
for world in (1, 2, 4, 8):
    for rank in range(world):
        mine = items[rank::world]
        for rep in range(reps):
            block = mts.paint_dealt_block_on_device(scene, screens, False, world, rank).cpu().numpy()[:len(mine)]
            d = int((block != whole[mine]).sum())
            bad += d
            if d:
                r, c = np.nonzero(block != whole[mine])
                print(f"world {world} rank {rank} run {rep}: {d} entries differ, first: item {mine[r[0]]} polygon {c[0]} whole {whole[mine][r[0], c[0]]} here {block[r[0], c[0]]}", flush=True)
    print(f"world {world}: done, {bad} differing entries so far", flush=True)
