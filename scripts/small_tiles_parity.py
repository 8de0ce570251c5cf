"""An eighth of the lln rows painted in small tiles, several times over, against the same rows painted whole (not a
test; the check that went wrong once in a dozen paintings before the phases of a full decision got flag values of
their own): python scripts/small_tiles_parity.py [min,per] [paintings]"""
import sys

sys.path.insert(0, ".")
import numpy as np
import torch

from raytracing_amd import _lib, scenarios
from raytracing_amd import geometry as geom
from raytracing_amd import multithread_solve as mts

setting = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "512,256").split(","))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
torch.zeros(1, device="cuda:0").add_(1)
place, tx = scenarios.lln_place()
scene = _lib.Scene(geom.scene_arrays_of(place.get_polygons_list()))
scene.set_visibility_tiling(0, 0)
want = mts.paint_dealt_block_on_device(scene, [], False, 8, 0).cpu().numpy()
scene.set_visibility_tiling(*setting)
bad = 0
for rep in range(reps):
    got = mts.paint_dealt_block_on_device(scene, [], False, 8, 0).cpu().numpy()
    d = int((got != want).sum())
    bad += d
    print(f"painting {rep}: {d} entries differ", flush=True)
print("all equal" if bad == 0 else f"{bad} differing entries in all")
