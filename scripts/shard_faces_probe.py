"""The faces of one rank's share of the lln matrix, one by one (not a test; needs a build with -DRT_VIS_STAMPS):
python scripts/shard_faces_probe.py <world> <rank>"""
import ctypes
import sys

import numpy as np

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
for rep in range(2):
    mts.paint_dealt_block_on_device(scene, [], False, world, rank)
    torch.cuda.synchronize()
print(scene.visibility_timing()["rows"])
n = (scene.n + world - 1) // world
log = np.zeros((n, 4), dtype=np.uint64)
_lib.load().rt_debug_visibility_face_log(ctypes.c_void_p(log.ctypes.data), n)
used = log[:, 1] > 0
start = (log[:, 0].astype(float) - log[used, 0].min()) / 1e5
dur = log[:, 1].astype(float) / 1e5
end = start + dur
print(f"{int(used.sum())} faces logged (a face in sectors: its last item); span {end[used].max():.1f} ms; sum of the logged times {dur[used].sum():.0f} ms")
for f in np.argsort(-end)[:12]:
    print(f"  ends last: face {f}: start {start[f]:.1f}, lasts {dur[f]:.1f} ms, candidates {int(log[f, 2]) & 0xffffffff}, visible {log[f, 3]}, "
          f"tile {(int(log[f, 2]) >> 32) & 0xffff}{' (in tiles)' if (int(log[f, 2]) >> 48) & 1 else ' (whole)'}")
for t in (10, 25, 50, 75, 100):
    print(f"  at {t} ms: {int((used & (start <= t) & (end > t)).sum())} in flight, {int((used & (start > t)).sum())} not started")
