"""How many diffraction edges of the lln map cannot be reached from the emitter at all?  Upper bound: both end points and
the middle of the edge are hidden from the emitter.  Not a test."""
import sys
import numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, ".")
from raytracing_amd import _lib, geometry as geom, scenarios

place, tx, rx = scenarios.lln_grid(100)
polygons = place.get_polygons_list()
scene = _lib.Scene(geom.scene_arrays_of(polygons))
pairs, edges = scene.sharp_edges()
cube = geom.Cube.by_point_and_side_length(tx, 0.2)
vis, screens = scene.visibility_with_faces([(s.points, s.get_parametric(), -1) for s in cube.polygons])
ev = np.logical_or.reduce(screens)
edges = np.asarray(edges, dtype=float).reshape(-1, 2, 3)
pairs = np.asarray(pairs).reshape(-1, 2)
cand = ev[pairs[:, 0]] | ev[pairs[:, 1]]
e = edges[cand]
print("edges", len(edges), "diffraction candidates", int(cand.sum()))
pts = [e[:, 0], e[:, 1], e.mean(axis=1), 0.75 * e[:, 0] + 0.25 * e[:, 1], 0.25 * e[:, 0] + 0.75 * e[:, 1]]
hidden = np.ones(len(e), dtype=bool)
for k, p in enumerate(pts):
    # pull the point a hair towards the emitter so that the edge's own faces do not count
    q = p + 1e-6 * (tx - p) / np.linalg.norm(tx - p, axis=1)[:, None]
    segs = np.concatenate([np.repeat(tx[None, :], len(q), 0), q], axis=1)
    hit = np.asarray(scene.obstructed(segs.reshape(-1, 2, 3))).astype(bool)
    hidden &= hit
    print("sample point", k, "hidden from the emitter:", int(hit.sum()), "| all so far:", int(hidden.sum()), flush=True)
