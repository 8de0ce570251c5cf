"""Where the precompute of a coverage map goes: scene file, device scene, sharp edges, visibility matrix, emitter
screens (separately and in one call), enumeration.  usage: precompute_probe.py [n_side]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracing_amd import _lib, scenarios  # noqa: E402
from raytracing_amd import geometry as geom  # noqa: E402


def timed(label, fn):
    t = time.perf_counter()
    out = fn()
    print(f"{label:34s} {time.perf_counter() - t:7.3f} s", flush=True)
    return out


n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 100
place, tx, rx = timed("scene file + receivers", lambda: scenarios.lln_grid(n_side))
polygons = timed("polygon list", place.get_polygons_list)
arrays = timed("scene arrays", lambda: geom.scene_arrays_of(polygons))
scene = timed("device scene", lambda: _lib.Scene(arrays))
pairs, edges = timed("sharp edges", scene.sharp_edges)
cube = geom.Cube.by_point_and_side_length(tx, 0.2)
screens = [(s.points, s.get_parametric(), -1) for s in cube.polygons]
for rep in range(2):
    vis = timed("visibility matrix", lambda: scene.visibility_matrix(False))
    sv = timed("emitter screens", lambda: scene.faces_visibility(screens, False))
    vis2, sv2 = timed("matrix + screens in one call", lambda: scene.visibility_with_faces(screens))
    assert (vis == vis2).all() and (sv == sv2).all()
    for i in range(6):
        timed(f"  screen {i} alone", lambda: scene.faces_visibility(screens[i:i + 1], False))
ev = np.logical_or.reduce(sv)
cand = timed("enumeration", lambda: scene.enumerate(vis, ev, pairs, edges, 2, with_diffraction=True))
cand.close()
scene.close()
