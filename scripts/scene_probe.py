import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
torch.cuda.init(); torch.zeros(1, device="cuda")
from raytracing_amd import _lib, geometry as geom
place = geom.load_place("data/lln.geojson")
polygons = place.get_polygons_list()
arrays = geom.scene_arrays_of(polygons)
for rep in range(3):
    t = time.perf_counter()
    scene = _lib.Scene(arrays, device=0)
    t1 = time.perf_counter()
    pairs, edges = scene.sharp_edges()
    t2 = time.perf_counter()
    print(f"scene {1e3*(t1-t):.1f} ms, sharp edges {1e3*(t2-t1):.1f} ms ({len(pairs)} pairs)", flush=True)
    del scene
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
scene = _lib.Scene(arrays, device=0); pairs, edges = scene.sharp_edges()
pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
