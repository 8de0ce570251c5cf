import sys, time, ctypes
sys.path.insert(0, ".")
import numpy as np
from raytracing_amd import _lib, geometry as geom
place = geom.load_place("data/lln/lln.geojson") if __import__("os").path.exists("data/lln/lln.geojson") else geom.load_place("data/lln.geojson")
polygons = place.get_polygons_list()
scene = _lib.Scene(geom.scene_arrays_of(polygons))
faces = [(p.points, p.get_parametric(), i) for i, p in enumerate(polygons)]
scene.faces_visibility(faces[:4], False)
st = (ctypes.c_ulonglong * 8)()
_lib.load().rt_debug_visibility_stats(st)
if hasattr(_lib.load(), "rt_debug_visibility_phases"):
    _lib.load().rt_debug_visibility_phases((ctypes.c_ulonglong * 16)())
t = time.perf_counter()
v = scene.faces_visibility(faces, False)
print("matrix ms", 1e3 * (time.perf_counter() - t), "sum", int(v.sum()), __import__("hashlib").sha1(np.ascontiguousarray(v).tobytes()).hexdigest())
if hasattr(_lib.load(), "rt_debug_visibility_phases"):
    ph = (ctypes.c_ulonglong * 16)()
    _lib.load().rt_debug_visibility_phases(ph)
    print("items: busy sum ms", ph[12] / 1e5, "longest item ms", ph[13] / 1e5, "span ms", (ph[14] - ph[15]) / 1e5, "mean busy groups", ph[12] / max(1, ph[14] - ph[15]))
for _ in range(2):
    t = time.perf_counter()
    v = scene.faces_visibility(faces, False)
    print("matrix ms", 1e3 * (time.perf_counter() - t))
_lib.load().rt_debug_visibility_stats(st)
print("stats", list(st))

if hasattr(_lib.load(), "rt_debug_visibility_phases"):
    ph = (ctypes.c_ulonglong * 16)()
    _lib.load().rt_debug_visibility_phases(ph)
    names = ["inside-one", "samples", "B edges", "collect", "canvas edges", "pair proof"]
    print("phases (us per visit, visits): " + ", ".join(f"{nm} {ph[2*i]/max(1,ph[2*i+1])/100:.1f} x{ph[2*i+1]}" for i, nm in enumerate(names)))
    print("per edge examination: pair iterations", ph[12] / max(1, ph[13]), "cut iterations", ph[14] / max(1, ph[13]), "cuts", ph[15] / max(1, ph[13]), "examinations", ph[13])
