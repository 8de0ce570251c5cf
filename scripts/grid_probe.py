"""How many candidates the painter's covered-cell bitmap settles (not a test): python scripts/grid_probe.py lln | ny8"""
import ctypes
import sys
import time

sys.path.insert(0, ".")
from raytracing_amd import _lib, geometry as geom  # noqa: E402

name = sys.argv[1]
if name == "ny8":  # ny.geojson tiled 8 x 8, as the bench's fourth configuration has it
    from raytracing_amd import scenarios

    scene = _lib.Scene(geom.scene_arrays_of(scenarios.ny_tiled(8)[0].get_polygons_list()))
else:
    scene = _lib.Scene(geom.load_scene(f"data/{name}.geojson").arrays)
st = (ctypes.c_ulonglong * 8)()
_lib.load().rt_debug_visibility_stats(st)
for rep in range(2):
    t = time.perf_counter()
    vis = scene.visibility_matrix(False)
    dt = time.perf_counter() - t
    _lib.load().rt_debug_visibility_stats(st)
    print(f"matrix {1e3 * dt:.1f} ms, kernels {scene.visibility_timing()['rows']}, visible pairs {int(vis.sum())}; "
          f"bitmap: {st[6]} candidates tested, {st[7]} settled ({100.0 * st[7] / max(1, st[6]):.1f} %)", flush=True)
if hasattr(_lib.load(), "rt_debug_visibility_phases"):
    # (a build with -DRT_VIS_STAMPS: where the faces' time goes, in 10 ns ticks of the wall clock, summed over faces)
    ph = (ctypes.c_ulonglong * 16)()
    _lib.load().rt_debug_visibility_phases(ph)
    vis = scene.visibility_matrix(False)
    _lib.load().rt_debug_visibility_phases(ph)
    names = ["samples", "inside-one", "own edges", "collect", "canvas edges"]
    full = sum(ph[2 * i] for i in range(5))
    print("full decisions: " + ", ".join(f"{nm} {ph[2 * i] / 1e5:.1f} ms x{ph[2 * i + 1]}" for i, nm in enumerate(names)), flush=True)
    if hasattr(_lib.load(), "rt_debug_visibility_hist"):
        hh = (ctypes.c_ulonglong * 128)()
        _lib.load().rt_debug_visibility_hist(hh)
        print("examinations of canvas edges by entries of the decision (rows: <=8, <=16, <=32, more) and cuts of the part (1, 2, 3-4, 5-8, 9-16, 17-32, 33-64, more): wave-ms / count")
        for c in range(4):
            print("   " + "  ".join(f"{hh[8 * c + b] / 1e5:8.0f}/{hh[32 + 8 * c + b]:<8d}" for b in range(8)))
        print("full decisions by the width of the candidate's box (up to 0.25, 0.5, 1, 2, 4 rad, more): workgroup-ms / count / found visible")
        print("   " + "  ".join(f"{hh[64 + c] / 1e5:8.0f}/{hh[72 + c]}/{hh[80 + c]}" for c in range(6)))
        print("inside an examination (own and other entries' edges), shader cycles / 1e9: " + ", ".join(
            f"{nm} {hh[86 + i] / 1e9:.1f}" for i, nm in enumerate(["gathering the cuts", "pieces", "entries near the part", "(piece, entry) tests", "more than 64 cuts"])))
        names_g = ["waiting for the group to gather", "listing and staging", "the shortcuts", "marking cells", "waiting for the group's verdicts",
                   "going through the verdicts (full decisions with them)", "a stretch's bitmap tests (every stretch, wave 0)"]
        print(f"a group of candidates, one in eight timed ({hh[127]} groups), shader cycles / 1e9 summed over the waves: " + ", ".join(
            f"{nm} {hh[96 + i] / 1e9:.2f}" for i, nm in enumerate(names_g)))
        names_v = ["a member made ready for its full decision", "the full decision", "a member that shows, until it is on the canvas", "its shared edges", "its cells"]
        print("  inside the verdicts, wave 0 only: " + ", ".join(f"{nm} {hh[104 + i] / 1e9:.2f}" for i, nm in enumerate(names_v)))
    print(f"faces: {ph[12] / 1e5:.1f} ms summed over the workgroups (longest {ph[13] / 1e5:.2f} ms), of which full decisions "
          f"{full / 1e5:.1f} ms; span of the launch {(ph[14] - ph[15]) / 1e5:.1f} ms", flush=True)
if hasattr(_lib.load(), "rt_debug_visibility_face_log"):
    import numpy as np

    n = scene.n
    log = np.zeros((n, 4), dtype=np.uint64)
    _lib.load().rt_debug_visibility_face_log(ctypes.c_void_p(log.ctypes.data), n)
    start = (log[:, 0] - log[:, 0].min()).astype(float) / 1e5
    dur = log[:, 1].astype(float) / 1e5
    cand, shown = (log[:, 2] & np.uint64(0xffffffff)).astype(int), log[:, 3].astype(int)  # (high bits: tile number)
    end = start + dur
    print(f"faces: span {end.max():.1f} ms; durations: mean {dur.mean():.2f}, p50 {np.median(dur):.2f}, p99 {np.percentile(dur, 99):.1f}, max {dur.max():.1f} ms")
    for f in np.argsort(-dur)[:12]:
        print(f"  longest: face {f}: start {start[f]:.1f} ms, lasts {dur[f]:.1f} ms, candidates {cand[f]}, visible {shown[f]}")
    late = np.argsort(-end)[:6]
    for f in late:
        print(f"  face {f}: start {start[f]:.1f} ms, lasts {dur[f]:.1f} ms, candidates {cand[f]}, visible {shown[f]}")
    for t in (50, 100, 125, 150, 175):
        print(f"  at {t} ms: {(int(((start <= t) & (end > t)).sum()))} faces in flight, {int((start > t).sum())} not started")
    c = np.corrcoef(cand, dur)[0, 1]
    print(f"  correlation of a face's time with its candidates {c:.2f}, with candidates x visible {np.corrcoef(cand * shown, dur)[0, 1]:.2f}, with visible {np.corrcoef(shown, dur)[0, 1]:.2f}")
