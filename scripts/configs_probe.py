"""Configs 4 and 5 of the benchmark description end to end on one GPU (not a test):
python scripts/configs_probe.py c4 [tiles] | c5 [n_rx]"""
import sys
import time

import numpy as np
import torch

torch.cuda.init()
sys.path.insert(0, ".")
from raytracing_amd import _lib, geometry as geom, scenarios  # noqa: E402


def precompute(place, tx):
    polygons = place.get_polygons_list()
    scene = _lib.Scene(geom.scene_arrays_of(polygons))
    t = time.perf_counter()
    pairs, edges = scene.sharp_edges()
    vis = scene.visibility_matrix(False)
    print(f"  polygons {len(polygons)}, sharp edges {len(pairs)}, visibility {int(vis.sum())}: {time.perf_counter()-t:.2f}s", flush=True)
    return scene, polygons, pairs, edges, vis


def emitter_visibility(scene, tx):
    cube = geom.Cube.by_point_and_side_length(tx, 0.2)
    return np.logical_or.reduce(scene.faces_visibility([(s.points, s.get_parametric(), -1) for s in cube.polygons], False))


which = sys.argv[1]
if which == "c4":
    tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    t0 = time.perf_counter()
    place, tx, rx = scenarios.ny_tiled(tiles)
    print(f"config 4: ny x {tiles}x{tiles}, {len(rx)} receivers, scene {time.perf_counter()-t0:.2f}s", flush=True)
    scene, polygons, pairs, edges, vis = precompute(place, tx)
    ev = emitter_visibility(scene, tx)
    t = time.perf_counter()
    cand = scene.enumerate(vis, ev, np.zeros((0, 2), np.int32), np.zeros((0, 2, 3)), 3, with_diffraction=False)
    print(f"  candidates {cand.counts.tolist()} in {time.perf_counter()-t:.3f}s", flush=True)
    for rep in range(2):
        t = time.perf_counter()
        res = scene.solve_paths(cand, tx, rx)
        dt = time.perf_counter() - t
        print(f"  solve: {dt*1e3:.1f} ms, {res['n_attempted']:.3e} path-solves, {res['n_attempted']/dt:.3e}/s, paths {len(res['rx'])}",
              scene.last_timing()["ms"], flush=True)
else:
    n_rx = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    t0 = time.perf_counter()
    place, emitters, rx, freqs = scenarios.saint_jean_sweep(n_rx)
    print(f"config 5: saint-jean, {len(emitters)} emitters, {len(rx)} receivers, {len(freqs)} frequencies, scene {time.perf_counter()-t0:.2f}s", flush=True)
    scene, polygons, pairs, edges, vis = precompute(place, emitters[0])
    total_pf = 0
    t_all = time.perf_counter()
    for tx in emitters:
        t = time.perf_counter()
        ev = emitter_visibility(scene, tx)
        cand = scene.enumerate(vis, ev, pairs, edges, 2, with_diffraction=True)
        res = scene.solve_paths(cand, tx, rx)
        t_solve = time.perf_counter() - t
        keep = res["group"] >= 0
        n = len(res["rx"])
        kind = np.zeros(n, dtype=np.int32); chain = np.zeros((n, _lib.RT_MAX_ORDER), dtype=np.int32)
        epolys = np.zeros((n, 2), dtype=np.int32); epts = np.zeros((n, 6))
        kind[keep], chain[keep], epolys[keep], epts[keep] = cand.describe(res["group"][keep], res["cand"][keep])
        t = time.perf_counter()
        out = scene.fields(tx, rx, res["rx"], kind, chain, epolys, epts, res["points"], freqs)
        t_fields = time.perf_counter() - t
        total_pf += n * len(freqs)
        # the field kernels alone (device time) as path-frequencies per second, and against HBM by the byte model of the
        # field step: per path its points, chain and edge in (24 (k + 2) + 4 k + 104 B), per path and frequency the field
        # and the power out (56 B)
        k_inter = (kind & 255) + (kind >> 8)
        model_bytes = float((24.0 * (k_inter + 2) + 4.0 * k_inter + 104.0).sum()) + 56.0 * n * len(freqs)
        kernel_ms = scene.fields_timing()["kernel"]
        total_kernel_ms = locals().get("total_kernel_ms", 0.0) + kernel_ms
        total_bytes = locals().get("total_bytes", 0.0) + model_bytes
        db = out["power_db"]
        print(f"  tx {np.round(tx,1)}: {res['n_attempted']:.3e} path-solves, {n} paths, emitter visibility + enumerate + solve {t_solve:.3f}s, "
              f"fields {t_fields*1e3:.1f} ms, status ok {(out['status']==0).mean():.3f}, covered receivers {int(np.isfinite(db[:,0]).sum())}, "
              f"median dB at {freqs[0]/1e9:g} / {freqs[-1]/1e9:g} GHz: {np.median(db[np.isfinite(db[:,0]),0]):.1f} / {np.median(db[np.isfinite(db[:,-1]),-1]):.1f}", flush=True)
        cand.close()
    print(f"  all emitters: {time.perf_counter()-t_all:.2f}s, {total_pf} path-frequencies", flush=True)
    print(f"  field kernels: {total_kernel_ms:.2f} ms device time = {total_pf / (total_kernel_ms * 1e-3):.3e} path-frequencies/s, "
          f"{total_bytes / (total_kernel_ms * 1e-3) / 1e9:.1f} GB/s of the byte model = {total_bytes / (total_kernel_ms * 1e-3) / 8e12:.4f} of HBM peak",
          flush=True)
