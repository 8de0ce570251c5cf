"""Makes tests/golden/hybr_vs_exact.json — CPU only, test infrastructure.

The reference keeps a candidate path when SciPy's hybr reports success on the min-path-tracing system and the path then
passes the checks (ray_tracing.py:307-314, 341-350, 369-376; geometry.py:682, 744, 823).  The HIP path keeps it when the
system HAS a root that passes the checks.  This script measures the difference: on fixed receiver samples of the five
BASELINE workloads (tests/parity_configs.py) it runs the oracle's restatement of RayTracingProblem.solve twice — with
SciPy hybr exactly as the reference calls it, and with exact roots — followed by the oracle's field computation on
both path sets, and records

    paths_hybr      every path the hybr run keeps                       [rx, kind, order, chain, edge key]
    only_exact      paths with a root that hybr did not report          (GPU has them, the reference does not)
    only_hybr       paths hybr reports that are no root                 (the reference has them, GPU does not)
    db_hybr / db_exact   per receiver and frequency 10 log10(sum of path powers / P_in) of either path set

Usage:  python tests/golden/make_hybr_vs_exact.py [--procs 6] [--only c1_small,c2_levant] [--cache /tmp/hve_cache]
A workload that is already in the output file is kept unless it is named in --only.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import parity_configs as pc  # noqa: E402

OUT = os.path.join(HERE, "hybr_vs_exact.json")
_G = {}


def _row_job(i):
    from oracle import visibility_oracle as vo

    return i, vo.polygon_visibility_vector(_G["polys"][i], _G["polys"], self_index=i, packed=_G["packed"])


def _init_rows(name):
    from oracle import visibility_oracle as vo

    w = pc.workload(name)
    _G["polys"] = pc.oracle_polys_with_buildings(w["place"])
    _G["packed"] = vo.PackedScene(_G["polys"])


def precompute(name, cache_dir, procs):
    """Visibility matrix (oracle painter, rows dealt to the processes), emitter visibility and sharp edges."""
    from oracle import geometry_oracle as go
    from oracle import ray_tracing_oracle as ro

    path = os.path.join(cache_dir, name + ".npz")
    scene_key = name.split("_tx")[0]
    mpath = os.path.join(cache_dir, scene_key + "_matrix.npz")
    w = pc.workload(name)
    polys = pc.oracle_polys_with_buildings(w["place"])
    if os.path.exists(mpath):
        z = np.load(mpath)
        matrix, pairs, edges = z["matrix"], z["pairs"], z["edges"]
    else:
        t0 = time.time()
        n = len(polys)
        matrix = np.zeros((n, n), dtype=bool)
        with mp.Pool(procs, initializer=_init_rows, initargs=(name,)) as pool:
            for i, row in pool.imap_unordered(_row_job, range(n), chunksize=16):
                matrix[i] = row
        matrix = matrix | matrix.T
        sharp = go.polygons_sharp_edges(polys)
        pairs = np.array([[i, j] for _, i, j in sharp], dtype=np.int64).reshape(-1, 2)
        edges = np.array([e for e, _, _ in sharp], dtype=float).reshape(-1, 2, 3)
        np.savez_compressed(mpath, matrix=matrix, pairs=pairs, edges=edges)
        print(f"[{name}] matrix {n}x{n} and {len(pairs)} sharp edges in {time.time() - t0:.0f} s", flush=True)
    sharp_edges = [((int(i), int(j)), e) for (i, j), e in zip(pairs, edges)]
    o = ro.RayTracingOracle(w["tx"], polys, np.zeros((0, 3)), visibility_matrix=matrix, sharp_edges=sharp_edges)
    np.savez_compressed(path, matrix=matrix, pairs=pairs, edges=edges, ev=o.emitter_visibility)
    return path


def _init_solve(name, cache_path):
    warnings.filterwarnings("ignore")
    w = pc.workload(name)
    z = np.load(cache_path)
    _G.update(w=w, polys=pc.oracle_polys_with_buildings(w["place"]), matrix=z["matrix"], ev=z["ev"],
              sharp=[((int(i), int(j)), e) for (i, j), e in zip(z["pairs"], z["edges"])])


def _paths(oracle, rx_id):
    return sorted(pc.path_key(rx_id, k, o, c, e) for (_, k, o, c, e) in oracle.path_set())


def _solve_job(rx_id):
    from oracle import em_oracle as eo
    from oracle import ray_tracing_oracle as ro

    w = _G["w"]
    rx = w["rx_all"][rx_id].reshape(1, 3)
    out = {"rx": int(rx_id)}
    t0 = time.time()
    for solver in ("exact", "hybr"):
        o = ro.RayTracingOracle(w["tx"], _G["polys"], rx, visibility_matrix=_G["matrix"], sharp_edges=_G["sharp"],
                                precompute=False, solver=solver)
        o.emitter_visibility = _G["ev"]
        # the line-of-sight part of precompute (ray_tracing.py:232-237) for this receiver
        visible = [p for p, v in zip(_G["polys"], _G["ev"]) if v]
        line = np.vstack([np.asarray(w["tx"], dtype=float).reshape(3), rx[0]])
        o.counts["LOS"] += 1
        from oracle import geometry_oracle as go

        if not go.polygons_obstruct_line_path(visible, line):
            o.los[0].append(line)
        o.solve(max_order=w["order"])
        out["paths_" + solver] = _paths(o, rx_id)
        out["counts"] = dict(o.counts)
        db, asserts = [], []
        for f in w["frequencies"]:
            try:
                rows = eo.FieldOracle(_G["polys"], frequency=f).field_computation(o, rx)
                db.append(float(eo.power_db_each_receiver(rows, 1)[0]))
            except AssertionError as err:  # one of the reference's physics assertions (electromagnetism.py:294...)
                db.append(None)
                asserts.append(str(err))
        out["db_" + solver] = db
        out["asserts_" + solver] = asserts
    out["seconds"] = time.time() - t0
    return out


def run(name, cache_dir, procs):
    t0 = time.time()
    cache_path = precompute(name, cache_dir, procs)
    w = pc.workload(name)
    rx_ids = [int(i) for i in w["rx_index"]]
    results = []
    with mp.Pool(procs, initializer=_init_solve, initargs=(name, cache_path)) as pool:
        for k, res in enumerate(pool.imap_unordered(_solve_job, rx_ids, chunksize=1)):
            results.append(res)
            print(f"[{name}] receiver {res['rx']} done ({k + 1}/{len(rx_ids)}, {res['seconds']:.0f} s, "
                  f"{sum(res['counts'].values())} path-solves)", flush=True)
    results.sort(key=lambda r: r["rx"])
    hybr = [p for r in results for p in r["paths_hybr"]]
    exact = [p for r in results for p in r["paths_exact"]]
    s_h, s_e = set(map(_freeze, hybr)), set(map(_freeze, exact))
    counts = {}
    for r in results:
        for k, v in r["counts"].items():
            counts[k] = counts.get(k, 0) + v
    db_h = [r["db_hybr"] for r in results]
    db_e = [r["db_exact"] for r in results]
    diffs = [abs(a - b) for ra, rb in zip(db_h, db_e) for a, b in zip(ra, rb)
             if a is not None and b is not None and np.isfinite(a) and np.isfinite(b)]
    one_sided = sum(1 for ra, rb in zip(db_h, db_e) for a, b in zip(ra, rb)
                    if (a is None) != (b is None) or (a is not None and b is not None and np.isfinite(a) != np.isfinite(b)))
    z = np.load(cache_path)
    return {
        "tx": [float(v) for v in w["tx"]],
        "order": int(w["order"]),
        "frequencies_hz": [float(f) for f in w["frequencies"]],
        "rx_index": rx_ids,
        "rx": [[float(v) for v in w["rx_all"][i]] for i in rx_ids],
        "n_polygons": int(len(z["ev"])),
        "visibility_matrix_popcount": int(z["matrix"].sum()),
        "emitter_visibility_popcount": int(z["ev"].sum()),
        "n_sharp_edges": int(len(z["pairs"])),
        "path_solves": counts,
        "n_paths_hybr": len(hybr),
        "n_paths_exact": len(exact),
        "paths_hybr": [list(map(_thaw, p)) for p in sorted(s_h)],
        "only_exact": [list(map(_thaw, p)) for p in sorted(s_e - s_h)],
        "only_hybr": [list(map(_thaw, p)) for p in sorted(s_h - s_e)],
        "db_hybr": db_h,
        "db_exact": db_e,
        "field_asserts_hybr": {str(r["rx"]): r["asserts_hybr"] for r in results if r["asserts_hybr"]},
        "field_asserts_exact": {str(r["rx"]): r["asserts_exact"] for r in results if r["asserts_exact"]},
        "max_abs_db_difference": max(diffs) if diffs else 0.0,
        "receivers_with_power_on_one_side_only": one_sided,
        "receivers_differing_by_more_than_0p01_db": int(sum(
            1 for ra, rb in zip(db_h, db_e)
            if any(a is not None and b is not None and np.isfinite(a) and np.isfinite(b) and abs(a - b) > 0.01
                   for a, b in zip(ra, rb)))),
        "cpu_seconds": round(sum(r["seconds"] for r in results), 1),
        "wall_seconds": round(time.time() - t0, 1),
    }


def _freeze(p):
    return (p[0], p[1], p[2], tuple(p[3]), tuple(p[4]))


def _thaw(v):
    return list(v) if isinstance(v, tuple) else v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=6)
    ap.add_argument("--only", default="")
    ap.add_argument("--cache", default="/tmp/hve_cache")
    args = ap.parse_args()
    os.makedirs(args.cache, exist_ok=True)
    data = {}
    if os.path.exists(OUT):
        with open(OUT) as f:
            data = json.load(f)
    names = [n for n in args.only.split(",") if n] or [n for n in pc.NAMES if n not in data]
    for name in names:
        data[name] = run(name, args.cache, args.procs)
        d = data[name]
        print(f"[{name}] hybr {d['n_paths_hybr']} paths, exact {d['n_paths_exact']}, only exact {len(d['only_exact'])}, "
              f"only hybr {len(d['only_hybr'])}, max |dB| {d['max_abs_db_difference']:.4g}", flush=True)
        with open(OUT, "w") as f:
            json.dump(data, f, separators=(",", ":"), sort_keys=True)
            f.write("\n")


if __name__ == "__main__":
    main()
