"""Makes tests/golden/{lln,levant}_rejudge.npz — CPU only, test infrastructure.

The reference keeps a candidate path when SciPy's hybr reports success on its system and the solver's own points pass
the checks (ray_tracing.py:307-314, 341-350, 369-376; geometry.py:682, 744, 823).  The GPU finds the candidates that
have a root in closed form, and in reference-compatible mode puts each of them through a restatement of that solver
call (k_rejudge).  This script takes EVERY path the GPU finds on a full-size workload in exact-root mode (dumped once
by scripts/dump_survivors.py on a GPU box: receiver, group, candidate, chain, edge), runs the oracle's solver functions
on it — SciPy's root exactly as the reference calls it — and then the oracle's checks on the solver's points, and
records per path

    success    what SciPy reports (sol.success)
    nfev       SciPy's function-evaluation count
    points     the solver's points (sol.x as the reference reshapes it; the diffraction point mapped back to the world)
    kept       success and the checks pass: the path is in the reference's result

and per receiver ``db`` = 10 log10(sum of the kept paths' powers / P_in) from the oracle's field computation on the
solver's points.  What this cannot find is a candidate that hybr reports although no root passes the checks (that needs
the full enumeration on the CPU: 100 CPU-minutes per receiver of lln); tests/golden/hybr_vs_exact.json shows none on
its receiver samples.

The oracle's obstruction test looks at every polygon for every segment (a Python loop of 7 615 polygons on lln); here
the polygons that cannot be hit — their box in the segment's frame, widened by a micrometre, misses the segment's foot
point or its extent — are left out before the oracle's own, unchanged test runs on the rest: same answers, 1 ms
instead of 40 per segment.

Usage:  python tests/golden/make_full_rejudge.py lln levant [--procs 8] [--dump gpurun_out]
"""
import argparse
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

_G = {}


def workload(name):
    from raytracing_amd import scenarios

    if name == "lln":
        return scenarios.lln_grid(100)
    if name == "levant":
        return scenarios.levant_grid(32)
    raise KeyError(name)


class NearPolygons:
    """The polygons a segment could hit, found with the arithmetic of the oracle's test on all vertices at once and a
    margin; the oracle's own function then decides on those."""

    def __init__(self, polys):
        from oracle import geometry_oracle as go

        self.exact_test = go.polygons_obstruct_line_path  # the oracle's own function, kept before anything wraps it
        self.polys = polys
        self.verts = np.concatenate([p.points for p in polys])
        self.starts = np.cumsum([0] + [len(p.points) for p in polys])[:-1]

    def near(self, points, margin=1e-6):
        from oracle import geometry_oracle as go

        matrix = go.projection_matrix_from_line_path(points).T
        pp = self.verts @ matrix.T
        ends = go.project_points(points, matrix)
        lo = np.minimum.reduceat(pp, self.starts, axis=0)
        hi = np.maximum.reduceat(pp, self.starts, axis=0)
        pa, z_min, z_max = ends[0], ends[0, 2], ends[1, 2]
        ok = (hi[:, 2] > z_min - margin) & (lo[:, 2] < z_max + margin) & (lo[:, 0] <= pa[0] + margin) & \
             (hi[:, 0] >= pa[0] - margin) & (lo[:, 1] <= pa[1] + margin) & (hi[:, 1] >= pa[1] - margin)
        return np.flatnonzero(ok)

    def obstruct(self, points, return_indices=False):
        from oracle import geometry_oracle as go

        idx = self.near(points)
        res = self.exact_test([self.polys[i] for i in idx], points, return_indices=return_indices)
        if return_indices:
            return [int(idx[k]) for k in res]
        return res


def _init(name):
    import parity_configs as pc

    warnings.filterwarnings("ignore")
    place, tx, rx = workload(name)
    polys = pc.oracle_polys_with_buildings(place)
    _G.update(polys=polys, tx=np.asarray(tx, float).reshape(3), rx=rx, near=NearPolygons(polys))


def _checks(lines, chain, edge):
    """check_reflections / check_reflections_and_diffraction of the oracle (ray_tracing.py:243-279) with the obstruction
    test on the near polygons."""
    from oracle import geometry_oracle as go

    polys, near = _G["polys"], _G["near"]
    refl = lines if edge is None else lines[:-1]
    for i, index in enumerate(chain):
        if not go.contains_point(polys[index], refl[i + 1, :], check_in_plane=True):
            return False
    if near.obstruct(refl[:2, :]):
        return False
    for i, _ in enumerate(chain):
        if near.obstruct(refl[i + 1:i + 3, :]):
            return False
    for i in range(len(refl) - 1):
        if np.linalg.norm(refl[i] - refl[i + 1]) < 0.005:
            return False
    if edge is not None:
        if not go.point_on_edge_(lines[-2, :], edge):
            return False
        if near.obstruct(lines[-2:, :]):
            return False
    return True


def _solve_block(job):
    """Paths [a, b) of the dump: solver call and checks."""
    from oracle import geometry_oracle as go

    rx_id, group, chain, n_refl, has_edge, epts = job
    tx, rx_all, polys = _G["tx"], _G["rx"], _G["polys"]
    n = len(rx_id)
    success = np.zeros(n, np.uint8)
    kept = np.zeros(n, np.uint8)
    nfev = np.zeros(n, np.int16)
    points = np.zeros((n, chain.shape[1], 3))
    for i in range(n):
        rx = rx_all[rx_id[i]]
        if group[i] < 0:  # a line of sight: no solver involved (ray_tracing.py:232-237)
            success[i] = kept[i] = 1
            continue
        ch = [int(c) for c in chain[i, :n_refl[i]]]
        planes = [polys[c].parametric for c in ch]
        edge = epts[i].reshape(2, 3).copy() if has_edge[i] else None
        if edge is None:
            pts, sol = go.reflexion_points_from_origin_destination_and_planes(tx, rx, planes)
        else:
            pts, sol = go.reflexion_points_and_diffraction_point_from_origin_destination_planes_and_edge(tx, rx, planes, edge)
        pts = np.asarray(pts, float).reshape(-1, 3)
        success[i] = 1 if sol.success else 0
        nfev[i] = min(int(sol.nfev), 32767)
        points[i, :len(pts)] = pts
        if sol.success:
            lines = np.vstack([tx.reshape(1, 3), pts, rx.reshape(1, 3)])
            kept[i] = 1 if _checks(lines, ch, edge) else 0
    return success, kept, nfev, points


class _Problem:
    """What FieldOracle.field_computation reads of a solved problem."""

    def __init__(self, emitter, n):
        from collections import defaultdict

        self.emitter = emitter
        self.los = {r: [] for r in range(n)}
        self.reflections = {r: defaultdict(list) for r in range(n)}
        self.diffractions = {r: defaultdict(list) for r in range(n)}


def _field_block(job):
    """dB of the receivers [r0, r1): the oracle's field computation on their kept paths, in the reference's row order."""
    from oracle import em_oracle as eo
    from oracle import geometry_oracle as go

    r0, r1, rx_id, group, chain, n_refl, has_edge, epolys, epts, points = job
    tx, rx_all, polys, near = _G["tx"], _G["rx"], _G["polys"], _G["near"]
    prob = _Problem(tx, r1 - r0)
    for i in range(len(rx_id)):  # sorted by (receiver, group, candidate): the reference's visiting order
        r = int(rx_id[i]) - r0
        rx = rx_all[rx_id[i]]
        if group[i] < 0:
            prob.los[r].append(np.vstack([tx, rx]))
            continue
        k = int(n_refl[i])
        ch = [int(c) for c in chain[i, :k]]
        npts = k + (1 if has_edge[i] else 0)
        lines = np.vstack([tx.reshape(1, 3), points[i, :npts], rx.reshape(1, 3)])
        if has_edge[i]:
            prob.diffractions[r][k + 1].append((lines, ch, (int(epolys[i, 0]), int(epolys[i, 1])), epts[i].reshape(2, 3)))
        else:
            prob.reflections[r][k].append((lines, ch))
    oracle = eo.FieldOracle(polys, frequency=12.5e9)
    real = go.polygons_obstruct_line_path
    try:
        # the field step's tree look-up scans every polygon per segment: the same near-polygon filter in front of it
        go.polygons_obstruct_line_path = lambda ps, pts, return_indices=False: near.obstruct(np.asarray(pts), return_indices)
        rows = oracle.field_computation(prob, rx_all[r0:r1])
    finally:
        go.polygons_obstruct_line_path = real
    return r0, r1, eo.power_db_each_receiver(rows, r1 - r0)


def make(name, dump_dir, procs):
    t0 = time.time()
    z = np.load(os.path.join(dump_dir, f"survivors_{name}.npz"))
    order = np.lexsort((z["exact_cand"], z["exact_group"], z["exact_rx"]))
    rx_id, group, cand = z["exact_rx"][order], z["exact_group"][order].astype(np.int8), z["exact_cand"][order]
    kind, chain = z["exact_kind"][order], z["exact_chain"][order]
    epolys, epts = z["exact_edge_polys"][order], z["exact_edge_points"][order]
    n_refl = (kind & 255).astype(np.int32)
    has_edge = kind >= 256
    n = len(rx_id)
    n_rx = len(z["rx"])
    step = 256
    jobs = [(rx_id[a:a + step], group[a:a + step], chain[a:a + step], n_refl[a:a + step], has_edge[a:a + step],
             epts[a:a + step]) for a in range(0, n, step)]
    with mp.Pool(procs, initializer=_init, initargs=(name,)) as pool:
        parts = []
        for k, res in enumerate(pool.imap(_solve_block, jobs, chunksize=1)):
            parts.append(res)
            if k % 50 == 0:
                print(f"[{name}] solver calls: block {k + 1}/{len(jobs)} ({time.time() - t0:.0f} s)", flush=True)
        success = np.concatenate([p[0] for p in parts])
        kept = np.concatenate([p[1] for p in parts])
        nfev = np.concatenate([p[2] for p in parts])
        points = np.concatenate([p[3] for p in parts])
        print(f"[{name}] {n} paths: hybr reports {int(success.sum())}, kept after the checks {int(kept.sum())} "
              f"({time.time() - t0:.0f} s)", flush=True)
        # fields of the kept paths, receivers in blocks
        sel = np.flatnonzero(kept)
        bounds = np.searchsorted(rx_id[sel], np.arange(0, n_rx + 1))
        blocks = []
        rstep = 64
        for r0 in range(0, n_rx, rstep):
            r1 = min(n_rx, r0 + rstep)
            s = sel[bounds[r0]:bounds[r1]]
            blocks.append((r0, r1, rx_id[s], group[s], chain[s], n_refl[s], has_edge[s], epolys[s], epts[s], points[s]))
        db = np.full(n_rx, -np.inf)
        for k, (r0, r1, part) in enumerate(pool.imap_unordered(_field_block, blocks, chunksize=1)):
            db[r0:r1] = part
            if k % 20 == 0:
                print(f"[{name}] fields: block {k + 1}/{len(blocks)} ({time.time() - t0:.0f} s)", flush=True)
    out = os.path.join(HERE, f"{name}_rejudge.npz")
    np.savez_compressed(out, rx=rx_id, group=group, cand=cand, success=success, kept=kept, nfev=nfev, points=points, db=db,
                        n_receivers=n_rx, frequency_hz=12.5e9)
    print(f"[{name}] wrote {out} ({os.path.getsize(out) / 1e6:.1f} MB) in {time.time() - t0:.0f} s", flush=True)
    # what the GPU's own reference-compatible mode gave in the same dump, for a first look
    ref_keys = set(zip(z["reference_rx"].tolist(), z["reference_group"].tolist(), z["reference_cand"].tolist()))
    want = set(zip(rx_id[kept == 1].tolist(), group[kept == 1].tolist(), cand[kept == 1].tolist()))
    print(f"[{name}] GPU reference mode kept {len(ref_keys)}, fixture keeps {len(want)}; only GPU {len(ref_keys - want)}, "
          f"only fixture {len(want - ref_keys)}", flush=True)
    dg = z["reference_db"]
    fin = np.isfinite(db) & np.isfinite(dg)
    print(f"[{name}] receivers with power: fixture {int(np.isfinite(db).sum())}, GPU {int(np.isfinite(dg).sum())}; largest dB "
          f"difference {np.abs(db - dg)[fin].max() if fin.any() else 0.0:.3e}", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*", default=["levant", "lln"])
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--dump", default=os.path.join(ROOT, "gpurun_out"))
    a = ap.parse_args()
    for nm in a.names:
        make(nm, a.dump, a.procs)
