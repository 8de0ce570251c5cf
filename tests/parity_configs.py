"""The five workloads of BASELINE.json at the sizes the parity fixtures are made for.

Shared by ``tests/golden/make_hybr_vs_exact.py`` (CPU only: runs the oracle with SciPy hybr and with exact roots on
these receivers and writes ``tests/golden/hybr_vs_exact.json``) and by the GPU tests that hold the HIP path against
that file.  Everything is seeded; a receiver is named by its index in the workload's full receiver array.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SWEEP_HZ = [3.5e9, 6e9, 12.5e9, 28e9, 30e9, 39e9, 60e9]


def _sample(n_total, n_pick, seed=0):
    return np.sort(np.random.default_rng(seed).choice(n_total, size=n_pick, replace=False))


def workload(name):
    """-> dict(place, tx (3,), rx_all (R,3), rx_index (sample, sorted), order, frequencies)."""
    from raytracing_amd import scenarios

    if name == "c1_small":
        place, tx, rx = scenarios.small_place(16)
        return dict(place=place, tx=tx, rx_all=rx, rx_index=np.arange(16), order=2, frequencies=[12.5e9])
    if name == "c2_levant":
        place, tx, rx = scenarios.levant_grid(32)
        return dict(place=place, tx=tx, rx_all=rx, rx_index=_sample(len(rx), 64), order=2, frequencies=[12.5e9])
    if name == "c3_lln":
        place, tx, rx = scenarios.lln_grid(100)
        return dict(place=place, tx=tx, rx_all=rx, rx_index=_sample(len(rx), 8), order=2, frequencies=[12.5e9])
    if name == "c4_ny":
        from raytracing_amd import geometry as geom

        place = geom.load_place(os.path.join(scenarios.DATA_DIR, "ny.geojson"))
        lo, hi = place.get_domain()
        tx = scenarios.emitter_outside(place, (lo[0] + hi[0]) / 2, (lo[1] + hi[1]) / 2, 20.0)
        rx = scenarios.lattice_receivers(place, 16, 1.5)
        return dict(place=place, tx=tx, rx_all=rx, rx_index=_sample(len(rx), 16), order=3, frequencies=[12.5e9])
    if name.startswith("c5_saint_jean_tx"):
        k = int(name[len("c5_saint_jean_tx"):])
        place, emitters, rx, _ = scenarios.saint_jean_sweep()
        return dict(place=place, tx=emitters[k], rx_all=rx, rx_index=_sample(len(rx), 32, seed=k), order=2,
                    frequencies=list(SWEEP_HZ))
    raise KeyError(name)


NAMES = ["c1_small", "c2_levant", "c3_lln", "c4_ny", "c5_saint_jean_tx0", "c5_saint_jean_tx1"]


def oracle_polys_with_buildings(place):
    """The oracle's view of a product place: same vertices, same reported planes, materials and building labels."""
    from oracle import geometry_oracle as go

    out, ids = [], {}
    for p in place.get_polygons_list():
        q = go.Poly(p.points, p.part)
        q.parametric = p.get_parametric().copy()
        q.material = p.properties["material"] if p.properties else "concrete"
        q.building = ids.setdefault(p.building_id, len(ids))
        out.append(q)
    return out


def path_key(rx, kind, order, chain, edge):
    return (int(rx), str(kind), int(order), tuple(int(c) for c in chain), tuple(int(c) for c in edge))


# ---- full-size workloads whose every surviving path is pinned against SciPy's hybr (tests/golden/*_rejudge.npz) ---------

FULL_SIZE = ("levant", "lln")


def full_size_workload(name):
    """-> (place, tx, rx, order, with_diffraction): config 2 (levant) and config 3 (lln) of BASELINE.json at full size."""
    from raytracing_amd import scenarios

    if name == "lln":
        place, tx, rx = scenarios.lln_grid(100)
        return place, tx, rx, 2, True
    if name == "levant":
        place, tx, rx = scenarios.levant_grid(32)
        return place, tx, rx, 2, False
    raise KeyError(name)


def full_size_solve(name, modes=("exact", "reference"), frequency=12.5e9):
    """Solves a full-size workload on the GPU in the given solver modes; per mode the paths sorted by (receiver, group,
    candidate) with their description, the per-receiver dB and the field status codes."""
    from raytracing_amd import _lib
    from raytracing_amd import geometry as geom

    place, tx, rx, order, with_diffraction = full_size_workload(name)
    polygons = place.get_polygons_list()
    scene = _lib.Scene(geom.scene_arrays_of(polygons))
    if with_diffraction:
        pairs, edges = scene.sharp_edges()
    else:
        pairs, edges = np.zeros((0, 2), np.int32), np.zeros((0, 2, 3))
    cube = geom.Cube.by_point_and_side_length(tx, 0.2)
    vis, screens = scene.visibility_with_faces([(s.points, s.get_parametric(), -1) for s in cube.polygons])
    ev = np.logical_or.reduce(screens)
    cand = scene.enumerate(vis, ev, pairs, edges, order, with_diffraction=with_diffraction)
    out = dict(tx=np.asarray(tx, float).reshape(3), rx=rx, order=order, with_diffraction=with_diffraction,
               n_polygons=len(polygons), ev=ev, edge_pairs=pairs, edges=edges, place=place)
    for mode in modes:
        scene.set_reference_solver(mode == "reference")
        res = scene.solve_paths(cand, tx, rx)  # sorted by (receiver, group, candidate)
        n = len(res["rx"])
        keep = res["group"] >= 0
        kind = np.zeros(n, dtype=np.int32)
        chain = np.zeros((n, _lib.RT_MAX_ORDER), dtype=np.int32)
        epolys = np.zeros((n, 2), dtype=np.int32)
        epts = np.zeros((n, 6))
        kind[keep], chain[keep], epolys[keep], epts[keep] = cand.describe(res["group"][keep], res["cand"][keep])
        fields = scene.fields(tx, rx, res["rx"], kind, chain, epolys, epts, res["points"], [frequency])
        out[mode] = dict(rx=res["rx"], group=res["group"].astype(np.int8), cand=res["cand"], points=res["points"][:, :order].copy(),
                         kind=kind, chain=chain[:, :order].copy(), edge_polys=epolys, edge_points=epts,
                         db=fields["power_db"][:, 0], status=fields["status"], n_attempted=res["n_attempted"],
                         rejudge=scene.rejudge_timing())
    scene.set_reference_solver(False)
    cand.close()
    scene.close()
    return out
