"""Generates tests/golden/*.npz / *.json from the REFERENCE's own code (run in the authoring container only).

    python tests/golden/make_golden.py

The reference is imported from /root/reference through `_ref_import` (pass-through stand-ins for the numba
decorators; Shapely/GeoPandas-backed functions are not used).  Every vector below is the output of a reference
function on inputs recorded next to it; nothing of the reference's source is stored.  The GPU box has no
/root/reference: tests only read the files this script wrote.
"""
import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
warnings.filterwarnings("ignore")

from _ref_import import import_reference_geometry  # noqa: E402

geom = import_reference_geometry()
import raytracing.array_utils as ref_array_utils  # noqa: E402
import raytracing.container_utils as ref_container_utils  # noqa: E402
import electromagnetism_fun.materials_properties as ref_mat  # noqa: E402


def solver_vectors(rng):
    out = {}
    cube = geom.Cube.by_point_and_side_length(np.zeros(3), 20.0)
    planes_all = np.array([p.get_parametric() for p in cube.polygons])
    out["cube_planes"] = planes_all
    out["cube_points"] = np.array([p.points for p in cube.polygons])
    # the reference's own example: examples/plot_chained_reflections.py:17-49 (cube side 20, A = 0, B = (6,6,6))
    cases = []
    for k in (1, 2, 3):
        for _ in range(25):
            faces = rng.choice(6, size=k)
            if any(faces[i] == faces[i + 1] for i in range(k - 1)):
                continue
            a = rng.uniform(-8, 8, 3)
            b = np.array([6.0, 6.0, 6.0]) + rng.uniform(-3, 3, 3)
            x, sol = geom.reflexion_points_from_origin_destination_and_planes(a, b, [planes_all[f] for f in faces])
            cases.append(dict(faces=faces.tolist(), a=a.tolist(), b=b.tolist(), x=x.tolist(), success=bool(sol.success),
                              nfev=int(sol.nfev), fun=float(np.abs(sol.fun).max())))
    a = np.zeros(3)
    b = np.array([6.0, 6.0, 6.0])
    x, sol = geom.reflexion_points_from_origin_destination_and_planes(a, b, [planes_all[2], planes_all[4]])
    cases.append(dict(faces=[2, 4], a=a.tolist(), b=b.tolist(), x=x.tolist(), success=bool(sol.success),
                      nfev=int(sol.nfev), fun=float(np.abs(sol.fun).max())))
    out_json = {"reflection": cases, "diffraction": [], "mixed": []}
    edges = [np.array([[10.0, 10.0, -10.0], [10.0, 10.0, 10.0]]), np.array([[-3.0, 2.0, 1.0], [4.0, 5.0, 7.0]])]
    for edge in edges:
        for _ in range(15):
            a = rng.uniform(-8, 8, 3)
            b = np.array([30.0, 12.0, 5.0]) + rng.uniform(-5, 5, 3)
            x, sol = geom.diffraction_point_from_origin_destination_and_edge(a, b, edge.copy())
            out_json["diffraction"].append(dict(edge=edge.tolist(), a=a.tolist(), b=b.tolist(), x=np.asarray(x).tolist(),
                                                success=bool(sol.success), fun=float(np.abs(sol.fun).max())))
            for faces in ([0], [0, 3]):
                x, sol = geom.reflexion_points_and_diffraction_point_from_origin_destination_planes_and_edge(
                    a, b, [planes_all[f] for f in faces], edge.copy())
                out_json["mixed"].append(dict(edge=edge.tolist(), faces=faces, a=a.tolist(), b=b.tolist(),
                                              x=np.asarray(x).tolist(), success=bool(sol.success),
                                              fun=float(np.abs(sol.fun).max())))
    return out, out_json


def helper_vectors(rng):
    out = {}
    v = rng.normal(size=(12, 3))
    out["frame_in"] = v
    out["frame_out"] = np.array([geom.projection_matrix_from_vector(x) for x in v])
    pts = rng.normal(size=(10, 4, 3))
    out["gamma_in"] = pts
    out["gamma_out"] = np.array([geom.__dict__["__gamma__"](p) for p in pts])
    la, lb = rng.normal(size=(10, 2, 2)), rng.normal(size=(10, 2, 2))
    out["lines_a"], out["lines_b"] = la, lb
    out["lines_x"] = np.array([geom.intersection_of_2d_lines(a, b) for a, b in zip(la, lb)])
    edge = np.array([[1.0, 2.0, 3.0], [4.0, -1.0, 5.0]])
    ts = np.array([-0.1, 0.0, 0.3, 1.0, 1.0 + 1e-10, 1.1])
    p_on = edge[0] + ts[:, None] * (edge[1] - edge[0])
    p_off = p_on + np.array([0.0, 1e-6, 0.0])
    out["edge"] = edge
    out["edge_points"] = np.vstack([p_on, p_off])
    out["edge_on"] = np.array([geom.point_on_edge_(p, edge) for p in out["edge_points"]])
    path = rng.normal(size=(8, 3, 3))
    out["path_in"] = path
    res = [geom.normalize_path(p) for p in path]
    out["path_si"] = np.array([r[0] for r in res])
    out["path_Si"] = np.array([r[1] for r in res])
    out["path_sr"] = np.array([r[2] for r in res])
    out["path_Sr"] = np.array([r[3] for r in res])
    out["path_len"] = np.array([geom.path_length(p) for p in path])
    arr = rng.integers(0, 3, size=(9, 3)).astype(float)
    out["sort_in"] = arr
    out["sort_out"] = ref_array_utils.sort_by_columns(arr)
    return out


def polygon_vectors(rng):
    out = {}
    polys = []
    for _ in range(8):
        n = int(rng.integers(3, 7))
        ang = np.sort(rng.uniform(0, 2 * np.pi, n))
        p2 = np.column_stack([np.cos(ang), np.sin(ang)]) * rng.uniform(1, 5)
        m = geom.projection_matrix_from_vector(rng.normal(size=3))
        pts = np.column_stack([p2, np.zeros(n)]) @ m.T + rng.uniform(-10, 10, 3)
        polys.append(pts)
    out["n_polys"] = np.array(len(polys))
    for i, pts in enumerate(polys):
        p = geom.OrientedPolygon(pts)
        out[f"poly{i}_points"] = pts
        out[f"poly{i}_parametric"] = p.get_parametric()
        out[f"poly{i}_matrix"] = p.get_matrix()
        out[f"poly{i}_domain"] = p.get_domain()
        out[f"poly{i}_centroid"] = p.get_centroid()
    # buildings: one clockwise and one counter-clockwise footprint, with and without the bottom face
    fp_ccw = np.array([[0.0, 0.0, 0.0], [4.0, 0.0, 0.0], [5.0, 3.0, 0.0], [2.0, 5.0, 0.0], [-1.0, 2.0, 0.0]])
    for name, fp in (("ccw", fp_ccw), ("cw", fp_ccw[::-1].copy())):
        for keep in (True, False):
            b = geom.Building.by_polygon_and_height(fp.copy(), 7.5, keep_ground=keep)
            tag = f"bld_{name}_{int(keep)}"
            out[tag + "_n"] = np.array(len(b.polygons))
            for j, q in enumerate(b.polygons):
                out[f"{tag}_{j}_points"] = q.points
                out[f"{tag}_{j}_parametric"] = q.get_parametric()
            out[tag + "_parts"] = np.array([q.part for q in b.polygons])
    out["bld_footprint"] = fp_ccw
    cube = geom.Cube.by_point_and_side_length(np.array([3.0, 38.0, 18.0]), 0.2)
    out["cube_points"] = np.array([q.points for q in cube.polygons])
    out["cube_parametric"] = np.array([q.get_parametric() for q in cube.polygons])
    out["cube_distance"] = np.array(cube.polygons[0].distance_to_point(np.array([3.0, 38.0, 18.0])))
    return out


def sharp_edge_vectors():
    """The reference's polygons_sharp_edges_iter on a scene made by the reference's own Building class from the
    footprints our loader produces for data/small.geojson (loader parity itself is a separate question)."""
    from raytracing_amd import geometry as mine

    footprints = mine.preprocess_geojson("/root/reference/data/small.geojson", seed=0)
    all_xy = np.concatenate([f["ring"] for f in footprints])
    shift = (all_xy.min(axis=0) + all_xy.max(axis=0)) / 2
    polygons = []
    lo, hi = all_xy.min(axis=0) - shift, all_xy.max(axis=0) - shift
    ground = np.array([[lo[0], lo[1], 0], [hi[0], lo[1], 0], [hi[0], hi[1], 0], [lo[0], hi[1], 0]], dtype=float)
    polygons.append(geom.OrientedPolygon(ground))
    for f in footprints:
        ring = f["ring"] - shift
        if np.array_equal(ring[0], ring[-1]):
            ring = ring[:-1]
        b = geom.Building.by_polygon_and_height(np.column_stack([ring, np.zeros(len(ring))]), f["height"],
                                                keep_ground=False)
        polygons.extend(b.polygons)
    out = {"n": np.array(len(polygons))}
    for i, p in enumerate(polygons):
        out[f"p{i}"] = p.points
        out[f"par{i}"] = p.get_parametric()
    edges = list(geom.polygons_sharp_edges_iter(polygons))
    out["pairs"] = np.array([(i, j) for _, i, j in edges], dtype=np.int64)
    out["edges"] = np.array([e for e, _, _ in edges])
    return out


def misc_json():
    d = ref_container_utils.ManyToOneDict()
    d[0, 1] = "a"
    d[1, 2] = "b"
    d[3, 4] = "c"
    d[2, 3] = "d"
    lookup = {"1": [list(k) for k, _ in d[1]], "1,4,...": [list(k) for k, _ in d[1, 4, ...]],
              "9,...": [list(k) for k, _ in d[9, ...]]}
    props = {t: ref_mat.set_properties(t) for t in
             ["office", "appartments", "garage", "ground", "road", "tree_trunk", "tree_crown", "unknown_type"]}
    consts = dict(FREQUENCY=ref_mat.FREQUENCY, LAMBDA=ref_mat.LAMBDA, K=ref_mat.K, Z_0=ref_mat.Z_0)
    eps = {row["material"]: [complex(row["epsilon_eff"]).real, complex(row["epsilon_eff"]).imag]
           for _, row in ref_mat.DF_PROPERTIES.iterrows()}
    return dict(many_to_one=lookup, set_properties=props, constants=consts, epsilon_eff=eps)


def main():
    rng = np.random.default_rng(20240511)
    arrays, solver_json = solver_vectors(rng)
    arrays.update(helper_vectors(rng))
    np.savez_compressed(os.path.join(HERE, "helpers.npz"), **arrays)
    np.savez_compressed(os.path.join(HERE, "polygons.npz"), **polygon_vectors(rng))
    np.savez_compressed(os.path.join(HERE, "sharp_edges_small.npz"), **sharp_edge_vectors())
    with open(os.path.join(HERE, "solvers.json"), "w") as f:
        json.dump(solver_json, f)
    with open(os.path.join(HERE, "misc.json"), "w") as f:
        json.dump(misc_json(), f, indent=1)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
