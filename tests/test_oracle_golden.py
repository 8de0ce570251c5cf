"""The CPU oracle against vectors produced by the reference's own code (tests/golden/make_golden.py) and against the
known answers the reference's tests hold.  No GPU."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

from oracle import geometry_oracle as go


@pytest.fixture(scope="module")
def helpers():
    return np.load(os.path.join(GOLDEN, "helpers.npz"))


@pytest.fixture(scope="module")
def solvers():
    with open(os.path.join(GOLDEN, "solvers.json")) as f:
        return json.load(f)


def test_frames_gamma_lines(helpers):
    for v, m in zip(helpers["frame_in"], helpers["frame_out"]):
        np.testing.assert_array_equal(go.projection_matrix_from_vector(v), m)
    for p, g in zip(helpers["gamma_in"], helpers["gamma_out"]):
        np.testing.assert_array_equal(go.gamma(p), g)
    for a, b, x in zip(helpers["lines_a"], helpers["lines_b"], helpers["lines_x"]):
        np.testing.assert_array_equal(go.intersection_of_2d_lines(a, b), x)
    got = [go.point_on_edge_(p, helpers["edge"]) for p in helpers["edge_points"]]
    assert got == list(helpers["edge_on"])
    np.testing.assert_array_equal(go.sort_by_columns(helpers["sort_in"]), helpers["sort_out"])


def test_normalize_path_known_answers():
    """Known answers of the reference's tests/test_geometry_advanced.py:85-103."""
    from raytracing_amd.geometry import normalize_path

    si, Si, sr, Sr = normalize_path(np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0]], dtype=float))
    assert (si == [1, 0, 0]).all() and Si == 1 and (sr == [0, 1, 0]).all() and Sr == 1
    si, Si, sr, Sr = normalize_path(np.array([[0, 0, 0], [0, 0, 2], [0, 3, 2]], dtype=float))
    assert (si == [0, 0, 1]).all() and Si == 2 and (sr == [0, 1, 0]).all() and Sr == 3


def test_solver_vectors(helpers, solvers):
    """Same SciPy, same zero functions, same first guess: the oracle reproduces the reference's iterates bit for bit."""
    planes = helpers["cube_planes"]
    n = 0
    for case in solvers["reflection"]:
        x, sol = go.reflexion_points_from_origin_destination_and_planes(
            np.array(case["a"]), np.array(case["b"]), [planes[f] for f in case["faces"]])
        assert bool(sol.success) == case["success"] and int(sol.nfev) == case["nfev"]
        np.testing.assert_array_equal(x, np.array(case["x"]))
        n += 1
        # where hybr found a root it is the exact one
        if case["success"] and case["fun"] < 1e-10:
            xe, se = go.exact_reflexion_points(np.array(case["a"]), np.array(case["b"]), [planes[f] for f in case["faces"]])
            assert se.success
            np.testing.assert_allclose(xe, x, atol=1e-6, rtol=0)
    for case in solvers["diffraction"]:
        x, sol = go.diffraction_point_from_origin_destination_and_edge(np.array(case["a"]), np.array(case["b"]),
                                                                       np.array(case["edge"]))
        assert bool(sol.success) == case["success"]
        np.testing.assert_array_equal(np.asarray(x), np.array(case["x"]))
        xe, _ = go.exact_diffraction_point(np.array(case["a"]), np.array(case["b"]), np.array(case["edge"]))
        if case["fun"] < 1e-12:
            np.testing.assert_allclose(xe, np.array(case["x"]), atol=1e-6, rtol=0)
        n += 1
    for case in solvers["mixed"]:
        x, sol = go.reflexion_points_and_diffraction_point_from_origin_destination_planes_and_edge(
            np.array(case["a"]), np.array(case["b"]), [planes[f] for f in case["faces"]], np.array(case["edge"]))
        assert bool(sol.success) == case["success"]
        np.testing.assert_array_equal(x, np.array(case["x"]))
        if case["success"] and case["fun"] < 1e-10:
            xe, se = go.exact_reflexion_points_and_diffraction_point(
                np.array(case["a"]), np.array(case["b"]), [planes[f] for f in case["faces"]], np.array(case["edge"]))
            np.testing.assert_allclose(xe, x, atol=1e-6, rtol=0)
        n += 1
    assert n > 100


def test_chained_reflection_example_is_the_image_solution(solvers, helpers):
    """examples/plot_chained_reflections.py of the reference: cube of side 20, A = 0, B = (6,6,6), faces x=10 then
    x=-10: the image construction gives the points by hand."""
    case = solvers["reflection"][-1]
    assert case["faces"] == [2, 4] and case["success"]
    # mirror A in x=10 -> (20,0,0), then in x=-10 -> (-40,0,0); straight line from (-40,0,0) to B
    x2 = np.array([-10.0, 6 * 30 / 46, 6 * 30 / 46])
    x1 = np.array([10.0, 6 * 10 / 46, 6 * 10 / 46])
    np.testing.assert_allclose(np.array(case["x"]), np.array([x1, x2]), atol=1e-7)


def test_polygon_attributes_and_buildings():
    g = np.load(os.path.join(GOLDEN, "polygons.npz"))
    for i in range(int(g["n_polys"])):
        pts = g[f"poly{i}_points"]
        par = go.parametric(pts)
        np.testing.assert_array_equal(par, g[f"poly{i}_parametric"])
        np.testing.assert_array_equal(go.frame_matrix(pts, par[:3]), g[f"poly{i}_matrix"])
        np.testing.assert_array_equal(go.domain(pts), g[f"poly{i}_domain"])
        np.testing.assert_array_equal(go.centroid(pts), g[f"poly{i}_centroid"])
    fp = g["bld_footprint"]
    for name, f in (("ccw", fp), ("cw", fp[::-1].copy())):
        for keep in (True, False):
            tag = f"bld_{name}_{int(keep)}"
            polys = go.building_by_polygon_and_height(f.copy(), 7.5, keep_ground=keep)
            assert len(polys) == int(g[tag + "_n"])
            assert [p.part for p in polys] == list(g[tag + "_parts"])
            for j, p in enumerate(polys):
                np.testing.assert_array_equal(p.points, g[f"{tag}_{j}_points"])
                np.testing.assert_array_equal(p.parametric, g[f"{tag}_{j}_parametric"])
    cube = go.cube_by_point_and_side_length(np.array([3.0, 38.0, 18.0]), 0.2)
    np.testing.assert_array_equal(np.array([p.points for p in cube]), g["cube_points"])
    np.testing.assert_array_equal(np.array([p.parametric for p in cube]), g["cube_parametric"])


def test_sharp_edges_of_small_scene():
    g = np.load(os.path.join(GOLDEN, "sharp_edges_small.npz"))
    polys = []
    for i in range(int(g["n"])):
        p = go.Poly(g[f"p{i}"])
        p.parametric = g[f"par{i}"]
        polys.append(p)
    got = go.polygons_sharp_edges(polys)
    assert [(i, j) for _, i, j in got] == [tuple(p) for p in g["pairs"]]
    np.testing.assert_array_equal(np.array([e for e, _, _ in got]), g["edges"])
    assert len(got) == 46


def test_pip_inclusive():
    sq = np.array([[0.0, 0.0], [2.0, 0.0], [2.0, 2.0], [0.0, 2.0]])
    assert go.pip_inclusive(sq, (1, 1)) and not go.pip_inclusive(sq, (3, 1)) and not go.pip_inclusive(sq, (1, -1e-9))
    for p in [(0, 0), (2, 2), (1, 0), (0, 1), (2, 0.5)]:  # vertices and edges count as inside
        assert go.pip_inclusive(sq, p)
    # non-convex ring
    l_shape = np.array([[0.0, 0], [3, 0], [3, 1], [1, 1], [1, 3], [0, 3]])
    assert go.pip_inclusive(l_shape, (0.5, 2.5)) and not go.pip_inclusive(l_shape, (2, 2))
    assert go.pip_inclusive(l_shape[::-1], (2.5, 0.5))


def test_obstruction_and_containment_known_answers():
    wall = go.Poly(np.array([[0.0, 0, 0], [0, 4, 0], [0, 4, 3], [0, 0, 3]]))
    seg = lambda a, b: np.array([a, b], dtype=float)
    assert go.polygons_obstruct_line_path([wall], seg([-1, 2, 1], [1, 2, 1]))
    assert not go.polygons_obstruct_line_path([wall], seg([-1, 2, 1], [-0.5, 2, 1]))
    assert not go.polygons_obstruct_line_path([wall], seg([-1, 5, 1], [1, 5, 1]))
    assert not go.polygons_obstruct_line_path([wall], seg([-1, 2, 1], [0, 2, 1]))  # ends on the wall: t = 1
    assert go.contains_point(wall, np.array([0.0, 1, 1]), check_in_plane=True)
    assert not go.contains_point(wall, np.array([1e-3, 1, 1]), check_in_plane=True)
    assert go.contains_point(wall, np.array([1e-3, 1, 1]), check_in_plane=False)
    assert not go.contains_point(wall, np.array([0.0, 5, 1]), check_in_plane=True)


def test_visibility_oracle_three_walls():
    """A wall hidden behind a nearer, larger one is not visible; the nearer one is; a wall behind the viewer is not;
    a wall that contains the viewing axis maps to a self-crossing (azimuth, polar) ring and is skipped as invalid,
    which is what the reference's `is_valid` test does to it (geometry.py:992)."""
    from oracle import visibility_oracle as vo

    def rect(x, y0, y1, z0, z1, flip=False):
        pts = np.array([[x, y0, z0], [x, y1, z0], [x, y1, z1], [x, y0, z1]], dtype=float)
        return go.Poly(pts[::-1] if flip else pts)

    viewer = rect(0.0, -1, 1, 0, 2)              # normal +x, centre (0, 0, 1)
    near = rect(5.0, 2, 10, 0, 8, flip=True)     # faces the viewer, off the viewing axis
    far = rect(9.0, 10, 11.6, 5.4, 7.4, flip=True)  # well inside the near wall's solid angle
    behind = rect(-3.0, -2, 2, 0, 4)
    ahead = rect(7.0, -4, 4, -3, 5, flip=True)   # contains the viewing axis
    assert viewer.normal[0] > 0 and near.normal[0] < 0
    polys = [viewer, near, far, behind, ahead]
    vis, stage = vo.polygon_visibility_vector(viewer, polys, self_index=0, return_stage=True)
    assert list(stage) == [0, 4, 2, 0, 1]
    assert [bool(v) for v in vis] == [False, True, False, False, False]
    # without the near wall the far one is seen
    vis = vo.polygon_visibility_vector(viewer, [viewer, far, behind], self_index=0)
    assert [bool(v) for v in vis] == [False, True, False]
    # a back-facing wall in view passes the painter but not the normal filter
    vis, stage = vo.polygon_visibility_vector(viewer, [viewer, rect(5.0, 2, 10, 0, 8)], self_index=0, return_stage=True)
    assert list(stage) == [0, 3] and not vis.any()
