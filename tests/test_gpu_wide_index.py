"""A reflection level of more than 2^31 - 1 candidates (ray_tracing.py:316-322 recurses to any order): order 4 on the lln
map without diffraction, whose deepest level -- 7.2e9 chains -- is never stored.  A path of that level carries its
candidate index split over the library's two words (rt_common.h); checked here: the level's size, that the paths found
with an index of 2^31 or more are described by the chain that index names (worked out independently from the level
above and the visibility matrix), that every order-4 path obeys the law of reflection on the polygons of its chain,
that the lower orders come out as an order-3 solve gives them, and that the reference-compatible mode reads the same
indices."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def order4():
    from raytracing_amd import _lib, scenarios
    from raytracing_amd import geometry as geom

    place, tx, rx_all = scenarios.lln_grid(100)
    rx = rx_all[::39][:256]
    polygons = place.get_polygons_list()
    scene = _lib.Scene(geom.scene_arrays_of(polygons))
    cube = geom.Cube.by_point_and_side_length(tx, 0.2)
    vis, screens = scene.visibility_with_faces([(s.points, s.get_parametric(), -1) for s in cube.polygons])
    ev = np.logical_or.reduce(screens)
    no_edges = (np.zeros((0, 2), np.int32), np.zeros((0, 2, 3)))
    cand = scene.enumerate(vis, ev, *no_edges, 4, with_diffraction=False)
    res = scene.solve_paths(cand, tx, rx)
    yield dict(scene=scene, cand=cand, res=res, vis=np.asarray(vis, dtype=bool), ev=np.asarray(ev, dtype=bool), tx=np.asarray(tx, float).reshape(3),
               rx=rx, polygons=polygons, no_edges=no_edges)
    cand.close()
    scene.close()


def test_the_deepest_level_is_wider_than_32_bits(order4):
    m = order4
    counts = m["cand"].counts
    assert counts[3] > 3 * 2**31 and counts[2] < 2**31
    assert m["res"]["n_attempted"] == len(m["rx"]) * (1 + int(counts.sum()))
    deep = m["res"]["group"] == 3
    assert deep.sum() >= 10 and (m["res"]["cand"][deep] >= 2**31).sum() >= 5
    assert m["res"]["cand"].dtype == np.int64 and int(m["res"]["cand"][deep].max()) < counts[3]
    # no path twice
    keys = np.stack([m["res"]["rx"].astype(np.int64), m["res"]["group"].astype(np.int64), m["res"]["cand"]], axis=1)
    assert len(np.unique(keys, axis=0)) == len(keys)


def test_a_wide_index_names_the_chain_it_should(order4):
    m = order4
    res, cand, vis = m["res"], m["cand"], m["vis"]
    deep = np.flatnonzero(res["group"] == 3)
    kind, chain, _, _ = cand.describe(res["group"][deep], res["cand"][deep])
    assert (kind == 4).all()
    # the level above, and where each of its chains' children start in the level that is not stored
    parents, _ = cand.group(2)
    degree = vis.sum(axis=1).astype(np.int64)
    first_child = np.concatenate([[0], np.cumsum(degree[parents[:, -1]])])
    assert first_child[-1] == cand.counts[3]
    for c, got in zip(res["cand"][deep], chain):
        p = int(np.searchsorted(first_child, c, side="right")) - 1
        child = np.flatnonzero(vis[parents[p, -1]])[c - first_child[p]]
        assert got.tolist() == parents[p].tolist() + [int(child)]
    # and a handful of indices spread over the whole level, found or not
    picks = np.linspace(0, cand.counts[3] - 1, 9).astype(np.int64)
    kind, chain, _, _ = cand.describe(np.full(len(picks), 3), picks)
    for c, got in zip(picks, chain):
        p = int(np.searchsorted(first_child, c, side="right")) - 1
        assert got.tolist() == parents[p].tolist() + [int(np.flatnonzero(vis[parents[p, -1]])[c - first_child[p]])]


def test_every_order_4_path_is_a_chain_of_reflections(order4):
    m = order4
    res, cand = m["res"], m["cand"]
    deep = np.flatnonzero(res["group"] == 3)
    _, chain, _, _ = cand.describe(res["group"][deep], res["cand"][deep])
    planes = np.array([p.get_parametric() for p in m["polygons"]])
    boxes = np.array([np.concatenate([p.points.min(axis=0), p.points.max(axis=0)]) for p in m["polygons"]])
    for i, polys in zip(deep, chain):
        assert m["ev"][polys[0]] and all(m["vis"][a, b] for a, b in zip(polys[:-1], polys[1:]))
        pts = [m["tx"]] + [res["points"][i, k] for k in range(4)] + [m["rx"][res["rx"][i]]]
        for k, p in enumerate(polys):
            n, d = planes[p, :3], planes[p, 3]
            x = pts[k + 1]
            assert abs(n @ x + d) < 1e-6
            assert (x >= boxes[p, :3] - 1e-6).all() and (x <= boxes[p, 3:] + 1e-6).all()
            a, b = x - pts[k], pts[k + 2] - x
            a, b = a / np.linalg.norm(a), b / np.linalg.norm(b)
            np.testing.assert_allclose(b, a - 2 * (a @ n) * n, atol=1e-9)  # the law of reflection


def test_the_lower_orders_are_those_of_an_order_3_solve(order4):
    m = order4
    cand3 = m["scene"].enumerate(m["vis"], m["ev"], *m["no_edges"], 3, with_diffraction=False)
    try:
        res3 = m["scene"].solve_paths(cand3, m["tx"], m["rx"])
    finally:
        cand3.close()
    low = m["res"]["group"] < 3
    assert low.sum() == len(res3["rx"]) > 100
    for key in ("rx", "group", "cand", "points"):
        np.testing.assert_array_equal(np.asarray(m["res"][key])[low], res3[key], err_msg=key)


def test_reference_compatible_mode_reads_the_same_indices(order4):
    """Every path re-judged by the reference's solver: what is kept is a subset of the exact solve's paths, named by the
    same (receiver, group, candidate), among them paths of the deepest level."""
    m = order4
    scene = m["scene"]
    some = np.unique(m["res"]["rx"][m["res"]["group"] == 3])[:12]
    scene.set_reference_solver(True)
    try:
        kept = scene.solve_paths(m["cand"], m["tx"], m["rx"][some])
    finally:
        scene.set_reference_solver(False)
    exact = {(int(np.flatnonzero(some == r)[0]), int(g), int(c)) for r, g, c in zip(m["res"]["rx"], m["res"]["group"], m["res"]["cand"])
             if r in some}
    got = {(int(r), int(g), int(c)) for r, g, c in zip(kept["rx"], kept["group"], kept["cand"])}
    # (the reference's solver loses more paths the longer the chains: here it keeps three in four)
    assert got <= exact and len(got) > 0.5 * len(exact)
    assert any(g == 3 for _, g, _ in got) and any(g == 3 and c >= 2**31 for _, g, c in got | exact)
