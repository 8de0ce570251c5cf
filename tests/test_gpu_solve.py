"""GPU parity of RayTracingProblem (precompute + solve) against the CPU oracle."""
import numpy as np
import pytest

from conftest import oracle_polys, small_tx_rx

pytestmark = pytest.mark.gpu


def compare_problems(problem, oracle, atol=1e-6):
    got, want = problem.path_set(), oracle.path_set()
    only_gpu, only_ref = got - want, want - got
    # interaction points of the common paths
    worst = 0.0
    for r in oracle.reflections:
        for order, lst in oracle.reflections[r].items():
            mine = {tuple(c): l for l, c in problem.reflections[r][order]}
            for lines, chain in lst:
                if tuple(chain) in mine:
                    worst = max(worst, np.abs(mine[tuple(chain)] - lines).max())
    for r in oracle.diffractions:
        for order, lst in oracle.diffractions[r].items():
            mine = {(tuple(it[1]), tuple(it[2])): it[0] for it in problem.diffractions[r][order]}
            for it in lst:
                key = (tuple(it[1]), tuple(it[2]))
                if key in mine:
                    worst = max(worst, np.abs(mine[key] - it[0]).max())
    return only_gpu, only_ref, worst


def test_small_place_order2_matches_oracle(places):
    from oracle import ray_tracing_oracle as ro
    from raytracing_amd import geometry as geom
    from raytracing_amd.ray_tracing import RayTracingProblem

    place = geom.load_place("data/small.geojson")
    tx, rx = small_tx_rx(16)
    place.add_set_of_points(rx)
    problem = RayTracingProblem(tx.reshape(1, 3), place)
    oracle = ro.RayTracingOracle(tx, oracle_polys(place), rx)
    np.testing.assert_array_equal(problem.visibility_matrix, oracle.visibility_matrix)
    np.testing.assert_array_equal(problem.emitter_visibility, oracle.emitter_visibility)
    assert [k for k in problem.sharp_edges.keys()] == [k for k, _ in oracle.sharp_edges]
    assert {r: len(v) for r, v in problem.los.items()} == {r: len(v) for r, v in oracle.los.items()}
    problem.solve(max_order=2)
    oracle.solve(max_order=2)
    assert problem.path_solves_attempted == sum(oracle.counts.values())
    # 1. against the oracle with exact roots: identical path set
    exact = ro.RayTracingOracle(tx, oracle_polys(place), rx, visibility_matrix=oracle.visibility_matrix,
                                sharp_edges=oracle.sharp_edges, solver="exact")
    exact.solve(max_order=2)
    only_gpu, only_ref, worst = compare_problems(problem, exact)
    assert not only_gpu and not only_ref, f"only GPU: {sorted(only_gpu)[:5]}, only oracle: {sorted(only_ref)[:5]}"
    assert worst <= 1e-6, f"interaction points differ by {worst} m"
    # 2. against the oracle with SciPy hybr as the reference calls it: every path hybr keeps is found, with the same
    #    points; the extra ones are candidates whose system HAS a root that passes every check but for which hybr
    #    stalled or returned success=False (that they have such a root is what step 1 established)
    only_gpu, only_ref, worst = compare_problems(problem, oracle)
    assert not only_ref, f"paths only the hybr oracle has: {sorted(only_ref)[:5]}"
    assert worst <= 1e-6, f"interaction points differ by {worst} m"
    assert len(only_gpu) <= 0.05 * len(problem.path_set()), f"{len(only_gpu)} paths hybr misses"
    oracle = exact
    assert len(problem.solved_receivers) == 16
    # the order of the stored paths is the reference's visiting order
    for r in oracle.reflections:
        for order in oracle.reflections[r]:
            assert [tuple(c) for _, c in problem.reflections[r][order]] == [tuple(c) for _, c in oracle.reflections[r][order]]
    for r in oracle.diffractions:
        for order in oracle.diffractions[r]:
            assert [(tuple(i[1]), tuple(i[2])) for i in problem.diffractions[r][order]] == \
                   [(tuple(i[1]), tuple(i[2])) for i in oracle.diffractions[r][order]]


def test_levant_subset_order2_reflections(places):
    """place_levant with a handful of receivers: reflections up to order 2 plus diffractions against the oracle."""
    from oracle import ray_tracing_oracle as ro
    from raytracing_amd import geometry as geom
    from raytracing_amd.ray_tracing import RayTracingProblem

    place = geom.load_place("data/place_levant.geojson")
    rng = np.random.default_rng(11)
    lo, hi = place.get_domain()
    rx = np.column_stack([rng.uniform(lo[0] * 0.9, hi[0] * 0.9, 3), rng.uniform(lo[1] * 0.9, hi[1] * 0.9, 3), np.full(3, 1.2)])
    tx = np.array([0.0, 0.0, 18.7])
    place.add_set_of_points(rx)
    problem = RayTracingProblem(tx.reshape(1, 3), place)
    oracle = ro.RayTracingOracle(tx, oracle_polys(place), rx, visibility_matrix=problem.visibility_matrix)
    np.testing.assert_array_equal(problem.emitter_visibility, oracle.emitter_visibility)
    exact = ro.RayTracingOracle(tx, oracle_polys(place), rx, visibility_matrix=problem.visibility_matrix,
                                sharp_edges=oracle.sharp_edges, solver="exact")
    problem.solve(max_order=1)
    oracle.solve(max_order=1)
    exact.solve(max_order=1)
    only_gpu, only_ref, worst = compare_problems(problem, exact)
    assert not only_gpu and not only_ref, f"only GPU: {sorted(only_gpu)[:5]}, only oracle: {sorted(only_ref)[:5]}"
    assert worst <= 1e-6
    only_gpu, only_ref, worst = compare_problems(problem, oracle)
    assert not only_ref and worst <= 1e-6


def test_solve_quirks(places):
    from raytracing_amd import geometry as geom
    from raytracing_amd.ray_tracing import RayTracingProblem

    place = geom.load_place("data/small.geojson")
    tx, rx = small_tx_rx(4)
    place.add_set_of_points(rx)
    problem = RayTracingProblem(tx.reshape(1, 3), place)
    problem.solve(max_order=0)
    assert problem.solved_receivers is None and not any(problem.reflections[r] for r in range(4))
    problem.solve(max_order=1)
    assert problem.solved_receivers is None  # the reference returns before assigning it when max_order < 2
    n1 = sum(len(v) for r in range(4) for v in problem.reflections[r].values())
    problem.solve(max_order=1, receivers_indexs=[2])
    n2 = sum(len(v) for r in range(4) for v in problem.reflections[r].values())
    assert n2 == n1 + len(problem.reflections[2][1]) // 2  # a second call appends again, for receiver 2 only
    with pytest.raises(AssertionError):
        problem.solve(max_order=1, receivers_indexs=[7])
    with pytest.raises(AssertionError):
        RayTracingProblem(tx.reshape(1, 3), place, n_screens=7)


def test_multithread_solve_place_single_gpu(tmp_path, monkeypatch):
    """The reference's coverage-map entry point end to end on one GPU: files written, table consistent."""
    from raytracing_amd import electromagnetism_utils as emu
    from raytracing_amd import geometry as geom
    from raytracing_amd.multithread_solve import multithread_solve_place
    from raytracing_amd.ray_tracing import RayTracingProblem

    monkeypatch.setenv("RAYTRACING_AMD_RESULTS", str(tmp_path))
    place = geom.load_place("data/small.geojson")
    tx, rx = small_tx_rx(6)
    place.add_set_of_points(rx)
    em_path, rays_path = multithread_solve_place(place, tx.reshape(1, 3), "small", order=2)
    df = emu.load_df(em_path)
    assert set(df["rx_id"]) == set(range(6)) and (df["path_type"] == "LOS").sum() == 6
    db = emu.get_power_db_each_receiver(df)
    assert np.isfinite(db).all() and (db < 0).all()
    back = RayTracingProblem.from_json(rays_path)
    direct = RayTracingProblem(tx.reshape(1, 3), place)
    direct.solve(max_order=2)
    assert back.path_set() == direct.path_set()


def test_repeated_solves_and_obstruction_routes_agree(places):
    """A scene remembers the sizes its last solve needed and sizes the next call's launches from them, and it tests
    the queued segments either in one launch or path by path, last segment first, depending on the batch size.  Every
    route and every kernel choice must give the same paths."""
    from raytracing_amd import _lib
    from raytracing_amd import geometry as geom
    from raytracing_amd import scenarios

    place, tx, rx = scenarios.levant_grid(16)
    polygons = place.get_polygons_list()
    scene = _lib.Scene(geom.scene_arrays_of(polygons))
    pairs, edges = scene.sharp_edges()
    vis = scene.visibility_matrix(False)
    cube = geom.Cube.by_point_and_side_length(tx, 0.2)
    ev = np.logical_or.reduce(scene.faces_visibility([(s.points, s.get_parametric(), -1) for s in cube.polygons], False))
    cand = scene.enumerate(vis, ev, pairs, edges, 2, with_diffraction=True)

    def used_points(g):
        if g < 0:
            return 0
        n_refl, has_edge = cand.group_kind(int(g))
        return n_refl + (1 if has_edge else 0)

    def as_set(res):  # only the first n points of a row are meaningful
        return {(int(r), int(g), int(c)) + tuple(np.round(p[:used_points(g)].ravel(), 9)) for r, g, c, p in
                zip(res["rx"], res["group"], res["cand"], res["points"])}

    first = scene.solve_paths(cand, tx, rx)            # sizes unknown: path-by-path rounds
    second = scene.solve_paths(cand, tx, rx)           # sizes known and small: one launch over all segments
    assert len(first["rx"]) > 500
    assert as_set(first) == as_set(second)
    assert scene.last_timing()["ms"]["reflect"] > 0 and scene.last_timing()["ms"]["obstruct"] > 0
    for mode in (1, 2):
        scene.set_occlusion_mode(mode)
        assert as_set(scene.solve_paths(cand, tx, rx)) == as_set(first)
    scene.set_occlusion_mode(0)
    # a path queue that is too small at first: grown, chunk redone, same paths
    scene.set_path_queue_capacity(300)
    assert as_set(scene.solve_paths(cand, tx, rx)) == as_set(first)
    scene.set_path_queue_capacity(0)
    # no receivers, and an output capacity that is too small at first (the host wrapper asks again with room)
    empty = scene.solve_paths(cand, tx, np.zeros((0, 3)))
    assert len(empty["rx"]) == 0 and empty["n_attempted"] == 0
    # a subset of the receivers gives the subset of the paths
    sub = scene.solve_paths(cand, tx, rx[40:90])
    want = {(t[0] - 40,) + t[1:] for t in as_set(first) if 40 <= t[0] < 90}
    assert as_set(sub) == want
    cand.close()
    scene.close()


def test_ny_order3_matches_exact_oracle(places):
    """Dense blocks, chains of three reflections and two reflections + a diffraction (the K = 3 and M = 2 kernels):
    147 k path-solves for two receivers, identical path set and points within 1e-6 m."""
    from oracle import ray_tracing_oracle as ro
    from raytracing_amd import geometry as geom
    from raytracing_amd import scenarios
    from raytracing_amd.ray_tracing import RayTracingProblem

    place = geom.load_place("data/ny.geojson")
    lo, hi = place.get_domain()
    tx = scenarios.emitter_outside(place, (lo[0] + hi[0]) / 2, (lo[1] + hi[1]) / 2, 15.0)
    rng = np.random.default_rng(3)
    rx = np.column_stack([rng.uniform(lo[0], hi[0], 2), rng.uniform(lo[1], hi[1], 2), np.full(2, 1.5)])
    place.add_set_of_points(rx)
    problem = RayTracingProblem(tx.reshape(1, 3), place)
    exact = ro.RayTracingOracle(tx, oracle_polys(place), rx, solver="exact")
    np.testing.assert_array_equal(problem.visibility_matrix, exact.visibility_matrix)
    np.testing.assert_array_equal(problem.emitter_visibility, exact.emitter_visibility)
    problem.solve(max_order=3)
    exact.solve(max_order=3)
    assert problem.path_solves_attempted == sum(exact.counts.values())
    assert exact.counts["RRR"] > 10000 and exact.counts["RRD"] > 10000
    only_gpu, only_ref, worst = compare_problems(problem, exact)
    assert not only_gpu and not only_ref, f"only GPU: {sorted(only_gpu)[:5]}, only oracle: {sorted(only_ref)[:5]}"
    assert worst <= 1e-6
    assert len(problem.path_set()) >= 5


def test_saint_jean_order1_with_diffraction_matches_exact_oracle(places):
    """A 679-polygon scene (the grid-walk obstruction kernel is the one chosen at this size): reflections and
    diffractions of order 1 for four receivers against the oracle with exact roots."""
    from oracle import ray_tracing_oracle as ro
    from raytracing_amd import geometry as geom
    from raytracing_amd import scenarios
    from raytracing_amd.ray_tracing import RayTracingProblem

    place = geom.load_place("data/place_saint_jean/place_saint_jean.geojson")
    lo, hi = place.get_domain()
    tx = scenarios.emitter_outside(place, (lo[0] + hi[0]) / 2, (lo[1] + hi[1]) / 2, 30.0)
    rx = scenarios.lattice_receivers(place, 2, 1.5)
    place.add_set_of_points(rx)
    problem = RayTracingProblem(tx.reshape(1, 3), place)
    exact = ro.RayTracingOracle(tx, oracle_polys(place), rx, visibility_matrix=problem.visibility_matrix, solver="exact")
    np.testing.assert_array_equal(problem.emitter_visibility, exact.emitter_visibility)
    assert [k for k in problem.sharp_edges.keys()] == [k for k, _ in exact.sharp_edges]
    assert {r: len(v) for r, v in problem.los.items()} == {r: len(v) for r, v in exact.los.items()}
    problem.solve(max_order=1)
    exact.solve(max_order=1)
    assert problem.path_solves_attempted == sum(exact.counts.values())
    only_gpu, only_ref, worst = compare_problems(problem, exact)
    assert not only_gpu and not only_ref, f"only GPU: {sorted(only_gpu)[:5]}, only oracle: {sorted(only_ref)[:5]}"
    assert worst <= 1e-6
    assert len(problem.path_set()) >= 8


def test_many_receivers_in_one_call(places):
    """300 000 receivers in one call (more than one launch can tile): the library cuts the batch into chunks; every
    receiver's paths are those of the same receiver solved in a small batch."""
    from raytracing_amd import _lib
    from raytracing_amd import geometry as geom

    place = places("small")
    polygons = place.get_polygons_list()
    scene = _lib.Scene(geom.scene_arrays_of(polygons))
    tx, _ = small_tx_rx(1)
    pairs, edges = scene.sharp_edges()
    vis = scene.visibility_matrix(False)
    cube = geom.Cube.by_point_and_side_length(tx, 0.2)
    ev = np.logical_or.reduce(scene.faces_visibility([(s.points, s.get_parametric(), -1) for s in cube.polygons], False))
    cand = scene.enumerate(vis, ev, pairs, edges, 1, with_diffraction=True)
    rng = np.random.default_rng(9)
    lo, hi = place.get_domain()
    n = 300_000
    rx = np.column_stack([rng.uniform(lo[0], hi[0], n), rng.uniform(lo[1], hi[1], n), np.full(n, 1.5)])
    res = scene.solve_paths(cand, tx, rx)
    assert res["n_attempted"] == n * (1 + cand.total)
    picks = np.concatenate([np.arange(0, 50), np.arange(262_100, 262_200), np.arange(n - 50, n)])
    sub = scene.solve_paths(cand, tx, rx[picks])
    want = {(int(picks[r]), int(g), int(c)) for r, g, c in zip(sub["rx"], sub["group"], sub["cand"])}
    chosen = np.isin(res["rx"], picks)
    got = {(int(r), int(g), int(c)) for r, g, c in zip(res["rx"][chosen], res["group"][chosen], res["cand"][chosen])}
    assert got == want and len(want) > 100
    cand.close()
    scene.close()


@pytest.mark.parametrize("name,order,n_side,diffraction", [("place_levant", 2, 40, True), ("ny", 3, 14, False),
                                                           ("lln", 2, 30, False)])
def test_tiled_reflection_kernel_finds_the_same_paths(places, name, order, n_side, diffraction):
    """Large batches sort the receivers into spatial tiles and reject whole tiles with an interval form of the
    pre-test; the paths (and their points, bit for bit) must be those of the one-by-one loop.  Receivers in a
    scrambled order, some far outside the scene, a count that is no multiple of the tile size."""
    from raytracing_amd import _lib
    from raytracing_amd import geometry as geom
    from raytracing_amd import scenarios

    place = places(name)
    polygons = place.get_polygons_list()
    lo, hi = place.get_domain()
    tx = scenarios.emitter_outside(place, (lo[0] + hi[0]) / 2, (lo[1] + hi[1]) / 2, 25.0)
    rng = np.random.default_rng(3)
    rx = scenarios.lattice_receivers(place, n_side, 1.5)
    far = np.column_stack([rng.uniform(lo[0] - 500, hi[0] + 500, 37), rng.uniform(lo[1] - 500, hi[1] + 500, 37),
                           rng.uniform(0.5, 60.0, 37)])
    rx = np.concatenate([rx, far])[rng.permutation(len(rx) + 37)][:len(rx) + 30]
    scene = _lib.Scene(geom.scene_arrays_of(polygons))
    pairs, edges = scene.sharp_edges()
    vis = scene.visibility_matrix(False)
    cube = geom.Cube.by_point_and_side_length(tx, 0.2)
    ev = np.logical_or.reduce(scene.faces_visibility([(s.points, s.get_parametric(), -1) for s in cube.polygons], False))
    cand = scene.enumerate(vis, ev, pairs, edges, order, with_diffraction=diffraction)

    def as_set(res):
        return {(int(r), int(g), int(c)) + tuple(p[:3 * max(cand.group_kind(int(g))[0], 0) if g >= 0 else 0].ravel().tolist())
                for r, g, c, p in zip(res["rx"], res["group"], res["cand"], res["points"].reshape(len(res["rx"]), -1))}

    scene.set_reflect_mode(1)
    one_by_one = scene.solve_paths(cand, tx, rx)
    scene.set_reflect_mode(2)
    tiled = scene.solve_paths(cand, tx, rx)
    again = scene.solve_paths(cand, tx, rx)
    scene.set_reflect_mode(0)
    assert tiled["n_attempted"] == one_by_one["n_attempted"]
    n_reflected = int(np.sum((one_by_one["group"] >= 0) & (one_by_one["group"] < order)))
    assert n_reflected > 100, n_reflected
    assert as_set(tiled) == as_set(one_by_one)
    assert as_set(again) == as_set(one_by_one)
    cand.close()
    scene.close()


@pytest.mark.parametrize("name,order,n_side,diffraction,reference_solver",
                         [("small", 3, 6, True, False), ("ny", 3, 12, False, False), ("place_levant", 2, 20, True, True),
                          ("ny", 4, 5, False, False)])
def test_deepest_level_formed_in_the_kernels_gives_the_stored_level(places, name, order, n_side, diffraction,
                                                                    reference_solver):
    """The deepest reflection level is the one whose chain list grows out of hand (ray_tracing.py:316-322 visits it by
    recursion and never holds it).  Not stored, its chains are formed inside the kernels from the level before: the
    candidates, their indices and order, the paths found (points bit for bit), their description and the paths the
    reference-compatible mode keeps must all be those of the stored level.  Both kernels (one by one, tiled)."""
    from raytracing_amd import _lib
    from raytracing_amd import geometry as geom
    from raytracing_amd import scenarios

    place = places(name)
    polygons = place.get_polygons_list()
    lo, hi = place.get_domain()
    tx = scenarios.emitter_outside(place, (lo[0] + hi[0]) / 2, (lo[1] + hi[1]) / 2, 25.0)
    rx = scenarios.lattice_receivers(place, n_side, 1.5)
    scene = _lib.Scene(geom.scene_arrays_of(polygons))
    pairs, edges = scene.sharp_edges()
    vis = scene.visibility_matrix(False)
    cube = geom.Cube.by_point_and_side_length(tx, 0.2)
    ev = np.logical_or.reduce(scene.faces_visibility([(s.points, s.get_parametric(), -1) for s in cube.polygons], False))
    scene.set_reference_solver(reference_solver)

    def run(limit):
        scene.set_stored_chain_limit(limit)
        cand = scene.enumerate(vis, ev, pairs, edges, order, with_diffraction=diffraction)
        out = []
        for mode in (1, 2):
            scene.set_reflect_mode(mode)
            res = scene.solve_paths(cand, tx, rx)
            keep = res["group"] >= 0
            o = np.lexsort((res["cand"], res["group"], res["rx"]))
            desc = cand.describe(res["group"][keep], res["cand"][keep])
            # only the first (reflections + diffraction) points of a path are defined
            n_pts = np.array([sum(cand.group_kind(int(g))) if g >= 0 else 0 for g in res["group"]])
            pts = np.where(np.arange(_lib.RT_MAX_ORDER)[None, :, None] < n_pts[:, None, None], res["points"], 0.0)
            out.append((res["n_attempted"], res["rx"][o], res["group"][o], res["cand"][o], pts[o], desc))
        scene.set_reflect_mode(0)
        top = cand.group(order - 1)[0].copy()
        counts = list(cand.counts)
        cand.close()
        return out, top, counts

    stored, top_stored, counts_stored = run(1 << 40)
    formed, top_formed, counts_formed = run(0)
    scene.set_stored_chain_limit(256 << 20)
    assert counts_formed == counts_stored
    assert top_formed.shape == top_stored.shape and top_formed.shape[1] == order and len(top_formed) > 500
    np.testing.assert_array_equal(top_formed, top_stored)
    for a, b in zip(stored, formed):
        assert a[0] == b[0]
        for x, y in zip(a[1:5], b[1:5]):
            np.testing.assert_array_equal(x, y)
        for x, y in zip(a[5], b[5]):
            np.testing.assert_array_equal(x, y)
    n_deep = int(np.sum(stored[0][2] == order - 1))
    assert n_deep > 0, "no path of the deepest level: the test shows nothing"
    scene.close()


@pytest.mark.parametrize("which", ["flat", "slanted"])
def test_levant_study_scenes_match_the_exact_oracle(which, tmp_path, monkeypatch):
    """The study scenes of place_utils.py:199-327 — flat ground with tree trunks; three grounds with a slope, the
    buildings rebuilt on them, some in several pieces — solved to order 2 with diffraction: visibility matrix, emitter
    visibility and sharp edges as the oracle finds them, the path set of the exact-root oracle, and the field step
    with the tree crowns added afterwards within 0.01 dB of the oracle's, path by path."""
    from oracle import em_oracle as eo
    from oracle import ray_tracing_oracle as ro
    from raytracing_amd import electromagnetism as em
    from raytracing_amd import place_utils as pu
    from raytracing_amd.ray_tracing import RayTracingProblem

    monkeypatch.setenv("RAYTRACING_AMD_RESULTS", str(tmp_path))
    build = pu.Place_du_levant.create_flat_levant if which == "flat" else pu.Place_du_levant.create_slanted_levant
    place, tx, name = build(npoints=6)
    rx = place.set_of_points.copy()

    def with_buildings(polys):
        ids = {}
        for p, q in zip(polys, place.get_polygons_list()):
            p.building = ids.setdefault(q.building_id, len(ids))
        return polys

    problem = RayTracingProblem(tx, place)
    solved_with = with_buildings(oracle_polys(place))
    oracle = ro.RayTracingOracle(tx.reshape(3), solved_with, rx, solver="exact")
    np.testing.assert_array_equal(problem.visibility_matrix, oracle.visibility_matrix)
    np.testing.assert_array_equal(problem.emitter_visibility, oracle.emitter_visibility)
    assert [k for k in problem.sharp_edges.keys()] == [k for k, _ in oracle.sharp_edges]
    problem.solve(max_order=2)
    oracle.solve(max_order=2)
    assert problem.path_solves_attempted == sum(oracle.counts.values())
    only_gpu, only_ref, worst = compare_problems(problem, oracle)
    assert not only_gpu and not only_ref, f"only GPU: {sorted(only_gpu)[:5]}, only oracle: {sorted(only_ref)[:5]}"
    assert worst <= 1e-6
    assert len(problem.path_set()) > 20

    df = em.my_field_computation(problem, name)
    assert len(place.get_polygons_list()) == len(solved_with) + 3 * 6  # the three crowns
    rows = eo.FieldOracle(with_buildings(oracle_polys(place))).field_computation(oracle, rx)
    assert list(df["path_type"]) == [r["path_type"] for r in rows]
    for (_, got), want in zip(df.iterrows(), rows):
        if want["path_type"] != "Impossible":
            assert abs(10 * np.log10(got["path_power"]) - 10 * np.log10(want["path_power"])) <= 0.01


def test_odd_receiver_count_goes_in_one_chunk_and_keeps_the_next_solve_sized(places):
    """A batch on receiver tiles whose count is no multiple of the tile size used to be cut into whole tiles plus a
    second chunk of the few receivers left over, and the sizes of that tiny chunk then steered the launches of the next
    solve (queues too small, the chunk redone, the one-launch obstruction route on millions of segments: a step of
    the 5 041-receiver lln map took 320 ms instead of 7).  One chunk, and the same launches solve after solve."""
    from raytracing_amd import _lib
    from raytracing_amd import geometry as geom
    from raytracing_amd import scenarios

    place = places("place_levant")
    polygons = place.get_polygons_list()
    lo, hi = place.get_domain()
    tx = scenarios.emitter_outside(place, (lo[0] + hi[0]) / 2, (lo[1] + hi[1]) / 2, 25.0)
    scene = _lib.Scene(geom.scene_arrays_of(polygons))
    pairs, edges = scene.sharp_edges()
    vis = scene.visibility_matrix(False)
    cube = geom.Cube.by_point_and_side_length(tx, 0.2)
    ev = np.logical_or.reduce(scene.faces_visibility([(s.points, s.get_parametric(), -1) for s in cube.polygons], False))
    cand = scene.enumerate(vis, ev, pairs, edges, 2, with_diffraction=True)
    scene.set_reflect_mode(2)  # receiver tiles, as large batches use them
    launches = {}
    for n_side in (48, 47):  # 2 304 = 144 tiles; 2 209 = 138 tiles and one receiver
        rx = scenarios.lattice_receivers(place, n_side, 1.5)
        assert (len(rx) % 16 == 0) == (n_side == 48)
        results = [scene.solve_paths(cand, tx, rx) for _ in range(3)]
        seen = [scene.last_timing()["launches"] for _ in range(1)]
        launches[n_side] = seen[0]
        for r in results[1:]:
            assert r["n_attempted"] == results[0]["n_attempted"]
            np.testing.assert_array_equal(np.sort(r["cand"]), np.sort(results[0]["cand"]))
    scene.set_reflect_mode(0)
    assert launches[47] == launches[48], launches  # one chunk each: the same number of launches of every family
    cand.close()
    scene.close()
