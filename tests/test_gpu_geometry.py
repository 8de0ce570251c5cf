"""GPU parity of the geometry entry points against the CPU oracle, through the C ABI (ctypes)."""
import numpy as np
import pytest

from conftest import oracle_polys, small_tx_rx

pytestmark = pytest.mark.gpu


def random_segments(place, n, seed):
    rng = np.random.default_rng(seed)
    lo, hi = place.get_domain()
    lo = lo - [5, 5, 0]
    hi = hi + [5, 5, 10]
    a = rng.uniform(lo, hi, size=(n, 3))
    b = rng.uniform(lo, hi, size=(n, 3))
    return np.concatenate([a, b], axis=1)


@pytest.mark.parametrize("name,n", [("small", 400), ("place_levant", 300)])
def test_obstruction_matches_oracle(places, name, n):
    from oracle import geometry_oracle as go
    from raytracing_amd import geometry as geom

    place = places(name)
    polys = oracle_polys(place)
    scene = geom.scene_arrays_of(place.get_polygons_list()).device_scene()
    segs = random_segments(place, n, 1)
    got = scene.obstructed(segs)
    want = np.array([go.polygons_obstruct_line_path(polys, s.reshape(2, 3)) for s in segs])
    assert got.sum() > 10 and (~got).sum() > 10
    np.testing.assert_array_equal(got, want)
    # masked form (what the line-of-sight step uses)
    mask = np.zeros(len(polys), dtype=bool)
    mask[::3] = True
    got = scene.obstructed(segs, mask=mask)
    sub = [p for p, m in zip(polys, mask) if m]
    want = np.array([go.polygons_obstruct_line_path(sub, s.reshape(2, 3)) for s in segs])
    np.testing.assert_array_equal(got, want)
    # every-polygon form
    hits = scene.segment_hits(segs[:40])
    for s, h in zip(segs[:40], hits):
        assert list(np.flatnonzero(h)) == go.polygons_obstruct_line_path(polys, s.reshape(2, 3), return_indices=True)


def awkward_segments(place, n, seed):
    """Segments that stress the grid walk: axis-aligned, vertical, along cell borders, far outside the scene, very
    long, very short, ending on polygon vertices."""
    rng = np.random.default_rng(seed)
    lo, hi = place.get_domain()
    ext = hi - lo
    segs = [random_segments(place, n, seed)]
    a = rng.uniform(lo, hi + [0, 0, 10], size=(n, 3))
    for axis in range(3):  # axis-aligned
        b = a.copy()
        b[:, axis] = rng.uniform(lo[axis] - 5, hi[axis] + 15, size=n)
        segs.append(np.concatenate([a, b], axis=1))
    far = a + rng.normal(size=(n, 3)) * ext * 20  # one end far outside the scene
    segs.append(np.concatenate([far, a], axis=1))
    segs.append(np.concatenate([a + ext * 30, far + ext * 30], axis=1))  # entirely outside
    tiny = a + rng.normal(size=(n, 3)) * 1e-3
    segs.append(np.concatenate([a, tiny], axis=1))
    verts = np.concatenate([p.points for p in place.get_polygons_list()])
    pick = verts[rng.integers(0, len(verts), size=n)]
    segs.append(np.concatenate([a, pick], axis=1))  # ending on a polygon vertex
    segs.append(np.concatenate([pick + [0, 0, 30.0], pick - [0, 0, 30.0]], axis=1))  # down a building edge
    pick2 = verts[rng.integers(0, len(verts), size=n)]
    segs.append(np.concatenate([pick2 + [0, 0, 0.5], pick + [0, 0, 0.5]], axis=1))  # vertex to vertex, lifted
    return np.concatenate(segs)


@pytest.mark.parametrize("name,n", [("small", 300), ("place_levant", 500), ("place_saint_jean/place_saint_jean", 2000),
                                    ("lln", 3000)])
def test_obstruction_grid_walk_equals_every_polygon_walk(places, name, n):
    """The obstruction kernel walks a uniform grid; the flags must be those of the walk over every polygon."""
    from raytracing_amd import geometry as geom

    place = places(name)
    scene = geom.scene_arrays_of(place.get_polygons_list()).device_scene()
    segs = awkward_segments(place, n, 7)
    mask = np.zeros(scene.n, dtype=bool)
    mask[::2] = True
    try:
        scene.set_occlusion_mode(1)
        want, want_masked = scene.obstructed(segs), scene.obstructed(segs, mask=mask)
        scene.set_occlusion_mode(2)
        got, got_masked = scene.obstructed(segs), scene.obstructed(segs, mask=mask)
    finally:
        scene.set_occlusion_mode(0)
    assert want.sum() > n // 10 and (~want).sum() > n // 10
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(got_masked, want_masked)


def test_obstruction_api_and_edge_cases(places):
    from raytracing_amd import geometry as geom

    place = places("small")
    polygons = place.get_polygons_list()
    scene = geom.scene_arrays_of(polygons).device_scene()
    assert scene.obstructed(np.zeros((0, 6))).shape == (0,)
    # a vertical segment through the first roof, and one that stops short of it
    top = polygons[1]
    c = top.points.mean(axis=0)
    assert geom.polygons_obstruct_line_path(polygons, np.array([c + [0, 0, 5], c - [0, 0, 5]]))
    assert not geom.polygons_obstruct_line_path(polygons, np.array([c + [0, 0, 5], c + [0, 0, 1]]))
    # a segment that ends exactly on the roof plane is not obstructed by it (1e-8 < t < 1 - 1e-8)
    assert not geom.polygons_obstruct_line_path([top], np.array([c + [0, 0, 5], c]))
    groups = geom.polygons_obstruct_line_path(polygons, np.array([c + [0, 0, 5], c - [0, 0, 5]]), return_polygons=True)
    assert len(groups) >= 1 and all(isinstance(g, list) for g in groups)


def test_contains_point_matches_oracle(places):
    from oracle import geometry_oracle as go
    from raytracing_amd import geometry as geom

    place = places("place_levant")
    polys = oracle_polys(place)
    scene = geom.scene_arrays_of(place.get_polygons_list()).device_scene()
    rng = np.random.default_rng(3)
    for idx in [0, 1, 2, 5, 40, 77, 150]:
        p = polys[idx]
        m = p.matrix()
        lo, hi = p.points.min(axis=0), p.points.max(axis=0)
        pts = rng.uniform(lo - 1, hi + 1, size=(200, 3))
        # put the points in the polygon's plane, a few slightly off it
        dist = pts @ p.normal + p.parametric[3]
        pts = pts - np.outer(dist, p.normal)
        pts[::10] += 1e-3 * p.normal
        # (the ring's own vertices and side middles are left to test_contains_point_includes_the_boundary, which builds
        # them where they are exact: on these polygons, whose frames are not axis-aligned, a point meant to lie on the
        # ring lands a rounding error to one side or the other of it, differently in a matrix product and in three
        # dot products, and 6 of 242 such points got different answers from the two sides)
        for check in (False, True):
            got = scene.contains_points(idx, pts, check, 1e-8)
            want = np.array([go.contains_point(p, q, check) for q in pts])
            np.testing.assert_array_equal(got, want)
            assert got.sum() > 5


def test_contains_point_includes_the_boundary(places):
    """The reference's test is shapely's Polygon.intersects(Point) (geometry.py:1803): a point ON the ring is inside.
    Boundary points are built in the polygon's own frame, where they are exact: rings whose in-plane coordinates are
    small integers and halves (every product and difference of the crossing test is then exact in binary64), the
    vertices themselves and the middles of the sides.  Outside points a hair off the sides stay outside."""
    from oracle import geometry_oracle as go
    from raytracing_amd import geometry as geom

    rings = [np.array([[0, 0], [8, 0], [8, 4], [0, 4]], dtype=float),
             np.array([[0, 0], [6, 0], [6, 2], [2, 2], [2, 6], [0, 6]], dtype=float),          # L shape (not convex)
             np.array([[-4, -2], [4, -2], [6, 2], [0, 6], [-6, 2]], dtype=float)]  # first side along an axis: exact frame
    planes = [("z", 2.0), ("x", -3.0), ("y", 0.0)]  # horizontal, and two walls: exact axis-aligned frames
    for ring in rings:
        for axis, level in planes:
            pts3 = {"z": np.column_stack([ring[:, 0], ring[:, 1], np.full(len(ring), level)]),
                    "x": np.column_stack([np.full(len(ring), level), ring[:, 0], ring[:, 1]]),
                    "y": np.column_stack([ring[:, 1], np.full(len(ring), level), ring[:, 0]])}[axis]
            polygon = geom.OrientedPolygon(pts3)
            oracle_poly = go.Poly(pts3)
            oracle_poly.parametric = polygon.get_parametric().copy()
            scene = geom.scene_arrays_of([polygon]).device_scene()
            mids = (pts3 + np.roll(pts3, -1, axis=0)) / 2
            quarter = pts3 + (np.roll(pts3, -1, axis=0) - pts3) / 4
            on_boundary = np.vstack([pts3, mids, quarter])
            for check in (False, True):
                got = scene.contains_points(0, on_boundary, check, 1e-8)
                assert got.all(), f"boundary points reported outside: {on_boundary[~got][:4].tolist()}"
                want = np.array([go.contains_point(oracle_poly, q, check) for q in on_boundary])
                assert want.all()
            # a hair outside each side (outward normal of a counter-clockwise / clockwise ring found from the centroid)
            centre = pts3.mean(axis=0)
            outward = mids - centre
            outside = mids + 1e-6 * outward / np.linalg.norm(outward, axis=1)[:, None]
            got = scene.contains_points(0, outside, False, 1e-8)
            want = np.array([go.contains_point(oracle_poly, q, False) for q in outside])
            np.testing.assert_array_equal(got, want)
            scene.close()


@pytest.mark.parametrize("name", ["small", "ny", "place_levant", "place_saint_jean/place_saint_jean"])
def test_visibility_matrix_matches_oracle(places, name):
    from oracle import visibility_oracle as vo
    from raytracing_amd import geometry as geom

    place = places(name)
    polys = oracle_polys(place)
    got = geom.polygons_visibility_matrix(place.get_polygons_list())
    want = vo.polygons_visibility_matrix(polys)
    assert got.dtype == bool and got.shape == want.shape
    assert not got.diagonal().any()
    np.testing.assert_array_equal(got, got.T)
    bad = np.argwhere(got != want)
    assert len(bad) == 0, f"{len(bad)} entries differ, first {bad[:10].tolist()}"


def test_emitter_visibility_matches_oracle(places):
    from oracle import geometry_oracle as go
    from oracle import visibility_oracle as vo
    from raytracing_amd import geometry as geom

    for name, tx in [("small", np.array([3.0, 38.0, 18.0])), ("place_levant", np.array([0.0, 0.0, 18.7]))]:
        place = places(name)
        polys = oracle_polys(place)
        polygons = place.get_polygons_list()
        cube_o = go.cube_by_point_and_side_length(tx, 0.2)
        cube_p = geom.Cube.by_point_and_side_length(tx, 0.2)
        for face_o, face_p in zip(cube_o, cube_p.polygons):
            np.testing.assert_array_equal(face_o.points, face_p.points)
            got = geom.polygon_visibility_vector(face_p, polygons)
            want = vo.polygon_visibility_vector(face_o, polys)
            np.testing.assert_array_equal(got, want)


def test_visibility_rows_reassemble_the_matrix(places):
    """The matrix is rows | rows^T whatever the split of the rows (what a node does over its GPUs)."""
    from raytracing_amd import geometry as geom
    from raytracing_amd.multithread_solve import gather_visibility_rows

    place = places("place_levant")
    polygons = place.get_polygons_list()
    scene = geom.scene_arrays_of(polygons).device_scene()
    full = scene.visibility_matrix(False)
    n = len(polygons)
    assert scene.visibility_rows(np.zeros(0, dtype=np.int32)).shape == (0, n)
    rows = np.zeros((n, n), dtype=bool)
    for part in (np.arange(0, n, 3), np.arange(1, n, 3), np.arange(2, n, 3)[::-1]):
        rows[part] = scene.visibility_rows(part, False)
    np.testing.assert_array_equal(rows | rows.T, full)
    np.testing.assert_array_equal(gather_visibility_rows(lambda ids: scene.visibility_rows(ids, False), n), full)
    with pytest.raises(Exception):
        scene.visibility_rows([n], False)


def test_visibility_rows_of_lln_match_oracle(places):
    """lln.geojson (7 615 polygons) is too large for the oracle's full matrix in a test; rows of faces spread over
    the scene (ground, roofs, walls) are compared instead, through the same entry point the matrix is built from."""
    from oracle import visibility_oracle as vo
    from raytracing_amd import geometry as geom

    place = places("lln")
    polygons = place.get_polygons_list()
    polys = oracle_polys(place)
    scene = geom.scene_arrays_of(polygons).device_scene()
    faces = [0, 1, 2, 3] + list(range(17, len(polygons), len(polygons) // 36))
    got = scene.faces_visibility([(polygons[i].points, polygons[i].get_parametric(), i) for i in faces], False)
    n_visible = 0
    packed = vo.PackedScene(polys)
    for row, i in zip(got, faces):
        want = vo.polygon_visibility_vector(polys[i], polys, self_index=i, packed=packed)
        bad = np.flatnonzero(row != want)
        assert len(bad) == 0, f"face {i}: {len(bad)} entries differ, first {bad[:10].tolist()}"
        n_visible += int(row.sum())
    assert n_visible > 1000


def test_every_row_of_the_lln_matrix_is_the_oracles(places):
    """All 7 615 rows (58 million entries) against tests/golden/lln_matrix_rows.npz, the oracle painter's rows made on
    the CPU by tests/golden/make_lln_matrix.py; the rows painted whole (one call with every face) and, for a sample,
    alone (a single face is painted in sectors)."""
    import os

    from conftest import GOLDEN
    from raytracing_amd import geometry as geom

    z = np.load(os.path.join(GOLDEN, "lln_matrix_rows.npz"))
    n = int(z["n"])
    want = np.unpackbits(z["bits"])[: n * n].reshape(n, n).astype(bool)
    place = places("lln")
    polygons = place.get_polygons_list()
    assert len(polygons) == n
    scene = geom.scene_arrays_of(polygons).device_scene()
    face = lambda i: (polygons[i].points, polygons[i].get_parametric(), i)  # noqa: E731
    got = np.asarray(scene.faces_visibility([face(i) for i in range(n)], False)).astype(bool)
    bad = np.argwhere(got != want)
    assert len(bad) == 0, f"{len(bad)} entries differ, first {bad[:10].tolist()}"
    assert int(got.sum()) == 428090
    for i in [5496, 6713, 12, 4000]:  # (the first two: a gap 3e-10 rad high decides whether one sees the other)
        alone = np.asarray(scene.faces_visibility([face(i)], False)[0]).astype(bool)
        np.testing.assert_array_equal(alone, want[i])


def test_every_share_of_the_lln_rows_is_the_oracles(places):
    """The rows of the lln matrix dealt to 2, 4 and 8 ranks, every rank's share painted on the one GPU with the tile
    rules the library chooses for a share of that size (its largest faces in tiles), against the oracle's rows
    (tests/golden/lln_matrix_rows.npz): all 58 million entries, three times over."""
    import os

    import torch

    from conftest import GOLDEN
    from raytracing_amd import geometry as geom
    from raytracing_amd import multithread_solve as mts

    z = np.load(os.path.join(GOLDEN, "lln_matrix_rows.npz"))
    n = int(z["n"])
    want = np.unpackbits(z["bits"])[: n * n].reshape(n, n).astype(bool)
    scene = geom.scene_arrays_of(places("lln").get_polygons_list()).device_scene()
    assert scene.n == n
    for world in (2, 4, 8):
        for rank in range(world):
            block = mts.paint_dealt_block_on_device(scene, [], False, world, rank)
            torch.cuda.synchronize()
            mine = np.arange(rank, n, world)
            got = block.cpu().numpy()[: len(mine)].astype(bool)
            bad = np.argwhere(got != want[mine])
            assert len(bad) == 0, f"world {world} rank {rank}: {len(bad)} entries differ, first {bad[:10].tolist()}"


def test_small_tiles_on_a_share_of_the_lln_rows(places):
    """An eighth of lln's rows with every face of 512 candidates and more forced into tiles of 256 (some 14 000 tiles,
    whose borders are examined like edges): the oracle's rows, painting after painting.  Before the phases of a full
    decision wrote flag values of their own, one such painting in a dozen came out wrong or did not end."""
    import os

    import torch

    from conftest import GOLDEN
    from raytracing_amd import geometry as geom
    from raytracing_amd import multithread_solve as mts

    z = np.load(os.path.join(GOLDEN, "lln_matrix_rows.npz"))
    n = int(z["n"])
    want = np.unpackbits(z["bits"])[: n * n].reshape(n, n).astype(bool)
    scene = geom.scene_arrays_of(places("lln").get_polygons_list()).device_scene()
    mine = np.arange(0, n, 8)
    try:
        scene.set_visibility_tiling(512, 256)
        for rep in range(4):
            block = mts.paint_dealt_block_on_device(scene, [], False, 8, 0)
            torch.cuda.synchronize()
            got = block.cpu().numpy()[: len(mine)].astype(bool)
            bad = np.argwhere(got != want[mine])
            assert len(bad) == 0, f"painting {rep}: {len(bad)} entries differ, first {bad[:10].tolist()}"
    finally:
        scene.set_visibility_tiling(-1, 0)


def test_faces_whose_canvas_outgrows_the_resident_store():
    """ny.geojson tiled 8 x 8: its large faces see five hundred polygons each, more than the four-wave painter's resident
    canvas holds (320), so what is left of such a face is decided one candidate at a time, with the covered-cell bitmap
    still in front.  The same rows painted 128 at a time go through the sixteen-wave painter instead, in tiles, whose
    canvas holds them: the two must agree entry for entry."""
    from raytracing_amd import geometry as geom
    from raytracing_amd import scenarios

    scene = geom.scene_arrays_of(scenarios.ny_tiled(8)[0].get_polygons_list()).device_scene()
    n = scene.n
    rows = np.unique(np.concatenate([np.arange(0, n, 17), [733, 1461, 2189, 2857, 2917, 3585, 3736, 4860]])).astype(np.int32)
    assert len(rows) > 256
    try:
        scene.set_visibility_tiling(0, 0)  # every face whole: the large ones outgrow the store
        whole = np.asarray(scene.visibility_rows(rows)).astype(bool)
    finally:
        scene.set_visibility_tiling(-1, 0)
    assert whole.sum(axis=1).max() > 320, "no row of this choice outgrows the resident store"
    for k0 in range(0, len(rows), 128):
        part = np.asarray(scene.visibility_rows(rows[k0:k0 + 128])).astype(bool)
        bad = np.argwhere(part != whole[k0:k0 + 128])
        assert len(bad) == 0, f"rows {k0}..: {len(bad)} entries differ, first {bad[:10].tolist()}"


@pytest.mark.parametrize("name", ["small", "ny", "place_levant", "place_saint_jean/place_saint_jean"])
def test_sharp_edges_match_oracle(places, name):
    from oracle import geometry_oracle as go
    from raytracing_amd import geometry as geom

    place = places(name)
    polys = oracle_polys(place)
    got = list(geom.polygons_sharp_edges_iter(place.get_polygons_list()))
    want = go.polygons_sharp_edges(polys)
    assert [(i, j) for _, i, j in got] == [(i, j) for _, i, j in want]
    for (eg, _, _), (ew, _, _) in zip(got, want):
        np.testing.assert_array_equal(eg, ew)
    assert len(got) > 10


def test_standalone_solvers_match_oracle():
    from oracle import geometry_oracle as go
    from raytracing_amd import geometry as geom

    cube = go.cube_by_point_and_side_length(np.zeros(3), 20.0)
    A = np.array([0.0, 0.0, 0.0])
    B = np.array([6.0, 6.0, 6.0])
    rng = np.random.default_rng(5)
    n_checked = 0
    for k in (1, 2, 3):
        for _ in range(40):
            faces = rng.choice(6, size=k, replace=True)
            if any(faces[i] == faces[i + 1] for i in range(k - 1)):
                continue
            planes = [cube[f].parametric for f in faces]
            a = A + rng.uniform(-8, 8, 3)
            b = B + rng.uniform(-3, 3, 3)
            x_gpu, sol_gpu = geom.reflexion_points_from_origin_destination_and_planes(a, b, planes)
            x_ref, sol_ref = go.reflexion_points_from_origin_destination_and_planes(a, b, planes)
            x_exact, sol_exact = go.exact_reflexion_points(a, b, planes)
            assert sol_gpu.success == sol_exact.success
            np.testing.assert_allclose(x_gpu, x_exact, atol=1e-9, rtol=0)
            if sol_ref.success and np.abs(sol_ref.fun).max() < 1e-9:  # hybr found a root: it is the same one
                assert sol_gpu.success
                np.testing.assert_allclose(x_gpu, x_ref, atol=1e-6, rtol=0)  # metres
                n_checked += 1
    assert n_checked > 40
    # diffraction and reflection + diffraction (the reflections aim at the edge's first endpoint)
    edge = np.array([[10.0, 10.0, -10.0], [10.0, 10.0, 10.0]])
    for _ in range(30):
        a = rng.uniform(-8, 8, 3)
        b = np.array([30.0, 12.0, 5.0]) + rng.uniform(-5, 5, 3)
        x_gpu, sol_gpu = geom.diffraction_point_from_origin_destination_and_edge(a, b, edge)
        x_ref, sol_ref = go.diffraction_point_from_origin_destination_and_edge(a, b, edge)
        assert sol_gpu.success and sol_ref.success
        np.testing.assert_allclose(x_gpu.reshape(-1), np.asarray(x_ref).reshape(-1), atol=1e-6, rtol=0)
        planes = [cube[0].parametric]
        x_gpu, sol_gpu = geom.reflexion_points_and_diffraction_point_from_origin_destination_planes_and_edge(a, b, planes, edge)
        x_ref, sol_ref = go.reflexion_points_and_diffraction_point_from_origin_destination_planes_and_edge(a, b, planes, edge)
        if sol_ref.success:
            np.testing.assert_allclose(x_gpu, x_ref, atol=1e-6, rtol=0)
    # no specular path: both points on the same side is fine, opposite sides is not
    plane = [np.array([0.0, 0.0, 1.0, 0.0])]
    _, sol = geom.reflexion_points_from_origin_destination_and_planes(np.array([0, 0, 1.0]), np.array([1.0, 0, -1.0]), plane)
    assert not sol.success
    _, sol = geom.reflexion_points_from_origin_destination_and_planes(np.array([0, 0, 1.0]), np.array([1.0, 0, 2.0]), plane)
    assert sol.success


@pytest.mark.parametrize("name", ["small", "place_levant"])
def test_matrix_and_screens_in_one_call_equal_the_separate_calls(places, name):
    """The visibility matrix (or some of its rows) and the emitter's screens painted side by side on two streams are
    the matrix, the rows and the screen vectors of the separate entry points."""
    from raytracing_amd import _lib
    from raytracing_amd import geometry as geom
    from raytracing_amd import scenarios

    place = places(name)
    polygons = place.get_polygons_list()
    lo, hi = place.get_domain()
    tx = scenarios.emitter_outside(place, (lo[0] + hi[0]) / 2, (lo[1] + hi[1]) / 2, 20.0)
    cube = geom.Cube.by_point_and_side_length(tx, 0.2)
    screens = [(s.points, s.get_parametric(), -1) for s in cube.polygons]
    scene = _lib.Scene(geom.scene_arrays_of(polygons))
    matrix = scene.visibility_matrix(False)
    vectors = scene.faces_visibility(screens, False)
    assert vectors.any() and matrix.any()
    for _ in range(2):  # the side stream is made on the first call and reused
        m2, v2 = scene.visibility_with_faces(screens)
        np.testing.assert_array_equal(m2, matrix)
        np.testing.assert_array_equal(v2, vectors)
    ids = np.arange(1, len(polygons), 3)
    r2, v3 = scene.visibility_with_faces(screens[:2], row_ids=ids)
    np.testing.assert_array_equal(r2, scene.visibility_rows(ids))
    np.testing.assert_array_equal(v3, vectors[:2])
    r0, v4 = scene.visibility_with_faces(screens, row_ids=np.zeros(0, np.int32))
    assert r0.shape == (0, len(polygons))
    np.testing.assert_array_equal(v4, vectors)
    # the matrix left in device memory is the same matrix, and the candidates enumerated from it there are those
    # enumerated from the host's copy
    import torch

    n = len(polygons)
    d_matrix = torch.full((n, n), 7, dtype=torch.uint8, device=f"cuda:{scene.device}")
    m5, v5 = scene.visibility_with_faces(screens, matrix_out=d_matrix)
    assert m5 is d_matrix
    np.testing.assert_array_equal(d_matrix.cpu().numpy().astype(bool), matrix)
    np.testing.assert_array_equal(v5, vectors)
    ev = np.logical_or.reduce(vectors)
    pairs, edges = scene.sharp_edges()
    from_host = scene.enumerate(matrix, ev, pairs, edges, 2)
    from_device = scene.enumerate(d_matrix, ev, pairs, edges, 2)
    np.testing.assert_array_equal(from_device.counts, from_host.counts)
    assert from_host.counts.sum() > 0
    for g in range(from_host.n_groups):
        a, b = from_host.group(g), from_device.group(g)
        np.testing.assert_array_equal(b[0], a[0])
        np.testing.assert_array_equal(b[1], a[1])
    with pytest.raises(ValueError):
        scene.visibility_with_faces(screens, row_ids=ids, matrix_out=d_matrix)
    with pytest.raises(ValueError):
        scene.enumerate(d_matrix[:-1], ev, pairs, edges, 2)
    scene.close()


@pytest.mark.parametrize("name,world", [("small", 1), ("small", 3), ("place_levant", 2), ("place_levant", 8),
                                        ("place_levant", 5), ("lln", 8)])
def test_blocks_painted_on_the_device_rank_by_rank_give_the_matrix_and_the_screens(places, name, world):
    """What each rank of a sharded run paints straight into its all-gather buffer (its matrix rows, then its screens,
    device memory), for every rank of a world in turn on the one GPU: the blocks stacked as the collective would leave
    them, put in order and symmetrised, are the matrix and the screen vectors of the single call."""
    import torch

    from raytracing_amd import _lib
    from raytracing_amd import geometry as geom
    from raytracing_amd import multithread_solve as mts
    from raytracing_amd import scenarios

    place = places(name)
    polygons = place.get_polygons_list()
    lo, hi = place.get_domain()
    tx = scenarios.emitter_outside(place, (lo[0] + hi[0]) / 2, (lo[1] + hi[1]) / 2, 20.0)
    cube = geom.Cube.by_point_and_side_length(tx, 0.2)
    screens = [(s.points, s.get_parametric(), -1) for s in cube.polygons]
    scene = _lib.Scene(geom.scene_arrays_of(polygons))
    n = len(polygons)
    want_matrix, want_screens = scene.visibility_with_faces(screens)
    blocks = [mts.paint_dealt_block_on_device(scene, screens, False, world, r) for r in range(world)]
    assert all(b.is_cuda and b.shape == blocks[0].shape for b in blocks)
    rows = mts.dealt_rows_in_order(torch.stack(blocks), n + len(screens))
    matrix = rows[:n] | rows[:n].T
    np.testing.assert_array_equal(matrix.cpu().numpy().astype(bool), want_matrix)
    np.testing.assert_array_equal(rows[n:].cpu().numpy().astype(bool), want_screens)
    # the padding rows of the last blocks stay zero
    pad = blocks[0].shape[0]
    for r, b in enumerate(blocks):
        used = len(range(r, n + len(screens), world))
        assert not b[used:pad].any()
    scene.close()


@pytest.mark.parametrize("name,per_tile", [("small", 3), ("ny", 5), ("place_levant", 7), ("place_levant", 40),
                                            ("place_saint_jean/place_saint_jean", 24)])
def test_faces_painted_in_sectors_see_what_faces_painted_whole_see(places, name, per_tile):
    """A face with very many candidates is painted in azimuth sectors by different workgroups (a polygon is visible iff
    it shows in some sector).  With the threshold forced down every face of these scenes is cut into sectors of a few
    candidates each: the matrix and the emitter's screens must be those of faces painted whole."""
    from raytracing_amd import _lib
    from raytracing_amd import geometry as geom
    from raytracing_amd import scenarios

    place = places(name)
    polygons = place.get_polygons_list()
    lo, hi = place.get_domain()
    tx = scenarios.emitter_outside(place, (lo[0] + hi[0]) / 2, (lo[1] + hi[1]) / 2, 20.0)
    cube = geom.Cube.by_point_and_side_length(tx, 0.2)
    screens = [(s.points, s.get_parametric(), -1) for s in cube.polygons]
    scene = _lib.Scene(geom.scene_arrays_of(polygons))
    scene.set_visibility_tiling(0, 0)
    whole_matrix, whole_screens = scene.visibility_with_faces(screens)
    scene.set_visibility_tiling(2 * per_tile, per_tile)
    matrix, vectors = scene.visibility_with_faces(screens)
    bad = np.argwhere(matrix != whole_matrix)
    assert len(bad) == 0, f"{len(bad)} matrix entries differ, first {bad[:10].tolist()}"
    bad = np.argwhere(vectors != whole_screens)
    assert len(bad) == 0, f"{len(bad)} screen entries differ, first {bad[:10].tolist()}"
    assert whole_matrix.sum() > 50 and whole_screens.sum() > 10
    scene.close()
