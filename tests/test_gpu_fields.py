"""GPU parity of the field step: against vectors produced by the reference's own field code, and against the CPU
oracle on a solved scene.  Tolerances: received power 0.01 dB (the bar of the port), fields 1e-9 relative."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, oracle_polys, small_tx_rx

pytestmark = pytest.mark.gpu

DB_TOL = 0.01


def db(x):
    return 10 * np.log10(x)


class CubeProblem:
    """Just enough of a RayTracingProblem for the per-step entry points: polygons and their device scene."""

    def __init__(self):
        from raytracing_amd import geometry as geom
        from raytracing_amd.materials_properties import set_properties

        cube = geom.Cube.by_point_and_side_length(np.zeros(3), 20.0)
        for p in cube.polygons:
            p.properties = set_properties("appartments")
        self.polygons = cube.polygons
        self._scene = geom.scene_arrays_of(self.polygons).device_scene()

    def _device_scene(self):
        return self._scene


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLDEN, "em.npz"))


def test_source_field_matches_reference_vectors(g):
    from raytracing_amd import electromagnetism as em

    for p, t, q, mat, E in zip(g["ant_pos"], g["ant_target"], g["ant_probe"], g["ant_mat"], g["src_field"]):
        a = em.Antenna(p)
        a.align_antenna_to_point(t)
        np.testing.assert_array_equal(a.mat, mat)
        got = em.ElectromagneticField.from_path([p, q], a)
        np.testing.assert_allclose(got, E, rtol=1e-9, atol=1e-12 * np.abs(E).max())


def test_reflection_matches_reference_vectors(g):
    from raytracing_amd import electromagnetism as em

    rtp = CubeProblem()
    ant = em.Antenna(g["refl_tx"])
    ant.align_antenna_to_point(g["refl_target"])
    for path, E in zip(g["refl_paths"], g["refl_E"]):
        E_i = em.ElectromagneticField.from_path([path[0], path[1]], ant)
        got = em.Reflection.my_reflect(E_i, path, [2], rtp)
        np.testing.assert_allclose(got, E, rtol=1e-9, atol=1e-12 * np.abs(E).max())


def test_wedge_diffraction_matches_reference_vectors(g):
    from raytracing_amd import electromagnetism as em

    rtp = CubeProblem()
    assert len(g["diff_paths"]) >= 30
    for path, E, E_i in zip(g["diff_paths"], g["diff_E"], g["diff_Ei"]):
        got = em.Diffraction.my_diff(E_i, path, (2, 3), g["diff_corner"], rtp, 0)
        np.testing.assert_allclose(got, E, rtol=1e-8, atol=1e-11 * np.abs(E).max())


def _solved_small(n_rx):
    from oracle import em_oracle as eo
    from oracle import ray_tracing_oracle as ro
    from raytracing_amd import geometry as geom
    from raytracing_amd.ray_tracing import RayTracingProblem

    place = geom.load_place("data/small.geojson")
    tx, rx = small_tx_rx(n_rx)
    place.add_set_of_points(rx)
    problem = RayTracingProblem(tx.reshape(1, 3), place)
    problem.solve(max_order=2)
    polys = oracle_polys(place)
    ids = {}
    for p, q in zip(polys, place.get_polygons_list()):
        p.building = ids.setdefault(q.building_id, len(ids))
    oracle = ro.RayTracingOracle(tx, polys, rx, visibility_matrix=problem.visibility_matrix, solver="exact")
    oracle.solve(max_order=2)
    assert problem.path_set() == oracle.path_set()
    return problem, oracle, polys, tx, rx, eo


def test_my_field_computation_matches_oracle(tmp_path, monkeypatch):
    from raytracing_amd import electromagnetism as em
    from raytracing_amd import electromagnetism_utils as emu

    monkeypatch.setenv("RAYTRACING_AMD_RESULTS", str(tmp_path))
    problem, oracle, polys, tx, rx, eo = _solved_small(8)
    df = em.my_field_computation(problem, "small_fields")
    rows = eo.FieldOracle(polys).field_computation(oracle, rx)
    assert len(df) == len(rows) > 50
    assert list(df["path_type"]) == [r["path_type"] for r in rows]
    assert list(df["rx_id"]) == [r["rx_id"] for r in rows]
    for (_, got), want in zip(df.iterrows(), rows):
        assert abs(db(got["path_power"]) - db(want["path_power"])) <= DB_TOL
        np.testing.assert_allclose(got["field_strength"], want["field_strength"], rtol=1e-6,
                                   atol=1e-9 * np.abs(want["field_strength"]).max())
        for key in ("tx_angles", "rx_angles"):
            assert abs(got[key][0] - want[key][0]) <= 1e-7
            if want[key][0] > 1e-6:  # looking straight down the antenna axis the azimuth is rounding noise
                assert abs(got[key][1] - want[key][1]) <= 1e-5
        assert abs(got["time_to_receiver"] - want["time_to_receiver"]) < 1e-15
    got_db = emu.get_power_db_each_receiver(df)
    want_db = eo.power_db_each_receiver(rows, len(rx))
    assert np.abs(got_db - want_db).max() <= DB_TOL
    # the CSV round trip keeps the table
    back = emu.load_df(os.path.join(str(tmp_path), "small_fields.csv"))
    assert list(back["path_type"]) == list(df["path_type"])
    np.testing.assert_allclose(np.array(list(back["path_power"]), dtype=float), np.array(list(df["path_power"]), dtype=float))


def test_frequency_sweep_matches_oracle():
    from raytracing_amd import electromagnetism as em

    problem, oracle, polys, tx, rx, eo = _solved_small(4)
    batch, _ = em.paths_of_problem(problem)
    freqs = [3.5e9, 12.5e9, 28e9, 60e9]
    res = batch.run(problem._device_scene(), tx, rx, freqs)
    assert not res["status"].any()
    for f, freq in enumerate(freqs):
        rows = eo.FieldOracle(polys, frequency=freq).field_computation(oracle, rx)
        want = np.array([r["path_power"] for r in rows])
        assert np.abs(db(res["power"][:, f]) - db(want)).max() <= DB_TOL
        want_db = eo.power_db_each_receiver(rows, len(rx))
        assert np.abs(res["power_db"][:, f] - want_db).max() <= DB_TOL


def test_a_sweep_gives_what_the_frequencies_give_one_by_one():
    """A call with several frequencies computes a path's geometry once and the frequency-dependent half per frequency
    (FIELDS_GEOMETRY / FIELDS_FREQUENCY); a call with one frequency does both in one thread (FIELDS_WHOLE).  The halves
    are the same arithmetic either way: fields equal to 1e-13 relative, the rest to the last bit or nearly."""
    from raytracing_amd import electromagnetism as em

    problem, oracle, polys, tx, rx, eo = _solved_small(4)
    batch, _ = em.paths_of_problem(problem)
    freqs = [3.5e9, 12.5e9, 28e9, 60e9]
    sweep = batch.run(problem._device_scene(), tx, rx, freqs)
    kinds = set(np.asarray(batch.kind).tolist()) if hasattr(batch, "kind") else None
    assert kinds is None or (0 in kinds and any(k >= 256 for k in kinds) and any(0 < k < 256 for k in kinds))
    for f, freq in enumerate(freqs):
        one = batch.run(problem._device_scene(), tx, rx, [freq])
        np.testing.assert_array_equal(one["status"], sweep["status"])
        np.testing.assert_array_equal(one["angles"], sweep["angles"])
        np.testing.assert_array_equal(one["time"], sweep["time"])
        scale = np.abs(one["field"][:, 0]).max(axis=-1, keepdims=True) + 1e-300
        assert (np.abs(one["field"][:, 0] - sweep["field"][:, f]) <= 1e-13 * scale).all()
        np.testing.assert_allclose(one["power"][:, 0], sweep["power"][:, f], rtol=1e-12, atol=0)
        np.testing.assert_allclose(one["power_db"][:, 0], sweep["power_db"][:, f], rtol=0, atol=1e-10)


def test_receiver_without_path_is_impossible(tmp_path, monkeypatch):
    from raytracing_amd import electromagnetism as em
    from raytracing_amd import geometry as geom
    from raytracing_amd.ray_tracing import RayTracingProblem

    monkeypatch.setenv("RAYTRACING_AMD_RESULTS", str(tmp_path))
    place = geom.load_place("data/small.geojson")
    below = np.array([0.0, 0.0, -5.0])  # a receiver under the ground: everything towards it is obstructed
    place.add_set_of_points(np.array([below, [5.0, 15.0, 1.5]]))
    problem = RayTracingProblem(np.array([[3.0, 38.0, 18.0]]), place)
    problem.solve(max_order=2)
    df = em.my_field_computation(problem, "impossible")
    first = df[df["rx_id"] == 0]
    assert list(first["path_type"]) == ["Impossible"] and float(first["path_power"].iloc[0]) == 0
    assert (df[df["rx_id"] == 1]["path_type"] != "Impossible").all()
    with pytest.raises(TypeError):
        unsolved = RayTracingProblem(np.array([[3.0, 38.0, 18.0]]), place)
        em.my_field_computation(unsolved)


def test_levant_fields_match_oracle(tmp_path, monkeypatch):
    """A scene with many-vertex roofs and 24 buildings: per-path power and per-receiver dB of an order-2 solve against
    the oracle's field computation on the oracle's own (exact-root) paths."""
    from oracle import em_oracle as eo
    from oracle import ray_tracing_oracle as ro
    from raytracing_amd import electromagnetism as em
    from raytracing_amd import electromagnetism_utils as emu
    from raytracing_amd import geometry as geom
    from raytracing_amd.ray_tracing import RayTracingProblem

    monkeypatch.setenv("RAYTRACING_AMD_RESULTS", str(tmp_path))
    place = geom.load_place("data/place_levant.geojson")
    tx = np.array([8.0, 0.0, 18.7])
    from raytracing_amd import scenarios

    rx = scenarios.lattice_receivers(place, 5, 1.5)[[3, 17]]  # two receivers in the open
    place.add_set_of_points(rx)
    problem = RayTracingProblem(tx.reshape(1, 3), place)
    problem.solve(max_order=2)
    polys = oracle_polys(place)
    ids = {}
    for p, q in zip(polys, place.get_polygons_list()):
        p.building = ids.setdefault(q.building_id, len(ids))
    oracle = ro.RayTracingOracle(tx, polys, rx, visibility_matrix=problem.visibility_matrix, solver="exact")
    oracle.solve(max_order=2)
    assert problem.path_set() == oracle.path_set()
    try:
        rows = eo.FieldOracle(polys).field_computation(oracle, rx)
    except AssertionError as err:  # the reference's own physics assertions: both sides must stop on the same one
        with pytest.raises(AssertionError) as caught:
            em.my_field_computation(problem, "levant_fields")
        assert str(err).split("(")[0].strip() in str(caught.value)
        return
    df = em.my_field_computation(problem, "levant_fields")
    print("levant: rows compared", len(rows))
    assert len(df) == len(rows) >= 10
    assert list(df["path_type"]) == [r["path_type"] for r in rows]
    for (_, got), want in zip(df.iterrows(), rows):
        assert abs(db(got["path_power"]) - db(want["path_power"])) <= DB_TOL
    got_db = emu.get_power_db_each_receiver(df)
    want_db = eo.power_db_each_receiver(rows, len(rx))
    assert np.abs(got_db - want_db).max() <= DB_TOL


def test_tree_crowns_attenuate_every_segment_they_cross(tmp_path, monkeypatch):
    """Trunks stand in the scene while it is solved; my_field_computation puts the crowns on them afterwards and every
    segment of every path loses what the crowns it crosses take (electromagnetism.py:179-199 at :338, 348, 506, 669,
    725).  Oracle: the same on the CPU, scanning every polygon for every segment as the reference does."""
    import random

    from oracle import em_oracle as eo
    from oracle import ray_tracing_oracle as ro
    from raytracing_amd import electromagnetism as em
    from raytracing_amd import geometry as geom
    from raytracing_amd.ray_tracing import RayTracingProblem

    monkeypatch.setenv("RAYTRACING_AMD_RESULTS", str(tmp_path))
    random.seed(11)
    place = geom.load_place("data/small.geojson")
    for spot, crown in (((6.0, 5.0, 0.0), (5.0, 6.0)), ((2.0, -12.0, 0.0), (6.0, 5.0)), ((10.0, 14.0, 0.0), (4.0, 7.0))):
        place.add_polyhedron(geom.Building.create_tree_trunk(np.array(spot), 0.6, 7.0, rotate=True, crown_size=crown[0],
                                                             crown_height=crown[1]))
    tx, rx = small_tx_rx(12)
    place.add_set_of_points(rx)
    problem = RayTracingProblem(tx.reshape(1, 3), place)
    problem.solve(max_order=2)

    def with_buildings(polys):
        ids = {}
        for p, q in zip(polys, place.get_polygons_list()):
            p.building = ids.setdefault(q.building_id, len(ids))
        return polys

    solved_with = with_buildings(oracle_polys(place))
    oracle = ro.RayTracingOracle(tx, solved_with, rx, visibility_matrix=problem.visibility_matrix, solver="exact")
    oracle.solve(max_order=2)
    assert problem.path_set() == oracle.path_set()

    df = em.my_field_computation(problem, "small_trees")
    assert len(place.get_polygons_list()) == len(solved_with) + 3 * 6  # three crowns: top, bottom, four sides
    with_crowns = with_buildings(oracle_polys(place))
    rows = eo.FieldOracle(with_crowns).field_computation(oracle, rx)
    bare = eo.FieldOracle(with_crowns)
    bare.account_for_trees = lambda E, points: E
    rows_bare = bare.field_computation(oracle, rx)
    assert len(df) == len(rows) == len(rows_bare) > 50
    assert list(df["path_type"]) == [r["path_type"] for r in rows]
    attenuated = {}
    for (_, got), want, plain in zip(df.iterrows(), rows, rows_bare):
        if want["path_type"] == "Impossible":
            continue
        assert abs(db(got["path_power"]) - db(want["path_power"])) <= DB_TOL
        np.testing.assert_allclose(got["field_strength"], want["field_strength"], rtol=1e-6,
                                   atol=1e-9 * np.abs(want["field_strength"]).max())
        if db(plain["path_power"]) - db(want["path_power"]) > 1.0:
            attenuated[want["path_type"]] = attenuated.get(want["path_type"], 0) + 1
    # the crowns matter in this scene: lines of sight, reflected and diffracted paths lose more than 1 dB to them
    assert attenuated.get("LOS", 0) >= 1 and sum(v for k, v in attenuated.items() if k != "LOS") >= 3, attenuated


def test_device_pointer_route_equals_host_array_route(places):
    """Solve, path description and fields through the device-pointer entry points (nothing but the result leaves the
    GPU) against the host-array entry points on the same scene: same paths, same fields, same dB per receiver."""
    import torch

    from raytracing_amd import _lib
    from raytracing_amd import geometry as geom

    place = places("small")
    tx, rx = small_tx_rx(16)
    polygons = place.get_polygons_list()
    scene = _lib.Scene(geom.scene_arrays_of(polygons))
    pairs, edges = scene.sharp_edges()
    vis = scene.visibility_matrix(False)
    cube = geom.Cube.by_point_and_side_length(tx, 0.2)
    ev = np.logical_or.reduce(scene.faces_visibility([(s.points, s.get_parametric(), -1) for s in cube.polygons], False))
    cand = scene.enumerate(vis, ev, pairs, edges, 2, with_diffraction=True)
    freqs = [3.6e9, 28e9]
    # host arrays
    res = scene.solve_paths(cand, tx, rx)
    n = len(res["rx"])
    keep = res["group"] >= 0
    kind = np.zeros(n, dtype=np.int32)
    chain = np.zeros((n, _lib.RT_MAX_ORDER), dtype=np.int32)
    epolys = np.zeros((n, 2), dtype=np.int32)
    epts = np.zeros((n, 6))
    kind[keep], chain[keep], epolys[keep], epts[keep] = cand.describe(res["group"][keep], res["cand"][keep])
    want = scene.fields(tx, rx, res["rx"], kind, chain, epolys, epts, res["points"], freqs)
    # device pointers
    dev = torch.device("cuda", 0)
    d_rx = torch.tensor(rx, dtype=torch.float64, device=dev)
    cap = 4096
    o_rx = torch.empty(cap, dtype=torch.int32, device=dev)
    o_group, o_cand = torch.empty_like(o_rx), torch.empty_like(o_rx)
    o_pts = torch.empty((cap, _lib.RT_MAX_ORDER, 3), dtype=torch.float64, device=dev)
    found, _ = scene.solve_paths_device(cand, tx, d_rx.data_ptr(), len(rx), o_rx.data_ptr(), o_group.data_ptr(),
                                        o_cand.data_ptr(), o_pts.data_ptr(), cap)
    assert found == n > 100
    d_kind = torch.empty(n, dtype=torch.int32, device=dev)
    d_chain = torch.empty((n, _lib.RT_MAX_ORDER), dtype=torch.int32, device=dev)
    d_epolys = torch.empty((n, 2), dtype=torch.int32, device=dev)
    d_epts = torch.empty((n, 6), dtype=torch.float64, device=dev)
    cand.describe_device(o_group.data_ptr(), o_cand.data_ptr(), n, d_kind.data_ptr(), d_chain.data_ptr(),
                         d_epolys.data_ptr(), d_epts.data_ptr())
    nf = len(freqs)
    d_field = torch.empty((n, nf, 6), dtype=torch.float64, device=dev)
    d_power = torch.empty((n, nf), dtype=torch.float64, device=dev)
    d_angles = torch.empty((n, 4), dtype=torch.float64, device=dev)
    d_time = torch.empty(n, dtype=torch.float64, device=dev)
    d_status = torch.empty(n, dtype=torch.int32, device=dev)
    d_db = torch.empty((len(rx), nf), dtype=torch.float64, device=dev)
    scene.fields_device(tx, d_rx.data_ptr(), len(rx), o_rx.data_ptr(), d_kind.data_ptr(), d_chain.data_ptr(),
                        d_epolys.data_ptr(), d_epts.data_ptr(), o_pts.data_ptr(), n, freqs, d_field.data_ptr(),
                        d_power.data_ptr(), d_angles.data_ptr(), d_time.data_ptr(), d_status.data_ptr(), d_db.data_ptr())
    # the two solves may list the paths in another order: compare by (receiver, group, candidate)
    def order(r, g, c):
        return np.lexsort((c, g, r))
    a = order(res["rx"], res["group"], res["cand"])
    b = order(o_rx[:n].cpu().numpy(), o_group[:n].cpu().numpy(), o_cand[:n].cpu().numpy())
    np.testing.assert_array_equal(res["rx"][a], o_rx[:n].cpu().numpy()[b])
    np.testing.assert_array_equal(res["cand"][a], o_cand[:n].cpu().numpy()[b])
    np.testing.assert_array_equal(kind[a], d_kind.cpu().numpy()[b])
    paths = res["group"][a] >= 0  # the host arrays of the lines of sight above were left at zero, the device form writes -1
    np.testing.assert_array_equal(chain[a][paths], d_chain.cpu().numpy()[b][paths])
    np.testing.assert_array_equal(want["status"][a], d_status.cpu().numpy()[b])
    np.testing.assert_array_equal(want["power"][a], d_power.cpu().numpy()[b])
    np.testing.assert_array_equal(want["angles"][a], d_angles.cpu().numpy()[b])
    # the per-receiver sums are accumulated with atomics: the order of the additions differs
    np.testing.assert_allclose(d_db.cpu().numpy(), want["power_db"], rtol=0, atol=1e-9)
    cand.close()
    scene.close()


@pytest.mark.gpu
def test_pipeline_waits_on_the_device_for_the_callers_stream(places):
    """Receivers that the caller's own kernels are still writing when the pipeline is called: the scene's stream is
    ordered after the caller's stream on the device (rt_scene_wait_for_stream), so the result is the one of receivers
    that were complete beforehand."""
    import torch

    from raytracing_amd import _lib
    from raytracing_amd import geometry as geom
    from raytracing_amd.device_pipeline import DevicePipeline

    place = places("small")
    tx, rx = small_tx_rx(16)
    scene = _lib.Scene(geom.scene_arrays_of(place.get_polygons_list()))
    pairs, edges = scene.sharp_edges()
    vis = scene.visibility_matrix(False)
    cube = geom.Cube.by_point_and_side_length(tx, 0.2)
    ev = np.logical_or.reduce(scene.faces_visibility([(s.points, s.get_parametric(), -1) for s in cube.polygons], False))
    cand = scene.enumerate(vis, ev, pairs, edges, 2, with_diffraction=True)
    dev = torch.device("cuda", 0)
    pipe = DevicePipeline(scene, cand, tx)
    ready = torch.tensor(rx, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    want = pipe.run(ready, [3.6e9]).to_host()
    assert len(want["rx"]) > 100
    # the same receivers, produced on a side stream of torch behind a second of queued work
    stream = torch.cuda.Stream(device=dev)
    big = torch.randn((8192, 8192), device=dev)
    late = torch.full_like(ready, 1.0e6)  # far outside the scene until the producer has run
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        for _ in range(20):
            big = torch.tanh(big @ big)
        late.copy_(ready)
        got = pipe.run(late, [3.6e9]).to_host()
    torch.cuda.synchronize()
    n_pts = (want["kind"] & 255) + (want["kind"] >> 8)
    used = np.arange(want["points"].shape[1])[None, :] < n_pts[:, None]  # the slots past a path's points are not written
    for name in ("rx", "group", "cand", "kind", "power", "power_db"):
        np.testing.assert_array_equal(got[name], want[name])
    np.testing.assert_array_equal(got["points"][used], want["points"][used])


@pytest.mark.gpu
def test_receiver_records_hold_what_the_path_arrays_say(places):
    """The fixed-size record per receiver (rt_receiver_records) against a NumPy count over the same block's path
    arrays: dB per frequency, paths per type, paths in all, flagged paths, worst status."""
    import torch

    from raytracing_amd import _lib
    from raytracing_amd import device_pipeline as dp
    from raytracing_amd import geometry as geom

    place = places("small")
    tx, rx = small_tx_rx(16)
    rx = np.concatenate([rx, [[500.0, 500.0, 1.5]]])  # a receiver no path reaches but the line of sight
    scene = _lib.Scene(geom.scene_arrays_of(place.get_polygons_list()))
    pairs, edges = scene.sharp_edges()
    vis = scene.visibility_matrix(False)
    cube = geom.Cube.by_point_and_side_length(tx, 0.2)
    ev = np.logical_or.reduce(scene.faces_visibility([(s.points, s.get_parametric(), -1) for s in cube.polygons], False))
    cand = scene.enumerate(vis, ev, pairs, edges, 3, with_diffraction=True)
    pipe = dp.DevicePipeline(scene, cand, tx)
    d_rx = torch.tensor(rx, dtype=torch.float64, device="cuda:0")
    freqs = [3.6e9, 28e9, 60e9]
    block = pipe.run(d_rx, freqs)
    block.status[::7] = 3  # as if the field step had flagged some paths
    block.status[5] = 9
    rec = dp.unpack_records(dp.receiver_records(block).cpu().numpy(), len(freqs))
    host = block.to_host()
    status = block.status.cpu().numpy()
    prx, kind = block.rx.cpu().numpy(), block.kind.cpu().numpy()
    assert dp.record_bytes(3) == 72 and dp.record_bytes(2) == 64 and dp.record_bytes(1) == 56
    np.testing.assert_array_equal(rec["power_db"], host["power_db"])
    types = (kind & 255) + (_lib.RT_MAX_ORDER + 1) * (kind >> 8)
    for r in range(len(rx)):
        mine = prx == r
        np.testing.assert_array_equal(rec["counts"][r], np.bincount(types[mine], minlength=dp.N_TYPES))
        assert rec["n_paths"][r] == mine.sum()
        assert rec["n_flagged"][r] == (status[mine] != 0).sum()
        assert rec["worst_status"][r] == (status[mine].max() if mine.any() and status[mine].max() > 0 else 0)
    assert rec["n_paths"].sum() == block.n_paths > 100 and rec["counts"][:, 0].sum() >= 10
    assert set(np.unique(types).tolist()) >= {0, 1, 2, 5, 6}  # LOS, R, RR, D, RD all occur


def test_steps_queued_behind_each_other_give_what_the_waited_for_steps_give(places):
    """rt_scene_set_deferred_sync: description, field step and records queued on the scene's stream with one wait at
    the end (DevicePipeline.run(records=True)) against the same three steps each waited for; twice, so that the
    temporary blocks handed out again behind queued work are covered."""
    import torch

    from raytracing_amd import _lib
    from raytracing_amd import device_pipeline as dp
    from raytracing_amd import geometry as geom

    place = places("place_levant")
    polygons = place.get_polygons_list()
    scene = _lib.Scene(geom.scene_arrays_of(polygons))
    lo, hi = place.get_domain()
    tx = np.array([(lo[0] + hi[0]) / 2, (lo[1] + hi[1]) / 2, 30.0])
    rng = np.random.default_rng(3)
    rx = np.column_stack((rng.uniform(lo[0], hi[0], 300), rng.uniform(lo[1], hi[1], 300), np.full(300, 1.5)))
    pairs, edges = scene.sharp_edges()
    vis = scene.visibility_matrix(False)
    cube = geom.Cube.by_point_and_side_length(tx, 0.2)
    ev = np.logical_or.reduce(scene.faces_visibility([(s.points, s.get_parametric(), -1) for s in cube.polygons], False))
    cand = scene.enumerate(vis, ev, pairs, edges, 2, with_diffraction=True)
    pipe = dp.DevicePipeline(scene, cand, tx)
    freqs = [3.6e9, 28e9]
    for n in (300, 117):
        d_rx = torch.tensor(rx[:n], dtype=torch.float64, device="cuda:0")
        waited = pipe.run(d_rx, freqs)
        want = waited.to_host()  # (sorted by receiver, group, candidate: a solve leaves its paths in no fixed order)
        want_rec = dp.receiver_records(waited).clone()
        queued = pipe.run(d_rx, freqs, records=True)
        assert queued.records is not None and queued.n_paths == waited.n_paths > 200
        got = queued.to_host()
        n_pts = (want["kind"] & 255) + (want["kind"] >> 8)
        used = np.arange(want["points"].shape[1])[None, :] < n_pts[:, None]  # the slots past a path's points are not written
        for k, v in want.items():
            if k == "points":
                np.testing.assert_array_equal(got[k][used], v[used], err_msg=k)
            else:
                np.testing.assert_array_equal(got[k], v, err_msg=k)
        assert torch.equal(dp.receiver_records(queued), want_rec)
        assert queued.fields_timing is not None
