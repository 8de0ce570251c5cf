"""The reference's analytic two-ray model (theoretical_validation.py:31-170) on its own scene (place_utils.py:106-123):
the known answer its authors hold the simulator against.  The oracle's pipeline reproduces it to rounding, and the GPU
pipeline (scene, visibility, enumeration, solve, obstruction, fields, per-receiver dB) must give it within the 0.01 dB
bar (it does to 1e-6 dB)."""
import numpy as np
import pytest

from conftest import oracle_polys
from oracle import em_oracle as eo
from oracle import ray_tracing_oracle as ro
from oracle import two_rays_oracle as two

NPOINTS = 20


def analytic_db(tx, rx):
    return np.array([two.two_rays_db(np.linalg.norm(r[:2] - tx[:2]), tx[2], r[2]) for r in rx])


def test_two_rays_geometry_known_answers():
    """Image construction by hand: TX 30 m and RX 10 m high, 30 m apart: the mirrored receiver is 40 m below the
    emitter's height, so the reflected ray is 50 m long and leaves the ground after 3/4 of the way."""
    dlos, dref, d1, d2, L1, L2, theta, theta_tx, theta_rx = two.two_rays_geometry(30.0, 30.0, 10.0)
    assert dref == 50.0 and abs(dlos - np.hypot(20.0, 30.0)) < 1e-12
    assert abs(d1 - 37.5) < 1e-12 and abs(d2 - 12.5) < 1e-12 and abs(L1 - 22.5) < 1e-12 and abs(L2 - 7.5) < 1e-12
    assert abs(theta - np.arctan2(4.0, 3.0)) < 1e-12
    with pytest.raises(AssertionError):
        two.two_rays_geometry(1.0, 30.0, 10.0)  # antennas too close (theoretical_validation.py:49-50)


def test_oracle_pipeline_reproduces_the_two_ray_model():
    from raytracing_amd import scenarios

    place, tx, rx = scenarios.two_rays_place(NPOINTS)
    place.add_set_of_points(rx)
    polys = oracle_polys(place)
    for p in polys:
        p.building = 0
    for solver in ("exact", "hybr"):
        oracle = ro.RayTracingOracle(tx, polys, rx, solver=solver)
        oracle.solve(max_order=2)
        assert sorted(k[1] for k in oracle.path_set()) == ["LOS"] * NPOINTS + ["R"] * NPOINTS
        rows = eo.FieldOracle(polys).field_computation(oracle, rx)
        got = eo.power_db_each_receiver(rows, len(rx))
        # hybr stops within ~1e-8 m of the reflection point; the power is taken from Re(E), whose phase turns by
        # 2 pi every 24 mm, so near a zero of the real part that shows as 1e-5 dB
        np.testing.assert_allclose(got, analytic_db(tx, rx), rtol=0, atol=1e-6 if solver == "exact" else 1e-3)
        for i, r in enumerate(rx):
            p_los, p_ref = two.two_rays_powers(np.linalg.norm(r[:2] - tx[:2]), tx[2], r[2])
            mine = {row["path_type"]: row["path_power"] for row in rows if row["rx_id"] == i}
            assert abs(mine["LOS"] / p_los - 1) < 1e-9 and abs(mine["R"] / p_ref - 1) < (1e-6 if solver == "exact" else 1e-3)


@pytest.mark.gpu
def test_gpu_pipeline_reproduces_the_two_ray_model(tmp_path, monkeypatch):
    from raytracing_amd import electromagnetism as em
    from raytracing_amd import electromagnetism_utils as emu
    from raytracing_amd import scenarios
    from raytracing_amd.ray_tracing import RayTracingProblem

    monkeypatch.setenv("RAYTRACING_AMD_RESULTS", str(tmp_path))
    place, tx, rx = scenarios.two_rays_place(NPOINTS)
    place.add_set_of_points(rx)
    problem = RayTracingProblem(tx.reshape(1, 3), place)
    problem.solve(max_order=2)
    df = em.my_field_computation(problem, "two_rays")
    assert sorted(df["path_type"]) == ["LOS"] * NPOINTS + ["R"] * NPOINTS
    got = emu.get_power_db_each_receiver(df)
    want = analytic_db(tx, rx)
    assert np.abs(got - want).max() <= 0.01  # the bar
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-6)
    # the reflection points are the image construction's: on the ground, L1 = L ztx / (ztx + zrx) from the emitter
    for i, r in enumerate(rx):
        (lines, chain), = problem.reflections[i][1]
        assert chain == [0]
        L = r[0] - tx[0]
        np.testing.assert_allclose(lines[1], [tx[0] + L * tx[2] / (tx[2] + r[2]), 12.0, 0.0], atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("freq", [3.5e9, 28e9, 60e9])
def test_gpu_two_ray_model_over_frequency(freq):
    from raytracing_amd import electromagnetism as em
    from raytracing_amd import scenarios
    from raytracing_amd.ray_tracing import RayTracingProblem

    place, tx, rx = scenarios.two_rays_place(8)
    place.add_set_of_points(rx)
    problem = RayTracingProblem(tx.reshape(1, 3), place)
    problem.solve(max_order=1 + 1)
    batch, _ = em.paths_of_problem(problem)
    res = batch.run(problem._device_scene(), tx, rx, [freq])
    assert not res["status"].any()
    want = np.array([two.two_rays_db(np.linalg.norm(r[:2] - tx[:2]), tx[2], r[2], frequency=freq) for r in rx])
    np.testing.assert_allclose(res["power_db"][:, 0], want, rtol=0, atol=1e-6)


@pytest.mark.gpu
def test_gpu_dummy_place_against_the_oracle(tmp_path, monkeypatch):
    """The reference's create_dummy_place (place_utils.py:76-103: a road at 10 m with three blocks, one receiver 60 m
    down the road): same paths as the oracle with exact roots, received power within 0.01 dB."""
    from raytracing_amd import electromagnetism as em
    from raytracing_amd import electromagnetism_utils as emu
    from raytracing_amd import scenarios
    from raytracing_amd.ray_tracing import RayTracingProblem

    monkeypatch.setenv("RAYTRACING_AMD_RESULTS", str(tmp_path))
    place, tx, rx = scenarios.dummy_place()
    place.add_set_of_points(rx)
    problem = RayTracingProblem(tx.reshape(1, 3), place)
    problem.solve(max_order=2)
    polys = oracle_polys(place)
    ids = {}
    for p, q in zip(polys, place.get_polygons_list()):
        p.building = ids.setdefault(q.building_id, len(ids))
    assert len(polys) == 19
    oracle = ro.RayTracingOracle(tx, polys, rx, solver="exact")
    oracle.solve(max_order=2)
    np.testing.assert_array_equal(problem.visibility_matrix, oracle.visibility_matrix)
    assert problem.path_set() == oracle.path_set()
    kinds = sorted(k[1] + str(k[2]) for k in problem.path_set())
    assert kinds.count("LOS0") == 1 and kinds.count("R1") == 2  # the road and the wall of the middle block
    df = em.my_field_computation(problem, "dummy")
    rows = eo.FieldOracle(polys).field_computation(oracle, rx)
    got = emu.get_power_db_each_receiver(df)
    want = eo.power_db_each_receiver(rows, len(rx))
    assert np.abs(got - want).max() <= 0.01
