"""The HIP path against the reference's own path set and received power on fixed receiver samples of the five BASELINE
workloads.  tests/golden/hybr_vs_exact.json (made on the CPU by tests/golden/make_hybr_vs_exact.py) lists, per
workload, every path the oracle keeps with SciPy's hybr exactly as the reference calls it, the paths that have a root
hybr did not report (only_exact) and the reverse (only_hybr), and the per-receiver dB of both path sets.

Default mode (exact roots): GPU paths minus hybr's paths must be EXACTLY the only_exact list, hybr's minus the GPU's
exactly only_hybr, and the per-receiver dB the exact-root oracle's within 0.01 dB.
Reference-compatible mode (every found path re-judged on the GPU with the restatement of hybr): the GPU path set must be
hybr's path set, and the per-receiver dB hybr's within 0.01 dB."""
import json
import os

import numpy as np
import pytest

import parity_configs as pc
from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DB_TOL = 0.01


def _fixture(name):
    with open(os.path.join(GOLDEN, "hybr_vs_exact.json")) as f:
        data = json.load(f)
    if name not in data:
        pytest.skip(f"{name} is not in the fixture yet")
    return data[name]


def _solve(name, reference_solver):
    from raytracing_amd import electromagnetism as em
    from raytracing_amd import electromagnetism_utils as emu
    from raytracing_amd.ray_tracing import RayTracingProblem

    fx = _fixture(name)
    w = pc.workload(name)
    assert [int(i) for i in w["rx_index"]] == fx["rx_index"]
    rx = w["rx_all"][w["rx_index"]]
    np.testing.assert_array_equal(rx, np.array(fx["rx"]))
    place = w["place"]
    place.add_set_of_points(rx)
    problem = RayTracingProblem(np.asarray(w["tx"]).reshape(1, 3), place)
    assert int(problem.visibility_matrix.sum()) == fx["visibility_matrix_popcount"]
    assert int(problem.emitter_visibility.sum()) == fx["emitter_visibility_popcount"]
    assert len(problem.sharp_edges) == fx["n_sharp_edges"]
    problem.reference_solver = reference_solver
    problem.solve(max_order=w["order"])
    assert problem.path_solves_attempted == sum(fx["path_solves"].values())
    local_to_global = {k: int(g) for k, g in enumerate(w["rx_index"])}
    got = {pc.path_key(local_to_global[r], kind, order, chain, edge) for (r, kind, order, chain, edge) in problem.path_set()}
    db = []
    for f in w["frequencies"]:
        df = em.my_field_computation(problem, frequency=f, save=False)
        db.append(emu.get_power_db_each_receiver(df))
    return fx, got, np.array(db).T  # [receiver, frequency]


def _compare_db(got_db, want_db, what):
    want = np.array([[np.nan if v is None else v for v in row] for row in want_db], dtype=float)
    assert got_db.shape == want.shape
    finite = np.isfinite(want)
    assert np.array_equal(np.isfinite(got_db), finite), f"{what}: receivers with and without power differ"
    worst = np.abs(got_db[finite] - want[finite]).max() if finite.any() else 0.0
    assert worst <= DB_TOL, f"{what}: per-receiver power differs by {worst} dB"
    return worst


@pytest.mark.parametrize("name", pc.NAMES)
def test_exact_root_mode_differs_from_hybr_by_exactly_the_listed_paths(name):
    fx, got, db = _solve(name, reference_solver=False)
    hybr = {pc.path_key(*p) for p in fx["paths_hybr"]}
    only_exact = {pc.path_key(*p) for p in fx["only_exact"]}
    only_hybr = {pc.path_key(*p) for p in fx["only_hybr"]}
    assert got - hybr == only_exact, f"extra: {sorted((got - hybr) - only_exact)[:5]}, missing: {sorted(only_exact - (got - hybr))[:5]}"
    assert hybr - got == only_hybr
    assert len(got) == fx["n_paths_exact"]
    if not fx["field_asserts_exact"]:
        _compare_db(db, fx["db_exact"], name + " against the exact-root oracle")


@pytest.mark.parametrize("name", pc.NAMES)
def test_reference_compatible_mode_keeps_hybrs_paths_and_power(name):
    fx, got, db = _solve(name, reference_solver=True)
    hybr = {pc.path_key(*p) for p in fx["paths_hybr"]}
    assert got == hybr, f"only GPU: {sorted(got - hybr)[:5]}, only hybr: {sorted(hybr - got)[:5]}"
    if not fx["field_asserts_hybr"]:
        _compare_db(db, fx["db_hybr"], name + " against the oracle with SciPy hybr")


def test_the_three_solver_functions_answer_as_the_references_solver_does():
    """geometry.reflexion_points_from_origin_destination_and_planes & co. with reference_solver=True reproduce, on the
    GPU and through the C ABI (rt_solve_like_reference), every vector recorded from the reference's own three solver
    functions (tests/golden/solvers.json: sol.x, sol.success, sol.nfev) bit for bit."""
    from raytracing_amd import geometry as geom

    with open(os.path.join(GOLDEN, "solvers.json")) as f:
        cases = json.load(f)
    planes = np.load(os.path.join(GOLDEN, "helpers.npz"))["cube_planes"]
    n = 0
    for kind in ("reflection", "diffraction", "mixed"):
        for case in cases[kind]:
            pl = [planes[f] for f in case.get("faces", [])]
            a, b = np.array(case["a"]), np.array(case["b"])
            if kind == "reflection":
                x, sol = geom.reflexion_points_from_origin_destination_and_planes(a, b, pl, reference_solver=True)
            elif kind == "diffraction":
                x, sol = geom.diffraction_point_from_origin_destination_and_edge(a, b, np.array(case["edge"]), reference_solver=True)
            else:
                x, sol = geom.reflexion_points_and_diffraction_point_from_origin_destination_planes_and_edge(
                    a, b, pl, np.array(case["edge"]), reference_solver=True)
            assert sol.success == case["success"]
            np.testing.assert_array_equal(np.asarray(x).reshape(-1, 3), np.array(case["x"]).reshape(-1, 3))  # bit for bit
            if "nfev" in case:
                assert sol.nfev == case["nfev"]
            n += 1
    assert n == 154
    # and without the flag the same functions give the exact root, which the solver's answer is close to when it succeeds
    case = cases["reflection"][0]
    pl = [planes[f] for f in case["faces"]]
    x_exact, sol_exact = geom.reflexion_points_from_origin_destination_and_planes(np.array(case["a"]), np.array(case["b"]), pl)
    if case["success"] and sol_exact.success:
        assert np.abs(np.asarray(x_exact) - np.array(case["x"]).reshape(-1, 3)).max() < 1e-5
