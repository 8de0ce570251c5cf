"""Save / load round trips as the reference tests them (tests/test_geometry_advanced.py:23-82 on polygons, polyhedra,
the ground surface and the place of create_small_place(npoints=3), before and after a solve; tests/test_ray_tracing.py:
15-31 on the solved problem, compared with the reference's own notion of equality, ray_tracing.py:146-200)."""
import os

import numpy as np
import pytest

from conftest import small_tx_rx


def small_place(npoints=3):
    from raytracing_amd import geometry as geom

    place = geom.load_place("data/small.geojson")
    tx, rx = small_tx_rx(npoints)
    place.add_set_of_points(rx)
    return place, tx.reshape(1, 3)


def check_all_save_load(place, tmp_path):
    from raytracing_amd import geometry as geom

    name = str(tmp_path / "saved.json")
    for polyhedron in place.polyhedra:
        polyhedron.to_json(name)
        assert geom.OrientedPolyhedron.from_json(filename=name) == polyhedron
        for polygon in polyhedron.polygons:
            polygon.to_json(name)
            loaded = geom.OrientedPolygon.from_json(filename=name)
            assert loaded == polygon
            np.testing.assert_array_equal(loaded.get_parametric(), polygon.get_parametric())
    place.surface.to_json(name)
    assert geom.OrientedSurface.from_json(filename=name) == place.surface
    place.to_json(name)
    loaded = geom.OrientedPlace.from_json(filename=name)
    if place.sharp_edges is not None:
        # a solve leaves the place's sharp edges cached on it; the file does not hold them (geometry.py:1219-1224 of
        # the reference) and the comparison looks at them (:1145), so the loaded place equals the solved one once it
        # has found its own
        assert loaded != place
        loaded.get_sharp_edges()
    assert loaded == place
    assert [p.part for p in loaded.get_polygons_iter()] == [p.part for p in place.get_polygons_iter()]
    return loaded


def test_all_save_load_before_solve(tmp_path):
    from raytracing_amd import geometry as geom

    place, _ = small_place()
    loaded = check_all_save_load(place, tmp_path)
    # and the comparisons do tell things apart
    other, _ = small_place()  # same file, fresh ids
    assert other != place
    moved = geom.OrientedPlace.from_json(data=place.to_json())
    moved.set_of_points = moved.set_of_points + 1.0
    assert moved != place and loaded == place
    poly = place.polyhedra[0].polygons[0]
    twin = geom.OrientedPolygon.from_json(data=poly.to_json())
    twin.points = twin.points + 1e-9
    assert twin != poly
    with pytest.raises(ValueError):
        geom.OrientedPolygon.from_json(data=place.surface.to_json())
    with pytest.raises(ValueError):
        geom.OrientedPlace.from_json(data=poly.to_json())


@pytest.mark.gpu
def test_all_save_load_after_solve_and_problem_round_trip(tmp_path):
    from raytracing_amd.ray_tracing import RayTracingProblem

    place, tx = small_place()
    problem = RayTracingProblem(tx, place)
    problem.solve(max_order=2, receivers_indexs=np.arange(len(place.set_of_points)))
    check_all_save_load(place, tmp_path)
    name = os.path.join(str(tmp_path), "saved_raytracing_problem.json")
    problem.to_json(filename=name)
    loaded = RayTracingProblem.from_json(filename=name)
    assert problem == loaded, "problems are not equal"
    assert loaded.path_set() == problem.path_set()  # more than the reference's == looks at
    np.testing.assert_array_equal(loaded.visibility_matrix, problem.visibility_matrix)
    again = RayTracingProblem(tx, loaded.place)  # the loaded place is the place that was saved: same solve
    again.solve(max_order=2)
    assert again.path_set() == problem.path_set()
    for r in problem.reflections:
        for k in problem.reflections[r]:
            for (a, ca), (b, cb) in zip(problem.reflections[r][k], loaded.reflections[r][k]):
                np.testing.assert_array_equal(a, b)
                assert list(ca) == list(cb)
    other = RayTracingProblem(tx + np.array([[0.0, 0.0, 1.0]]), place)
    assert other != problem


@pytest.mark.gpu
def test_json_written_from_the_arrays_is_the_json_of_the_dictionaries(tmp_path):
    """to_json writes the path containers straight from the solve's arrays while nobody has looked at the
    dictionaries; once they have been filled it walks them.  Same file, character for character."""
    from raytracing_amd import geometry as geom
    from raytracing_amd.ray_tracing import RayTracingProblem

    place = geom.load_place("data/small.geojson")
    tx, rx = small_tx_rx(12)
    place.add_set_of_points(rx)
    problem = RayTracingProblem(tx.reshape(1, 3), place)
    problem.solve(max_order=2)
    assert len(problem._pending) == 1
    problem.to_json(str(tmp_path / "from_arrays.json"))
    assert len(problem._pending) == 1  # writing the file has not filled the dictionaries
    n_refl = sum(len(v) for by_order in problem.reflections.values() for v in by_order.values())
    n_diff = sum(len(v) for by_order in problem.diffractions.values() for v in by_order.values())
    assert not problem._pending and n_refl > 20 and n_diff > 20
    problem.to_json(str(tmp_path / "from_dictionaries.json"))
    assert open(tmp_path / "from_arrays.json").read() == open(tmp_path / "from_dictionaries.json").read()
    loaded = RayTracingProblem.from_json(str(tmp_path / "from_arrays.json"))
    assert loaded == problem and loaded.path_set() == problem.path_set()


@pytest.mark.gpu
def test_json_of_receivers_solved_out_of_order(tmp_path):
    """solve(receivers_indexs=[2, 0]) hands the paths over receiver 2 first: the array route of to_json needs them by
    ascending receiver, so the file then comes from the dictionaries, and says what a solve in order says."""
    from raytracing_amd import geometry as geom
    from raytracing_amd.ray_tracing import RayTracingProblem

    def solved(indices):
        place = geom.load_place("data/small.geojson")
        tx, rx = small_tx_rx(4)
        place.add_set_of_points(rx)
        problem = RayTracingProblem(tx.reshape(1, 3), place)
        problem.solve(max_order=2, receivers_indexs=indices)
        return problem

    backwards = solved([2, 0])
    backwards.to_json(str(tmp_path / "backwards.json"))
    loaded = RayTracingProblem.from_json(str(tmp_path / "backwards.json"))
    assert loaded.path_set() == backwards.path_set()
    forwards = solved([0, 2])
    assert forwards.path_set() == backwards.path_set() and len(forwards.path_set()) > 10
    for r in (0, 2):
        for k in forwards.reflections[r]:
            for (a, ca), (b, cb) in zip(forwards.reflections[r][k], loaded.reflections[r][k]):
                np.testing.assert_array_equal(a, b)
                assert list(ca) == list(cb)
    twice = solved([1, 1])  # a receiver listed twice has its paths twice, as in the reference (results are appended)
    twice.to_json(str(tmp_path / "twice.json"))
    again = RayTracingProblem.from_json(str(tmp_path / "twice.json"))
    once = solved([1])
    n_once = sum(len(v) for by in once.reflections.values() for v in by.values())
    n_twice = sum(len(v) for by in again.reflections.values() for v in by.values())
    assert n_twice == 2 * n_once and n_once > 0
