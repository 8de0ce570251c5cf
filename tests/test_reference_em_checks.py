"""The checks of the reference's own tests/test_electromagnetism.py, run against the host mirror and the oracle alike:
antenna frame invariants (:58-164; the reference calls them from its __main__ with an antenna aligned to a point) and
the known answers of Reflection.split_reflections_path (:170-185)."""
import numpy as np
import pytest

from oracle import em_oracle as eo


def antennas():
    from raytracing_amd import electromagnetism as em

    rng = np.random.default_rng(11)
    cases = [(np.array([5.0, 12.0, 30.0]), np.array([55.0, 12.0, 10.0])),
             (np.array([0.0, 0.0, 0.0]), np.array([6.0, 6.0, 6.0])),
             (np.array([3.0, 38.0, 18.0]), np.array([5.0, 15.0, 1.5]))]
    cases += [(rng.uniform(-50, 50, 3), rng.uniform(-50, 50, 3)) for _ in range(5)]
    out = []
    for cls in (em.Antenna, eo.Antenna):
        for pos, point in cases:
            a = cls(pos)
            a.align_antenna_to_point(point)
            out.append((a, point))
    return out


@pytest.mark.parametrize("antenna,point", antennas())
def test_pp_transform(antenna, point):
    np.testing.assert_allclose(antenna.pp_transform(antenna.position, "W2A"), np.zeros(3), atol=1e-12)
    np.testing.assert_allclose(antenna.pp_transform(np.zeros(3), "A2W"), antenna.position, atol=1e-12)
    dist = np.linalg.norm(antenna.position - point)
    np.testing.assert_allclose(antenna.pp_transform(point, "W2A"), [0, 0, dist], atol=1e-9)
    np.testing.assert_allclose(antenna.pp_transform(np.array([0, 0, dist]), "A2W"), point, atol=1e-9)


@pytest.mark.parametrize("antenna,point", antennas())
def test_vv_transform_and_orthonormality(antenna, point):
    basis = [np.asarray(b) for b in antenna.basis]
    for k in range(3):
        in_ant = antenna.vv_transform(basis[k], "W2A")
        np.testing.assert_allclose(in_ant, np.eye(3)[k], atol=1e-12)
        np.testing.assert_allclose(antenna.vv_transform(in_ant, "A2W"), basis[k], atol=1e-12)
    for i in range(3):
        for j in range(i + 1, 3):
            assert abs(np.dot(basis[i], basis[j])) < 1e-12
    np.testing.assert_allclose(antenna.polarisation, basis[0])


@pytest.mark.parametrize("antenna,point", antennas())
def test_double_conversion(antenna, point):
    back = antenna.pp_transform(antenna.pp_transform(point, "W2A"), "A2W")
    np.testing.assert_allclose(back, point, atol=1e-9)
    back = antenna.vv_transform(antenna.vv_transform(point, "W2A"), "A2W")
    np.testing.assert_allclose(back, point, atol=1e-9)


@pytest.mark.parametrize("antenna,point", antennas())
def test_incident_angles(antenna, point):
    for axis, el_deg, az_deg in ((np.array([10.0, 0, 0]), 90, 0), (np.array([0, 10.0, 0]), 90, 90),
                                 (np.array([0, 0, 10.0]), 0, None)):
        r, el, az = antenna.incident_angles(antenna.pp_transform(axis, "A2W"))
        assert abs(r - 10) < 1e-9 and abs(np.degrees(el) - el_deg) < 1e-6
        if az_deg is not None:
            assert abs(np.degrees(az) - az_deg) < 1e-6


def test_split_reflections_path():
    from raytracing_amd.electromagnetism import Reflection

    p = [[0, 0, 0], [1, 1, 1], [2, 2, 2], [3, 3, 3]]
    assert Reflection.split_reflections_path(p) == [[[0, 0, 0], [1, 1, 1], [2, 2, 2]], [[1, 1, 1], [2, 2, 2], [3, 3, 3]]]
    p = [[0, 0, 0], [1, 1, 1], [2, 2, 2]]
    assert Reflection.split_reflections_path(p) == [[[0, 0, 0], [1, 1, 1], [2, 2, 2]]]
    p = [[i, i, i] for i in range(6)]
    assert Reflection.split_reflections_path(p) == [[[i, i, i], [i + 1] * 3, [i + 2] * 3] for i in range(4)]
    with pytest.raises(AssertionError):
        Reflection.split_reflections_path(p[:2])
