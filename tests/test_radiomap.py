"""Radiomap (electromagnetism_fun/scenario_measures.py:49-67): the oracle's containment rule on known answers, and the
GPU interpolation / mask against scipy's griddata (the reference's own call) and the oracle."""
import numpy as np
import pytest

from oracle import radiomap_oracle as oracle

# an L-shaped street with a notch: concave, with horizontal and vertical edges the grid lines fall on
STREET = np.array([[0.0, 0.0], [10.0, 0.0], [10.0, 4.0], [6.0, 4.0], [6.0, 5.0], [4.0, 9.0], [0.0, 9.0]])


def test_oracle_contains_known_answers():
    cases = {(1.0, 1.0): True, (9.0, 3.0): True, (5.0, 6.0): True, (3.0, 8.5): True,
             (0.0, 0.0): False, (5.0, 0.0): False, (10.0, 2.0): False, (6.0, 4.5): False, (5.0, 7.0): False,  # boundary
             (7.0, 5.0): False, (-1.0, 1.0): False, (11.0, 0.0): False, (5.5, 8.0): False, (2.0, 9.5): False}
    for (x, y), want in cases.items():
        assert oracle.contains(STREET, x, y) is want, (x, y)
    closed = np.vstack((STREET, STREET[:1]))
    assert oracle.contains(closed, 1.0, 1.0) and not oracle.contains(closed, 0.0, 3.0)


def test_oracle_interpolation_is_griddata():
    pts = np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0], [1.0, 1.0]])
    vals = np.array([0.0, 1.0, 2.0, 3.0])  # a plane: z = x + 2 y
    x, y = np.meshgrid(np.linspace(-0.5, 1.5, 9), np.linspace(0.0, 1.0, 5))
    z = oracle.interpolate(pts, vals, x, y, -7.0)
    inside = (x >= 0) & (x <= 1)
    np.testing.assert_allclose(z[inside], (x + 2 * y)[inside], atol=1e-15)
    assert (z[~inside] == -7.0).all()


def receivers_and_powers(n, seed):
    rng = np.random.default_rng(seed)
    pts = np.column_stack((rng.uniform(-1.0, 11.0, n), rng.uniform(-1.0, 10.0, n)))
    powers = -60.0 - 40.0 * rng.random(n)
    return pts, powers


@pytest.mark.gpu
@pytest.mark.parametrize("n,res", [(3, 17), (40, 64), (700, 300), (5000, 300)])
def test_gpu_interpolation_equals_griddata(n, res):
    """Same triangulation and the same unfused sums as scipy: a grid point that lies in one triangle only gets scipy's
    value bit for bit; on a shared edge either neighbour may be picked and the two values differ by rounding
    (tolerance 1e-12 dB), so nearly all points must be identical and all within the tolerance."""
    from raytracing_amd import scenario_measures as sm

    pts, powers = receivers_and_powers(n, n)
    x, y = sm.radiomap_grid(STREET, res)
    fill = float(np.min(powers))
    z = sm.interpolate_on_grid(pts, powers, x, y, fill)
    want = oracle.interpolate(pts, powers, x, y, fill)
    np.testing.assert_allclose(z, want, rtol=0, atol=1e-12)
    assert np.mean(z == want) > 0.999


@pytest.mark.gpu
def test_gpu_interpolation_on_a_lattice_of_receivers():
    """Receivers on a regular lattice (the coverage-map case): Qhull's triangles then have grid lines on their edges,
    and every query point on a lattice line sits on a shared edge."""
    from raytracing_amd import scenario_measures as sm

    gx, gy = np.meshgrid(np.linspace(0.0, 10.0, 21), np.linspace(0.0, 9.0, 19))
    pts = np.column_stack((gx.ravel(), gy.ravel()))
    powers = -70.0 + 10.0 * np.sin(pts[:, 0]) * np.cos(0.7 * pts[:, 1])
    x, y = sm.radiomap_grid(STREET, 181)
    z = sm.interpolate_on_grid(pts, powers, x, y, -200.0)
    want = oracle.interpolate(pts, powers, x, y, -200.0)
    np.testing.assert_allclose(z, want, rtol=0, atol=1e-12)
    assert not (z == -200.0).any()


@pytest.mark.gpu
def test_gpu_mask_equals_contains():
    from raytracing_amd import scenario_measures as sm

    x, y = sm.radiomap_grid(STREET, 91)  # steps of 1/9 and 1/10: many grid points on the ring's edges and vertices
    got = sm.street_mask(STREET, x, y)
    want = oracle.contains_many(STREET, x, y)
    np.testing.assert_array_equal(got, want)
    assert want.any() and not want.all()
    assert not got[0, :].any() and not got[:, 0].any()  # the bounds are boundary: not contained


@pytest.mark.gpu
def test_gpu_radiomap_of_a_table():
    import pandas as pd

    from raytracing_amd import scenario_measures as sm

    pts, _ = receivers_and_powers(60, 3)
    rng = np.random.default_rng(4)
    rows = []
    for i, p in enumerate(pts):
        for _ in range(int(rng.integers(1, 4))):
            rows.append({"rx_id": i, "receiver": np.array([p[0], p[1], 1.2]), "path_type": "R",
                         "path_power": float(10.0 ** rng.uniform(-12, -8))})
    df = pd.DataFrame(rows)
    x, y, zm = sm.radiomap(df, STREET, 50)
    from raytracing_amd import electromagnetism_utils as emu

    powers = np.asarray(emu.get_power_db_each_receiver(df), dtype=float)
    want = oracle.interpolate(pts, powers, x, y, np.min(powers))
    mask = oracle.contains_many(STREET, x, y)
    np.testing.assert_array_equal(np.ma.getmaskarray(zm), ~mask)
    np.testing.assert_allclose(zm.data, want, rtol=0, atol=1e-12)
