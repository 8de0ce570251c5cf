"""The field-step oracle against vectors produced by the reference's own field code (tests/golden/make_golden_em.py)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

from oracle import em_oracle as eo
from oracle import geometry_oracle as go


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLDEN, "em.npz"))


def test_antenna_frames_and_source_field(g):
    fo = eo.FieldOracle([])
    for p, t, q, mat, ang, E in zip(g["ant_pos"], g["ant_target"], g["ant_probe"], g["ant_mat"], g["ant_angles"],
                                    g["src_field"]):
        a = eo.Antenna(p)
        a.align_antenna_to_point(t)
        np.testing.assert_array_equal(a.mat, mat)
        np.testing.assert_array_equal(np.array(a.incident_angles(q)), ang)
        np.testing.assert_array_equal(fo.from_path([p, q], a), E)
    a = eo.Antenna(np.array([1.0, 2.0, 3.0]))
    a.align_antenna_to_point(np.array([1.0, 2.0, 9.0]))
    np.testing.assert_array_equal(a.mat, g["ant_vertical_mat"])


def test_fresnel_coefficients(g):
    fo = eo.FieldOracle([])
    material = {"office": "glass", "appartments": "brick", "garage": "concrete", "ground": "medium_dry_ground",
                "tree_trunk": "wood"}
    for t, row in zip(g["fresnel_types"], g["fresnel_coeffs"]):
        poly = go.Poly(np.array([[0.0, 0, 0], [1, 0, 0], [1, 1, 0]]), material=material[str(t)])
        for th, (r_par, r_per) in zip(g["fresnel_thetas"], row):
            got = fo.fresnel_coeffs(th, poly)
            np.testing.assert_allclose(got, (r_par, r_per), rtol=1e-13)


def test_reflection_dyadic_and_field(g):
    face = go.Poly(g["refl_polygon"], material="brick")
    face.parametric = g["refl_parametric"]
    fo = eo.FieldOracle([None, None, face])
    fo.account_for_trees = lambda E, points: E  # as in the generator of the vectors (make_golden_em.py)
    ant = eo.Antenna(g["refl_tx"])
    ant.align_antenna_to_point(g["refl_target"])
    for path, R, E in zip(g["refl_paths"], g["refl_R"], g["refl_E"]):
        np.testing.assert_allclose(fo.dyadic_ref_coeff(path, face), R, rtol=1e-12, atol=1e-15)
        E_i = fo.from_path([path[0], path[1]], ant)
        np.testing.assert_allclose(fo.my_reflect(E_i, path, [2]), E, rtol=1e-12, atol=1e-18)


def test_transition_function_and_A(g):
    for x, f in zip(g["F_x"], g["F_val"]):
        np.testing.assert_array_equal(eo.FieldOracle.F(x), f)
    for x, n, ap, am in zip(g["A_x"], g["A_n"], g["A_plus"], g["A_minus"]):
        assert eo.FieldOracle.A(x, n, "plus") == ap and eo.FieldOracle.A(x, n, "minus") == am


def test_wedge_diffraction(g):
    f0 = go.Poly(g["diff_face0"], material="brick")
    f0.parametric = g["diff_par0"]
    f1 = go.Poly(g["diff_face1"], material="brick")
    f1.parametric = g["diff_par1"]
    fo = eo.FieldOracle([None, None, f0, f1])
    fo.account_for_trees = lambda E, points: E  # as in the generator of the vectors (make_golden_em.py)
    assert len(g["diff_paths"]) >= 30
    for path, comps, E, E_i in zip(g["diff_paths"], g["diff_D"], g["diff_E"], g["diff_Ei"]):
        got = fo.diff_dyadic_components(path, f0, f1, np.array([0.0, 0.0, 1.0]))
        np.testing.assert_allclose(np.array(got), comps, rtol=1e-12, atol=1e-18)
        np.testing.assert_allclose(fo.my_diff(E_i, path, (2, 3), g["diff_corner"]), E, rtol=1e-11, atol=1e-18)
