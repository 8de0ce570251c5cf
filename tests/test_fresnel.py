"""The library's Fresnel integrals (raytracing_amd/csrc/rt_fresnel.h: fixed-length Chebyshev series for S, C and the
auxiliary functions f, g) compiled for the host and held against scipy.special.fresnel -- the function the reference
calls (electromagnetism.py:355-363) -- and against mpmath at 40 digits.  The GPU runs the same header."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import scipy.special

from conftest import ROOT

NATIVE = os.path.join(ROOT, "tests", "native")


@pytest.fixture(scope="module")
def lib():
    src = os.path.join(NATIVE, "fresnel_host.cpp")
    lib_path = os.path.join(NATIVE, "libfresnel_host.so")
    headers = [os.path.join(ROOT, "raytracing_amd", "csrc", h) for h in ("rt_fresnel.h", "rt_fresnel_tables.h")]
    if not os.path.exists(lib_path) or os.path.getmtime(lib_path) < max(os.path.getmtime(p) for p in [src] + headers):
        subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-o", lib_path, src, "-lm"])
    so = ctypes.CDLL(lib_path)
    so.rt_fresnel_host.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p]
    so.rt_fresnel_fg_host.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p]
    so.rt_fresnel_x0.restype = ctypes.c_double
    return so


def fresnel(lib, x):
    x = np.ascontiguousarray(x, dtype=float)
    s, c = np.empty_like(x), np.empty_like(x)
    lib.rt_fresnel_host(x.ctypes.data, len(x), s.ctypes.data, c.ctypes.data)
    return s, c


def test_integrals_match_scipy(lib):
    rng = np.random.default_rng(1)
    x0 = lib.rt_fresnel_x0()
    edges = np.array([0.0, 1e-300, 1e-9, x0, 2.8, 5.0])
    x = np.concatenate([rng.uniform(0, 10, 200_000), edges, np.nextafter(edges, 0), np.nextafter(edges, 10), -rng.uniform(0, 10, 1000)])
    s, c = fresnel(lib, x)
    s_ref, c_ref = scipy.special.fresnel(x)
    # both sides take sine and cosine of pi x^2 / 2 in double precision, whose rounding grows with x^2: |f| pi x^2 / 2 eps
    # with f ~ 1 / (pi x)
    bound = 5e-16 + 2e-16 * np.abs(x)
    assert (np.abs(s - s_ref) < bound).all() and (np.abs(c - c_ref) < bound).all()
    small = np.abs(x) < x0
    assert np.abs(s - s_ref)[small].max() < 6e-16 and np.abs(c - c_ref)[small].max() < 6e-16  # values up to 0.8: a few ulp
    x = 10 ** rng.uniform(1, 4, 50_000)
    s, c = fresnel(lib, x)
    s_ref, c_ref = scipy.special.fresnel(x)
    bound = 5e-16 + 2e-16 * x
    assert (np.abs(s - s_ref) < bound).all() and (np.abs(c - c_ref) < bound).all()
    s, c = fresnel(lib, np.array([np.inf, -np.inf, 1e200]))
    np.testing.assert_allclose(s, [0.5, -0.5, 0.5], atol=1e-15)
    np.testing.assert_allclose(c, [0.5, -0.5, 0.5], atol=1e-15)


def test_auxiliary_functions_match_mpmath(lib):
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 40
    rng = np.random.default_rng(2)
    x0 = lib.rt_fresnel_x0()
    x = np.concatenate([rng.uniform(x0, 8, 1500), 10 ** rng.uniform(0.9, 6, 500), [x0, 2.8, 5.0, np.nextafter(2.8, 0), np.nextafter(5.0, 0)]])
    f, g = np.empty_like(x), np.empty_like(x)
    lib.rt_fresnel_fg_host(x.ctypes.data, len(x), f.ctypes.data, g.ctypes.data)
    half = mp.mpf(1) / 2
    for xi, fi, gi in zip(x, f, g):
        xm = mp.mpf(float(xi))
        z = mp.pi * xm * xm / 2
        s, c = mp.fresnels(xm), mp.fresnelc(xm)
        f_true = (half - s) * mp.cos(z) - (half - c) * mp.sin(z)
        g_true = (half - c) * mp.cos(z) + (half - s) * mp.sin(z)
        # relative to the function's own size: the transition function multiplies them by 2 sqrt(x) sqrt(pi/2)
        assert abs(fi - f_true) < 4e-16 * abs(f_true) + 1e-300, (xi, fi, f_true)
        # (g is the smaller of the two by pi x^2: a few ulp of its own, far below an ulp of f)
        assert abs(gi - g_true) < 2e-15 * abs(g_true) + 1e-300 and abs(gi - g_true) < 1e-16 * abs(f_true) + 1e-300, (xi, gi, g_true)


def test_transition_function_through_f_and_g_equals_the_reference_formula(lib):
    """F(x) = 2 sqrt(x) sqrt(pi/2) (f + j g) against the reference's expression with SciPy's integrals."""
    rng = np.random.default_rng(3)
    x0 = lib.rt_fresnel_x0()
    factor = np.sqrt(np.pi / 2)
    x = np.concatenate([rng.uniform((x0 * factor) ** 2, 60, 5000), 10 ** rng.uniform(1.5, 3, 1000)])
    t = np.sqrt(x) / factor
    f, g = np.empty_like(t), np.empty_like(t)
    lib.rt_fresnel_fg_host(np.ascontiguousarray(t).ctypes.data, len(t), f.ctypes.data, g.ctypes.data)
    mine = 2 * np.sqrt(x) * factor * (f + 1j * g)
    s_ref, c_ref = scipy.special.fresnel(t)
    ref = 2j * np.sqrt(x) * np.exp(1j * x) * (factor * ((1 - 1j) / 2 - c_ref + 1j * s_ref))  # electromagnetism.py:355-363
    # the reference's form loses digits to the cancellation of C and S against 1/2 as x grows: about eps x in F
    assert (np.abs(mine - ref) < 5e-15 + 1e-15 * x).all()
