"""The library's restatement of the reference's solver (raytracing_amd/csrc/rt_hybrd.h: MINPACK hybr on the reference's
zero functions) compiled for the host and held against SciPy: the vectors the reference's own functions produced
(tests/golden/solvers.json) bit for bit, and the path set the oracle finds with SciPy's hybr on config 1
(tests/golden/hybr_vs_exact.json).  The GPU runs the same header (tests/test_gpu_hybr_parity.py)."""
import ctypes
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

NATIVE = os.path.join(ROOT, "tests", "native")


@pytest.fixture(scope="module")
def host_solver():
    src = os.path.join(NATIVE, "hybr_host.cpp")
    lib_path = os.path.join(NATIVE, "libhybr_host.so")
    header = os.path.join(ROOT, "raytracing_amd", "csrc", "rt_hybrd.h")
    if not os.path.exists(lib_path) or os.path.getmtime(lib_path) < max(os.path.getmtime(src), os.path.getmtime(header)):
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", lib_path, src, "-lm"])
    lib = ctypes.CDLL(lib_path)
    lib.rt_hybr_host_solve.restype = ctypes.c_int
    lib.rt_hybr_host_solve.argtypes = [ctypes.c_void_p] * 7

    def solve(origin, destination, planes, edge):
        a = np.ascontiguousarray(origin, dtype=float).reshape(3)
        b = np.ascontiguousarray(destination, dtype=float).reshape(3)
        k = len(planes)
        pl = np.ascontiguousarray(np.array(planes, dtype=float).reshape(-1, 4)) if k else np.zeros((1, 4))
        e = None if edge is None else np.ascontiguousarray(edge, dtype=float).reshape(2, 3)
        out = np.zeros((k + (1 if e is not None else 0), 3))
        nfev = ctypes.c_int(0)
        ok = lib.rt_hybr_host_solve(a.ctypes.data, b.ctypes.data, pl.ctypes.data, k, e.ctypes.data if e is not None else None,
                                    out.ctypes.data, ctypes.byref(nfev))
        return out, bool(ok), nfev.value

    return solve


def test_restatement_reproduces_scipy_on_the_reference_vectors(host_solver):
    with open(os.path.join(GOLDEN, "solvers.json")) as f:
        cases = json.load(f)
    planes = np.load(os.path.join(GOLDEN, "helpers.npz"))["cube_planes"]
    n = 0
    for kind in ("reflection", "diffraction", "mixed"):
        for case in cases[kind]:
            pl = [planes[f] for f in case.get("faces", [])]
            edge = np.array(case["edge"]) if "edge" in case else None
            x, ok, nfev = host_solver(case["a"], case["b"], pl, edge)
            assert ok == case["success"]
            np.testing.assert_array_equal(x, np.array(case["x"]))  # bit for bit
            if "nfev" in case:
                # SciPy's count is two above the evaluations the algorithm itself makes (1 + n per Jacobian + 1 per
                # step), on every one of the vectors
                assert case["nfev"] == nfev + 2
            n += 1
    assert n == 154


def test_restatement_keeps_the_paths_scipy_keeps_on_config_1(host_solver):
    """The oracle's solve of config 1 (16 receivers, order 2 with diffraction, 4 032 path-solves) with the restatement in
    place of SciPy keeps exactly the paths the fixture lists for SciPy's hybr."""
    import parity_configs as pc
    from oracle import ray_tracing_oracle as ro

    with open(os.path.join(GOLDEN, "hybr_vs_exact.json")) as f:
        want = json.load(f)["c1_small"]
    w = pc.workload("c1_small")
    polys = pc.oracle_polys_with_buildings(w["place"])

    class Sol:
        def __init__(self, ok):
            self.success = ok

    oracle = ro.RayTracingOracle(w["tx"], polys, w["rx_all"], solver="hybr")
    oracle.solve_reflections = lambda o, d, planes: (lambda r: (r[0], Sol(r[1])))(host_solver(o, d, planes, None))
    oracle.solve_mixed = lambda o, d, planes, edge: (lambda r: (r[0], Sol(r[1])))(host_solver(o, d, planes, edge))
    oracle.solve(max_order=2)
    got = sorted(pc.path_key(*p) for p in oracle.path_set())
    assert got == sorted(pc.path_key(*p) for p in want["paths_hybr"])
    assert len(want["only_exact"]) == 2 and not want["only_hybr"]
