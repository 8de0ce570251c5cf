"""The library's JSON encoder of the solved problem's path containers (host code, no GPU) against json.dumps of the
nested lists the reference's to_json writes (ray_tracing.py:66-132)."""
import json

import numpy as np


def test_paths_json_is_what_json_dumps_writes():
    from raytracing_amd import _lib

    rng = np.random.default_rng(0)
    n, n_rx = 20000, 300
    rx = np.sort(rng.integers(0, n_rx, n))
    order = np.zeros(n, np.int32)
    for r in np.unique(rx):  # within a receiver: order 1 first, then order 2
        sel = np.flatnonzero(rx == r)
        k = rng.integers(0, len(sel) + 1)
        order[sel[:k]], order[sel[k:]] = 1, 2
    n_points = order + 2
    pts = rng.normal(size=(n, 4, 3)) * 10.0 ** rng.integers(-7, 18, size=(n, 4, 3))
    pts[rng.random((n, 4, 3)) < 0.05] = 0.0
    integral = rng.random((n, 4, 3)) < 0.05
    pts[integral] = np.round(pts[integral])
    # where Python's repr changes form, and the ends of the range
    pts[0, 0] = [1e16, 1e15, -0.0]
    pts[0, 1] = [1e-4, 1e-5, 123456789012345680.0]
    pts[0, 2] = [9999999999999998.0, 0.00010000000000000002, 9.999999999999999e-05]
    pts[1, 0] = [5e-324, 1.7976931348623157e308, 0.1 + 0.2]
    pts[1, 1] = [1.5, -2.25, 100.0]
    pts[1, 2] = [12345.678, 1e22, 2.2250738585072014e-308]
    chain = rng.integers(0, 7000, (n, 2)).astype(np.int32)
    keys = np.arange(n_rx + 5)  # the last receivers have no path

    got = _lib.format_paths_json(keys, rx, order, n_points, pts, order, chain)
    want = {int(k): {} for k in keys}
    for i in range(n):
        want[int(rx[i])].setdefault(int(order[i]), []).append([pts[i, :n_points[i]].tolist(), chain[i, :order[i]].tolist()])
    assert got == json.dumps(want)
    assert json.loads(got) == json.loads(json.dumps(want))

    # diffraction items carry the edge's polygons and end points; a chain may be empty
    ek = rng.integers(0, 7000, (n, 2)).astype(np.int32)
    ep = rng.normal(size=(n, 2, 3)) * 100
    got = _lib.format_paths_json(keys, rx, order, n_points, pts, order - 1, chain, ek, ep)
    want = {int(k): {} for k in keys}
    for i in range(n):
        want[int(rx[i])].setdefault(int(order[i]), []).append(
            [pts[i, :n_points[i]].tolist(), chain[i, :order[i] - 1].tolist(), ek[i].tolist(), ep[i].tolist()])
    assert got == json.dumps(want)

    assert _lib.format_paths_json([3, 4], [], [], [], np.zeros((0, 4, 3)), [], np.zeros((0, 2))) == json.dumps({3: {}, 4: {}})
    assert _lib.format_paths_json([], [], [], [], np.zeros((0, 4, 3)), [], np.zeros((0, 2))) == "{}"
