"""Error behaviour of the C ABI: every entry point returns a status (or a null handle) and leaves its message in
rt_last_error(); nothing is launched on bad input.  Called through ctypes the way a foreign binding would."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def lib():
    from raytracing_amd import _lib

    return _lib.require_gpu()


def message():
    return lib().rt_last_error().decode()


def triangle_scene(material=1, first_offset=0, n_vertices=3):
    off = np.array([first_offset, first_offset + n_vertices], dtype=np.int32)
    verts = np.array([[0.0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=np.float64)
    planes = np.array([[0.0, 0, 1, 0]], dtype=np.float64)
    mat = np.array([material], dtype=np.int32)
    return lib().rt_scene_create(0, 1, off.ctypes.data, verts.ctypes.data, planes.ctypes.data, mat.ctypes.data, None)


def test_scene_create_rejects_bad_input():
    L = lib()
    assert not L.rt_scene_create(0, 0, None, None, None, None, None) and "empty scene" in message()
    assert not triangle_scene(n_vertices=2) and "fewer than 3 vertices" in message()
    assert not triangle_scene(first_offset=1) and "start at 0" in message()
    assert not triangle_scene(material=7) and "material id 7" in message()
    assert not triangle_scene(material=-1) and "material id -1" in message()
    off = np.array([0, 3], dtype=np.int32)
    v = np.zeros((3, 3))
    p = np.zeros((1, 4))
    assert not L.rt_scene_create(99, 1, off.ctypes.data, v.ctypes.data, p.ctypes.data, None, None)
    assert "no such device" in message()
    h = triangle_scene()
    assert h and L.rt_scene_polygon_count(h) == 1
    L.rt_scene_destroy(h)


def test_entry_points_reject_null_and_negative_arguments():
    L = lib()
    h = triangle_scene()
    try:
        out = np.zeros(4, dtype=np.uint8)
        seg = np.array([[0.2, 0.2, -1.0, 0.2, 0.2, 1.0]])
        assert L.rt_obstruct_segments(h, seg.ctypes.data, 1, None, out.ctypes.data, 0) == 0 and out[0] == 1
        assert L.rt_obstruct_segments(h, seg.ctypes.data, -1, None, out.ctypes.data, 0) != 0 and message()
        assert L.rt_obstruct_segments(None, seg.ctypes.data, 1, None, out.ctypes.data, 0) != 0 and message()
        assert L.rt_obstruct_segments(h, None, 1, None, out.ctypes.data, 0) != 0 and message()
        assert L.rt_obstruct_segments(h, seg.ctypes.data, 0, None, out.ctypes.data, 0) == 0  # nothing to do is fine
        found = np.zeros(2, dtype=np.int64)
        tx = np.zeros(3)
        assert L.rt_solve_paths(h, None, tx.ctypes.data, None, 0, None, None, None, None, 0, found.ctypes.data,
                                found.ctypes.data + 8, 0) != 0
        assert "null argument" in message()
        assert L.rt_visibility_matrix(None, 0, out.ctypes.data, 0) != 0 and message()
    finally:
        L.rt_scene_destroy(h)
    L.rt_scene_destroy(None)        # destroying nothing is allowed
    L.rt_candidates_destroy(None)


def test_table_entry_points_reject_bad_input():
    L = lib()
    off = np.array([0, 2, 1], dtype=np.int64)  # decreasing
    x = np.zeros(2)
    xyz = np.zeros((2, 3))
    los = np.zeros(2, dtype=np.uint8)
    out = np.zeros((2, 8))
    args = (off.ctypes.data, x.ctypes.data, x.ctypes.data, x.ctypes.data, x.ctypes.data, xyz.ctypes.data, los.ctypes.data)
    assert L.rt_channel_stats(0, 2, *args, 1, 0.99, out.ctypes.data, 0) != 0 and "offsets" in message()
    off[:] = [0, 1, 2]
    assert L.rt_channel_stats(0, 2, *args, 1, 1.5, out.ctypes.data, 0) != 0 and "bad parameter" in message()
    assert L.rt_channel_stats(0, 2, *args, 1, 0.99, out.ctypes.data, 0) == 0
    # radiomap: triangle index out of range, ring too short, negative counts
    vals = np.zeros(3)
    tris = np.array([[0, 1, 3]], dtype=np.int32)
    tr = np.zeros((1, 6))
    q = np.zeros((1, 2))
    z = np.zeros(1)
    inside = np.zeros(1, dtype=np.uint8)
    assert L.rt_radiomap(0, 3, vals.ctypes.data, 1, tris.ctypes.data, tr.ctypes.data, 1, q.ctypes.data, 0.0,
                         z.ctypes.data, 0, None, None) != 0 and "out of range" in message()
    ring = np.zeros((2, 2))
    assert L.rt_radiomap(0, 0, None, 0, None, None, 1, q.ctypes.data, 0.0, None, 2, ring.ctypes.data,
                         inside.ctypes.data) != 0 and "at least 3" in message()
    assert L.rt_radiomap(0, 0, None, 0, None, None, -1, q.ctypes.data, 0.0, None, 0, None, None) != 0
    assert L.rt_radiomap(99, 0, None, 0, None, None, 1, q.ctypes.data, 0.0, z.ctypes.data, 0, None, None) != 0
    assert "no such device" in message()
    # no triangles: every point gets the fill value
    assert L.rt_radiomap(0, 0, None, 0, None, None, 1, q.ctypes.data, -7.0, z.ctypes.data, 0, None, None) == 0
    assert z[0] == -7.0
