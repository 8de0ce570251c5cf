"""CPU ORACLE — test infrastructure only.  Never imported by the product path.

ctypes front end of ``visibility_oracle.c`` (the restatement of the reference's ``polygon_visibility_vector`` and
``polygons_visibility_matrix``, raytracing/geometry.py:924-1048).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libvisibility_oracle.so"])


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libvisibility_oracle.so")
        src = os.path.join(_HERE, "visibility_oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.oracle_face_visibility.restype = ctypes.c_int
        _LIB.oracle_face_visibility.argtypes = [
            ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
            ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    return _LIB


class PackedScene:
    def __init__(self, polys):
        self.n = len(polys)
        counts = [len(p.points) for p in polys]
        self.vert_off = np.zeros(self.n + 1, dtype=np.int32)
        np.cumsum(counts, out=self.vert_off[1:])
        self.verts = np.ascontiguousarray(np.concatenate([p.points for p in polys]), dtype=np.float64)
        self.planes = np.ascontiguousarray(np.array([p.parametric for p in polys]), dtype=np.float64)


def polygon_visibility_vector(poly_a, polys, self_index=-1, out=None, packed=None, return_stage=False):
    """geometry.py:924-1019; ``out`` (bool, N) is OR-ed into; a fresh one starts all-False."""
    packed = packed or PackedScene(polys)
    vis = np.zeros(packed.n, dtype=np.uint8)
    stage = np.zeros(packed.n, dtype=np.uint8)
    a_pts = np.ascontiguousarray(poly_a.points, dtype=np.float64)
    a_normal = np.ascontiguousarray(poly_a.parametric[:3], dtype=np.float64)
    _lib().oracle_face_visibility(a_pts.ctypes.data, len(a_pts), a_normal.ctypes.data, packed.n,
                                  packed.vert_off.ctypes.data, packed.verts.ctypes.data, packed.planes.ctypes.data,
                                  int(self_index), vis.ctypes.data, stage.ctypes.data)
    vis = vis.astype(bool)
    if out is not None:
        out |= vis
        vis = out
    return (vis, stage) if return_stage else vis


def polygons_visibility_matrix(polys, rows=None):
    """geometry.py:1036-1048.  ``rows`` restricts the work to some rows (the result is then NOT symmetrised)."""
    packed = PackedScene(polys)
    n = packed.n
    matrix = np.zeros((n, n), dtype=bool)
    for i in (range(n) if rows is None else rows):
        polygon_visibility_vector(polys[i], polys, self_index=i, out=matrix[i, :], packed=packed)
    if rows is None:
        matrix = matrix | matrix.T
    return matrix


def polygon_visibility_vector_strict(poly_a, polys, self_index=-1):
    """The strict branch of the reference's polygon_visibility_vector (geometry.py:959-970 then :1004-1017 with the
    painter of :974-1001 left out): polygon B is kept when, moved to A's frame about the centre of A's box, it reaches
    z > 1e-4, and some vertex p of A has normalize(p - centre(B)) . n_B > 1e-4.  Plain NumPy.  Parity unpinned: the
    reference holds no vector of a visibility matrix, strict or not."""
    from . import geometry_oracle as go

    vis = np.zeros(len(polys), dtype=bool)
    a_pts = np.asarray(poly_a.points, dtype=float)
    matrix_a = poly_a.matrix().T                 # :959
    centre_a = go.centroid(a_pts)                # :960, centre of the bounding box
    for i, b in enumerate(polys):
        if i == self_index:                      # :964
            continue
        projected = (np.asarray(b.points, dtype=float) - centre_a) @ matrix_a.T   # translate, then project (:963)
        if not projected[:, 2].max() > 1e-4:     # :967-970
            continue
        centre_b, n_b = go.centroid(np.asarray(b.points, dtype=float)), b.normal
        for p in a_pts:                          # :1004-1014
            v = p - centre_b
            nv = np.linalg.norm(v)
            v = v / nv if nv != 0 else v
            if np.dot(v, n_b) > 1e-4:
                vis[i] = True
                break
    return vis


def polygons_visibility_matrix_strict(polys):
    """geometry.py:1036-1048 with strict=True."""
    n = len(polys)
    m = np.zeros((n, n), dtype=bool)
    for i, a in enumerate(polys):
        m[i] = polygon_visibility_vector_strict(a, polys, i)
    return m | m.T
