"""TEST INFRASTRUCTURE ONLY (CPU restatement; the product never imports this).

Radiomap arithmetic of electromagnetism_fun/scenario_measures.py:49-67.

* ``interpolate``: the reference's own call, ``scipy.interpolate.griddata(method="linear")`` (:66) — SciPy is the
  reference's dependency and is installed here, so the oracle for the interpolation IS the reference's arithmetic.
* ``contains``: ``shapely.geometry.Polygon.contains(Point)`` (:54) restated for a ring without holes (Shapely/GEOS is
  absent here): True strictly inside, False on the boundary and outside.  Pure-Python exact arithmetic
  (fractions) so that "on the boundary" is decided without rounding.  Parity unpinned: no reference fixture holds a
  mask; known answers in tests/test_radiomap.py follow from the definition."""
from fractions import Fraction

import numpy as np


def interpolate(points, values, x, y, fill_value):
    from scipy.interpolate import griddata

    return griddata(np.asarray(points, dtype=float), np.asarray(values, dtype=float), (x, y), method="linear",
                    fill_value=fill_value)


def contains(ring, px, py):
    ring = [(Fraction(float(a)), Fraction(float(b))) for a, b in np.asarray(ring, dtype=float)]
    if ring[0] == ring[-1]:
        ring = ring[:-1]
    px, py = Fraction(float(px)), Fraction(float(py))
    inside = False
    n = len(ring)
    for k in range(n):
        (ax, ay), (bx, by) = ring[k - 1], ring[k]
        cross = (bx - ax) * (py - ay) - (by - ay) * (px - ax)
        if cross == 0 and min(ax, bx) <= px <= max(ax, bx) and min(ay, by) <= py <= max(ay, by):
            return False  # on the boundary: not contained
        if (ay > py) != (by > py):
            # the edge crosses the horizontal line through the point: to the right of the point?
            if (cross > 0) == (by > ay):
                inside = not inside
    return inside


def contains_many(ring, x, y):
    return np.array([contains(ring, a, b) for a, b in zip(np.ravel(x), np.ravel(y))], dtype=bool).reshape(np.shape(x))
