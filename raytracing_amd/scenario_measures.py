"""Radiomap of a result table with the reference's steps (``electromagnetism_fun/scenario_measures.py:31-88``,
``compare_radiomaps``): a regular grid over the bounds of the street polygon (:49-53), the per-receiver power
interpolated linearly over it (``griddata(..., method="linear", fill_value=min(powers))``, :66) and the grid points the
street polygon contains (:54).  The figure itself is not part of this package: ``radiomap`` returns what the reference
hands to ``pcolormesh`` (:67-68).

The interpolation and the containment test run on the GPU (``rt_radiomap``).  The Delaunay triangulation of the
receivers is Qhull's, through ``scipy.spatial.Delaunay``: it is what ``griddata`` builds too, so a grid point gets the
triangle scipy would give it."""
import numpy as np

from . import _lib
from . import electromagnetism_utils


def _ring_of(street):
    """Outer ring of the street polygon as an (n, 2) array without the closing vertex: an array of vertices, or
    anything with ``exterior.coords`` (a shapely polygon)."""
    if hasattr(street, "exterior"):
        if len(getattr(street, "interiors", ())) > 0:
            raise ValueError("radiomap: street polygons with holes are not supported")
        street = np.asarray(street.exterior.coords, dtype=float)
    ring = np.asarray(street, dtype=float).reshape(-1, 2 if np.asarray(street).shape[-1] == 2 else 3)[:, :2]
    if len(ring) > 1 and np.array_equal(ring[0], ring[-1]):
        ring = ring[:-1]
    if len(ring) < 3:
        raise ValueError("radiomap: the street polygon needs at least 3 vertices")
    return np.ascontiguousarray(ring)


def radiomap_grid(street, grid_resolution=300):
    """The reference's grid (:49-53): ``meshgrid`` of ``grid_resolution`` abscissae and ordinates over the bounds."""
    ring = _ring_of(street)
    (x0, y0), (x1, y1) = ring.min(axis=0), ring.max(axis=0)
    return np.meshgrid(np.linspace(x0, x1, grid_resolution), np.linspace(y0, y1, grid_resolution))


def interpolate_on_grid(points, values, x, y, fill_value):
    """``griddata(points, values, (x, y), method="linear", fill_value=fill_value)`` (:66)."""
    from scipy.spatial import Delaunay

    points = np.asarray(points, dtype=float).reshape(-1, 2)
    values = np.asarray(values, dtype=float).reshape(-1)
    if len(points) != len(values):
        raise ValueError("different number of values and points")
    tri = Delaunay(points)
    query = np.column_stack((np.ravel(x), np.ravel(y)))
    z, _ = _lib.radiomap(values, tri.simplices, tri.transform, query, fill_value)
    return z.reshape(np.shape(x))


def street_mask(street, x, y):
    """``street.contains(Point(x_, y_))`` for every grid point (:54), as a boolean array of the grid's shape."""
    query = np.column_stack((np.ravel(x), np.ravel(y)))
    _, inside = _lib.radiomap(None, None, None, query, 0.0, ring=_ring_of(street), interpolate=False)
    return inside.reshape(np.shape(x))


def radiomap(df, street, grid_resolution=300):
    """(x, y, z_masked) of one table as ``compare_radiomaps`` computes them (:49-67): receivers and powers from the
    table (:59-60), interpolation and mask in one call to the library."""
    from scipy.spatial import Delaunay

    receivers = electromagnetism_utils.get_receiver_coordinates(df, two_dimensional=True)
    powers = np.asarray(electromagnetism_utils.get_power_db_each_receiver(df), dtype=float)
    x, y = radiomap_grid(street, grid_resolution)
    tri = Delaunay(np.asarray(receivers, dtype=float))
    query = np.column_stack((x.ravel(), y.ravel()))
    z, inside = _lib.radiomap(powers, tri.simplices, tri.transform, query, np.min(powers), ring=_ring_of(street))
    z = z.reshape(x.shape)
    return x, y, np.ma.masked_where(np.logical_not(inside.reshape(x.shape)), z)
