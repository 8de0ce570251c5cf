"""Scene model and geometry entry points of the path-solve path.

Host side of the reference's ``raytracing/geometry.py``: the polygon / building / place containers with the
same names, construction rules and polygon ORDER (everything downstream — visibility matrix rows, candidate
chains, path records — is indexed by that order), plus the geometry entry points, which hand the work to the
HIP library through ``raytracing_amd._lib`` (no CPU implementation of those lives here).

Reference lines followed:
  plane / frame / bounding box of a polygon ........ geometry.py:1437-1465, 1747-1853
  building extrusion, emitter cube ................. geometry.py:2313-2415, 2508-2531
  place and its polygon order ...................... geometry.py:2534-2663
  GeoJSON -> place ................................. geometry.py:2806-2920
The reference builds scenes with GeoPandas/PROJ/GDAL and unseeded random heights; ``load_place`` does the same
steps with a small tolerant GeoJSON reader, an own ETRS89-LAEA (EPSG:3035) forward projection and a seeded
generator, so a scene is a pure function of (file, seed).
"""
import itertools
import json
import math
import random
import re
import os
import uuid

import numpy as np

from . import array_utils
from . import container_utils
from .materials_properties import (MATERIAL_ID, MAX_TREE_HEIGHT, MIN_TREE_HEIGHT, material_of_building_type,
                                   set_properties)


# --------------------------------------------------------------------------------------------------
# small numeric helpers (reference geometry.py:33, 72-87, 135-146, 150-181)
# --------------------------------------------------------------------------------------------------

def midpoint(p1, p2):
    return (p1 + p2) / 2


def normalize_path(points):
    """Unit vectors and lengths of the two legs of a 3-point path (reference geometry.py:72-87)."""
    assert len(points) == 3
    vectors = np.diff(points, axis=0)
    norms = np.linalg.norm(vectors, axis=1)
    return vectors[0] / norms[0], norms[0], vectors[1] / norms[1], norms[1]


def path_length(points):
    return np.sum(np.linalg.norm(np.diff(points, axis=0), axis=1))


def enclosed_area(points):
    """Signed shoelace sum of the xy projection; NEGATIVE for counter-clockwise (reference :150-167)."""
    x = points[:, 0]
    y = points[:, 1]
    s = np.sum((x[1:] - x[:-1]) * (y[1:] + y[:-1]))
    s += (x[0] - x[-1]) * (y[0] + y[-1])
    return s / 2


def is_ccw(points):
    return enclosed_area(points) < 0


def projection_matrix_from_vector(vector):
    """Orthonormal frame whose last axis is along ``vector`` (reference geometry.py:414-435)."""
    z = vector / np.linalg.norm(vector)
    y = np.ones_like(z)
    y = y - y.dot(z) * z
    y = y / np.linalg.norm(y)
    x = np.cross(y, z)
    matrix = np.empty((3, 3))
    matrix[:, 0] = x
    matrix[:, 1] = y
    matrix[:, 2] = z
    return matrix


# --------------------------------------------------------------------------------------------------
# containers
# --------------------------------------------------------------------------------------------------


def polygon_line_intersection(polygon, points):
    """Point where the line through ``points[0]`` and ``points[1]`` meets the plane of ``polygon``; None when the line
    runs along the plane (reference geometry.py:89-106: same operations in the same order)."""
    normal = polygon.get_normal()
    anchor = polygon.points[1, :]
    start = np.asarray(points[0], dtype=float)
    direction = np.asarray(points[1], dtype=float) - start
    direction = direction / np.linalg.norm(direction)
    along = np.dot(direction, normal)
    if abs(along) < 1e-6:
        return None
    offset = start - anchor
    return offset + (-np.dot(offset, normal) / along) * direction + anchor


def ring_contains(ring, x, y):
    """Boundary-inclusive point-in-polygon of the points (x, y) (arrays) in the 2-D ring (n, 2): crossing number plus an
    on-edge test, the predicate ``contains_point`` evaluates on the GPU (reference geometry.py:1778-1803 goes through
    Shapely's ``intersects``).  Host code, for the scene builders."""
    ring = np.asarray(ring, dtype=float)
    x = np.atleast_1d(np.asarray(x, dtype=float))
    y = np.atleast_1d(np.asarray(y, dtype=float))
    before = np.concatenate([ring[-1:], ring[:-1]])  # the vertex each side starts from
    ax, ay = before[:, 0][:, None], before[:, 1][:, None]
    bx, by = ring[:, 0][:, None], ring[:, 1][:, None]
    cross = (bx - ax) * (y[None, :] - ay) - (by - ay) * (x[None, :] - ax)
    on_edge = (cross == 0.0) & (np.minimum(ax, bx) <= x) & (x <= np.maximum(ax, bx)) & (np.minimum(ay, by) <= y) & \
              (y <= np.maximum(ay, by))
    crosses = ((ay > y) != (by > y)) & ((cross > 0.0) == (by > ay))
    return on_edge.any(axis=0) | (crosses.sum(axis=0) % 2 == 1)


def ring_distance(ring, x, y):
    """Distance from the points (x, y) to the boundary of the 2-D ring."""
    ring = np.asarray(ring, dtype=float)
    pts = np.column_stack([np.atleast_1d(x), np.atleast_1d(y)]).astype(float)
    a, b = np.roll(ring, 1, axis=0), ring
    ab = b - a
    t = np.einsum("pek,ek->pe", pts[:, None, :] - a[None], ab) / np.maximum(np.einsum("ek,ek->e", ab, ab), 1e-300)
    nearest = a[None] + np.clip(t, 0.0, 1.0)[..., None] * ab[None]
    return np.sqrt(((pts[:, None, :] - nearest) ** 2).sum(axis=2)).min(axis=1)


def _drop_collinear(ring, tol=1e-12):
    """The ring without repeated vertices and without vertices on the line through their neighbours."""
    ring = [np.asarray(p, dtype=float) for p in ring]
    changed = True
    while changed and len(ring) > 3:
        changed = False
        for i in range(len(ring)):
            a, b, c = ring[i - 1], ring[i], ring[(i + 1) % len(ring)]
            scale = max(np.abs(b - a).max(), np.abs(c - b).max(), 1e-300)
            if abs((b[0] - a[0]) * (c[1] - b[1]) - (b[1] - a[1]) * (c[0] - b[0])) <= tol * scale * scale:
                del ring[i]
                changed = True
                break
    return np.array(ring)


def _clip_half_plane(ring, p, q, keep_left):
    """Part of the 2-D ring on one side of the line p -> q (Sutherland-Hodgman step): the left side when
    ``keep_left``, else the right.  Points on the line belong to both."""
    def side(v):
        s = (q[0] - p[0]) * (v[1] - p[1]) - (q[1] - p[1]) * (v[0] - p[0])
        return s if keep_left else -s

    out = []
    for i in range(len(ring)):
        a, b = ring[i - 1], ring[i]
        sa, sb = side(a), side(b)
        if sa >= 0:
            out.append(a)
        if (sa > 0 and sb < 0) or (sa < 0 and sb > 0):
            out.append(a + (b - a) * (sa / (sa - sb)))
    return out


def ring_minus_convex(ring, convex):
    """Pieces of the 2-D ring that lie outside the convex ring ``convex``: for each of its sides, what is beyond that
    side and within the sides before it.  The pieces do not overlap; pieces without area are dropped."""
    convex = np.asarray(convex, dtype=float)
    if not is_ccw(convex):
        convex = convex[::-1]  # counter-clockwise: inside is to the left of every side
    pieces = []
    inside_so_far = [np.asarray(v, dtype=float) for v in ring]
    for i in range(len(convex)):
        p, q = convex[i - 1], convex[i]
        if len(inside_so_far) < 3:
            break
        beyond = _clip_half_plane(inside_so_far, p, q, keep_left=False)
        if len(beyond) >= 3:
            piece = _drop_collinear(beyond)
            if len(piece) >= 3 and abs(enclosed_area(piece)) > 1e-9:
                pieces.append(piece)
        inside_so_far = _clip_half_plane(inside_so_far, p, q, keep_left=True)
    return pieces



def _json_save(filename, data):
    with open(filename, "w") as f:
        json.dump(data, f)
    return data


def _json_data(data, filename):
    if data is None:
        with open(filename) as f:
            data = json.load(f)
    return dict(data)


def _random_id():
    """A random identifier in the layout of str(uuid.uuid4()) (a scene makes ten thousand of them)."""
    h = os.urandom(16).hex()
    return f"{h[:8]}-{h[8:12]}-4{h[13:16]}-{h[16:20]}-{h[20:]}"


class OrientedGeometry:
    """Base container (reference geometry.py:1093-1507): id, attributes and cached bounding box."""

    def __init__(self, **attributes):
        self.attributes = attributes
        self._id = attributes.pop("id", None)  # None: a random one, made when it is first asked for
        self.domain = None
        self.centroid = None
        self.visibility_matrix = None
        self.sharp_edges = None

    @property
    def id(self):
        """The identifier given at construction, or a random one (reference :1117-1123).  A scene holds ten thousand
        geometries whose identifiers nobody reads: the random ones are drawn on first use."""
        if self._id is None:
            self._id = _random_id()
        return self._id

    @id.setter
    def id(self, value):
        self._id = value

    def __eq__(self, other):
        """Same id and same sharp edges (reference :1142-1147); the subclasses add their own members."""
        return self.id == other.id and self.sharp_edges == other.sharp_edges

    __hash__ = object.__hash__

    def get_polygons_iter(self):
        raise NotImplementedError

    def get_polygons_list(self):
        return list(self.get_polygons_iter())

    def get_polygons_count(self):
        return sum(1 for _ in self.get_polygons_iter())

    def get_points(self):
        return np.concatenate([polygon.points for polygon in self.get_polygons_iter()])

    def get_domain(self, force=False):
        if force or self.domain is None:
            points = self.get_points()
            self.domain = np.array([points.min(axis=0), points.max(axis=0)])
        return self.domain

    def get_centroid(self, force=False):
        """Centre of the bounding box, not the centre of mass (reference :1452-1465)."""
        if force or self.centroid is None:
            self.centroid = self.get_domain().mean(axis=0)
        return self.centroid

    def get_visibility_matrix(self, strict=False, force=False):
        if force or self.visibility_matrix is None:
            self.visibility_matrix = polygons_visibility_matrix(self.get_polygons_list(), strict=strict)
        return self.visibility_matrix

    def get_sharp_edges(self, min_angle=10.0, force=False):
        if force or self.sharp_edges is None:
            self.sharp_edges = sharp_edges_visibility_dict(
                polygons_sharp_edges_iter(self.get_polygons_list(), min_angle=min_angle))
        return self.sharp_edges

    def apply_on_points(self, func, *args, **kwargs):
        raise NotImplementedError

    def translate(self, vector):
        vector = np.asarray(vector, dtype=float)
        return self.apply_on_points(lambda p: p.reshape(-1, 3) + vector.reshape(1, 3))

    def project(self, matrix, around_point=None):
        matrix = np.asarray(matrix, dtype=float)
        if around_point is None:
            return self.apply_on_points(lambda p: p @ matrix.T)
        a = np.asarray(around_point, dtype=float).reshape(1, 3)
        return self.apply_on_points(lambda p: (p - a) @ matrix.T + a)


class OrientedPolygon(OrientedGeometry):
    """Planar polygon given by its vertices in order (reference geometry.py:1572-1853)."""

    def __init__(self, points, **attributes):
        super().__init__(**attributes)
        self.points = np.asarray(points).astype(float)
        self.parametric = None
        self.matrix = None
        self.properties = None
        self.building_id = None
        self.part = "side"

    def __len__(self):
        return self.points.shape[0]

    def __str__(self):
        return (f"Polygon({len(self)} points) part {self.part}, building id: {self.building_id}, "
                f"properties set?:{self.properties is not None}")

    def __eq__(self, other):
        if not isinstance(other, OrientedPolygon):
            return False
        return (OrientedGeometry.__eq__(self, other) and np.array_equal(self.points, other.points)
                and self.building_id == other.building_id and self.part == other.part
                and self.properties == other.properties)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None

    def get_polygons_iter(self):
        yield self

    def apply_on_points(self, func, *args, **kwargs):
        return OrientedPolygon(func(self.points, *args, **kwargs))

    def get_parametric(self, force=False):
        """[a, b, c, d] with unit normal from the first three vertices, d = -n·p2 (reference :1747-1765)."""
        if force or self.parametric is None:
            # (a scene computes thousands of these: the cross product on Python floats, which are the same IEEE
            # operations in the same order as on NumPy scalars; norm and dot product stay NumPy's, whose rounding is
            # the BLAS routine's)
            p = self.points
            (x0, y0, z0), (x1, y1, z1), (x2, y2, z2) = p[:3].tolist()
            ax, ay, az = x1 - x0, y1 - y0, z1 - z0
            bx, by, bz = x2 - x1, y2 - y1, z2 - z1
            out = np.empty(4)
            normal = out[:3]
            normal[0] = ay * bz - az * by
            normal[1] = az * bx - ax * bz
            normal[2] = ax * by - ay * bx
            normal /= math.sqrt(normal.dot(normal))
            out[3] = -normal.dot(p[2])
            self.parametric = out
        return self.parametric

    def get_normal(self):
        return self.get_parametric()[:3]

    def get_matrix(self, force=False):
        """Columns [x, y, z]: x along the first side, z the normal, y = z × x (reference :1819-1853)."""
        if not force and self.matrix is not None:
            return self.matrix
        normal = self.get_normal()
        matrix = np.empty((3, 3))
        matrix[0] = self.points[1] - self.points[0]
        matrix[1] = np.cross(normal, matrix[0])
        matrix[2] = normal
        for i in range(2):
            matrix[i] /= np.linalg.norm(matrix[i])
        self.matrix = matrix.T
        return self.matrix

    def distance_to_point(self, point):
        return np.dot(self.get_normal(), (np.asarray(point, dtype=float).reshape(3) - self.points[0]))

    def contains_point(self, point, check_in_plane=False, plane_tol=1e-8):
        """Point-in-polygon in the polygon's own frame, boundary included (reference :1778-1803).
        Runs on the GPU through the library's containment entry point."""
        from . import _lib

        return bool(_lib.polygon_contains_points(self, np.asarray(point, dtype=float).reshape(1, 3),
                                                 check_in_plane, plane_tol)[0])

    def rotate(self, axis, angle_deg):
        """Rotates the polygon about ``axis`` through its bounding-box centre (reference :1900-1920)."""
        from scipy.spatial.transform import Rotation

        center = self.get_centroid()
        matrix = Rotation.from_rotvec(np.radians(angle_deg) * np.asarray(axis, dtype=float)).as_matrix()
        self.points = np.dot(self.points - center, matrix.T) + center
        self.parametric = self.matrix = self.domain = self.centroid = None
        return self

    def contains_points_host(self, points):
        """``contains_point`` without the plane check for many points, evaluated on the host: the points are dropped
        into the polygon's own frame and tested against its ring there, boundary included."""
        matrix = self.get_matrix()
        own = np.asarray(points, dtype=float).reshape(-1, 3) @ matrix
        ring = self.points @ matrix
        return ring_contains(ring[:, :2], own[:, 0], own[:, 1])

    def points_contained(self, polygon):
        """1.0 for every vertex of ``polygon`` that this polygon contains, seen along this polygon's normal
        (reference :1805-1817)."""
        return self.contains_points_host(polygon.points).astype(float)

    def split(self, conflicting_polygons):
        """For each of the polygons in ``conflicting_polygons``: the part of this polygon's footprint (x, y) that none
        of the OTHERS covers, i.e. where polygons tile the plane, the part that lies on it (reference :1922-1940).  An
        entry is an ``OrientedPolygon`` at z = 0, or a list of them when the part is in several pieces.  The others
        must have convex footprints (the grounds of the study scenes are rectangles); the reference subtracts with
        GEOS, whose pieces can start at another vertex."""
        parts = []
        for keep in conflicting_polygons:
            pieces = [self.points[:, :2]]
            for other in conflicting_polygons:
                if other is keep:
                    continue
                pieces = [q for piece in pieces for q in ring_minus_convex(piece, other.points[:, :2])]
            polygons = [OrientedPolygon(np.column_stack([q, np.zeros(len(q))])) for q in pieces]
            parts.append(polygons[0] if len(polygons) == 1 else polygons)
        return parts

    def to_json(self, filename=None):
        data = dict(id=self.id, **self.attributes, geotype="polygon", properties=self.properties,
                    building_id=self.building_id, part=self.part, points=self.points.tolist())
        # Roofs and floors report their plane with the normal turned up / down whatever their first three vertices
        # give (reference :2360-2363), and the reference's file loses that.  The plane is written as one more key
        # (the reference's loader keeps unknown keys as attributes), so that a loaded place is the place that was saved.
        data["parametric"] = self.get_parametric().tolist()
        return _json_save(filename, data) if filename is not None else data

    @classmethod
    def from_json(cls, data=None, filename=None):
        data = _json_data(data, filename)
        if data.pop("geotype") != "polygon":
            raise ValueError("Cannot cast geotype to polygon.")
        part = data.pop("part")
        properties = data.pop("properties")
        building_id = data.pop("building_id")
        parametric = data.pop("parametric", None)  # absent from files the reference wrote
        polygon = cls(np.array(data.pop("points")), **data)
        polygon.part, polygon.properties, polygon.building_id = part, properties, building_id
        if parametric is not None:
            polygon.parametric = np.array(parametric, dtype=float)
        return polygon


class Square(OrientedPolygon):
    """Rectangle with sides along x and y, given by two opposite corners or by centre and side (what the reference's
    ``Square`` builds, geometry.py:1942-2002: same vertices in the same order)."""

    @staticmethod
    def by_2_corner_points(points):
        corners = np.asarray(points, dtype=float)
        if corners.shape != (2, 3):
            raise ValueError("a square takes two corner points of three coordinates")
        assert not np.array_equal(corners[0], corners[1]), "points must be different (P1=P2)"
        assert corners[0, 1] != corners[1, 1] and corners[0, 0] != corners[1, 0], "points must be diagonally opposite"
        low, high = array_utils.sort_by_columns(corners)  # the corner with the smaller x first
        # round the ring from the low corner: along y at the low corner's height, across to the high corner, back
        # along y at the high corner's height
        ring = np.array([low, (low[0], high[1], low[2]), high, (high[0], low[1], high[2])], dtype=float)
        return Square(ring)

    @staticmethod
    def by_center_and_side(center, side, rotate=False):
        """``rotate=True`` turns the square about z by ``random.randint(0, 90)`` degrees, as the reference does
        (seed Python's ``random`` for a reproducible scene)."""
        center = np.asarray(center, dtype=float).reshape(3)
        reach = np.array([side / 2, side / 2, 0.0])
        square = Square.by_2_corner_points(np.stack([center + reach, center - reach]))
        if rotate:
            square.rotate(axis=np.array([0.0, 0.0, 1.0]), angle_deg=random.randint(0, 90))
        return square

    @staticmethod
    def center(points):
        """Mean of the mid-points of the sides 0-1 and 2-3."""
        ring = np.asarray(points, dtype=float)
        return ((ring[0] + ring[1]) / 2 + (ring[2] + ring[3]) / 2) / 2


class OrientedSurface(OrientedGeometry):
    """A set of polygons, e.g. the ground (reference geometry.py:2006-2099)."""

    def __init__(self, polygons, **attributes):
        super().__init__(**attributes)
        self.building_type = attributes.pop("building_type", None)
        if self.building_type is None:
            self.building_type = "road"
        if type(polygons) != list:
            polygons = [polygons]
        if isinstance(polygons[0], np.ndarray):
            self.polygons = [OrientedPolygon(points=p) for p in polygons]
            for polygon in self.polygons:
                polygon.properties = set_properties(self.building_type)
                polygon.part = "ground"
                polygon.building_id = self.id
        elif isinstance(polygons[0], OrientedPolygon):
            self.polygons = polygons
        else:
            raise ValueError("OrientedSurface needs a nump.ndarray or OrientedPolygon as input")

    def __len__(self):
        return len(self.polygons)

    def get_polygons_iter(self):
        return iter(self.polygons)

    def apply_on_points(self, func, *args, **kwargs):
        return OrientedSurface([p.apply_on_points(func, *args, **kwargs) for p in self.polygons])

    def __eq__(self, other):
        """Reference :2042-2054."""
        if not isinstance(other, OrientedSurface) or not OrientedGeometry.__eq__(self, other):
            return False
        if self.building_type != other.building_type or len(self.polygons) != len(other.polygons):
            return False
        return all(a == b for a, b in zip(self.polygons, other.polygons))

    __hash__ = object.__hash__

    def to_json(self, filename=None):
        attributes = {k: v for k, v in self.attributes.items() if k != "building_type"}
        data = dict(id=self.id, **attributes, geotype="surface", building_type=self.building_type,
                    polygons=[p.to_json() for p in self.polygons])
        return _json_save(filename, data) if filename is not None else data

    @classmethod
    def from_json(cls, data=None, filename=None):
        data = _json_data(data, filename)
        if data.pop("geotype") != "surface":
            raise ValueError("Cannot cast geotype to surface.")
        polygons = [OrientedPolygon.from_json(p) for p in data.pop("polygons")]
        return cls(polygons, **data)


class OrientedPolyhedron(OrientedGeometry):
    """Closed set of outward-oriented polygons (reference geometry.py:2122-2268)."""

    def __init__(self, polygons, **attributes):
        super().__init__(**attributes)
        self.aux_surface = OrientedSurface(polygons)
        self.polygons = self.aux_surface.polygons
        self.building_type = attributes.pop("building_type", None)
        for polygon in self.polygons:
            polygon.building_id = self.id

    def __len__(self):
        return len(self.polygons)

    def get_polygons_iter(self):
        return iter(self.polygons)

    def get_top_face(self):
        for polygon in self.polygons:
            if polygon.part == "top":
                return polygon
        assert False, f"could not get top face of {self}"

    def get_bottom_face(self):
        for polygon in self.polygons:
            if polygon.part == "bottom":
                return polygon
        assert False, f"could not get bottom face of {self}"

    def apply_on_points(self, func, *args, **kwargs):
        return OrientedPolyhedron(self.aux_surface.apply_on_points(func, *args, **kwargs).polygons)

    def overlap(self, polyhedron):
        """True when the top faces of the two polyhedra, seen from above, share area; touching along a side or at a
        corner does not count (reference :2239-2248)."""
        a, b = self.get_top_face().points[:, :2], polyhedron.get_top_face().points[:, :2]
        if np.any(a.max(axis=0) <= b.min(axis=0)) or np.any(b.max(axis=0) <= a.min(axis=0)):
            return False
        # sides that cross properly
        for i in range(len(a)):
            p, q = a[i - 1], a[i]
            for j in range(len(b)):
                r, t = b[j - 1], b[j]
                d1 = (q[0] - p[0]) * (r[1] - p[1]) - (q[1] - p[1]) * (r[0] - p[0])
                d2 = (q[0] - p[0]) * (t[1] - p[1]) - (q[1] - p[1]) * (t[0] - p[0])
                d3 = (t[0] - r[0]) * (p[1] - r[1]) - (t[1] - r[1]) * (p[0] - r[0])
                d4 = (t[0] - r[0]) * (q[1] - r[1]) - (t[1] - r[1]) * (q[0] - r[0])
                if d1 * d2 < 0 and d3 * d4 < 0:
                    return True
        # one inside the other (or sharing sides only in part): a point well inside one of them lies inside the other
        def strictly_inside(ring, other):
            pts = np.concatenate([ring, (ring + np.roll(ring, 1, axis=0)) / 2, ring.mean(axis=0, keepdims=True)])
            inside = ring_contains(other, pts[:, 0], pts[:, 1]) & (ring_distance(other, pts[:, 0], pts[:, 1]) > 1e-9)
            return bool(inside.any())

        return strictly_inside(a, b) or strictly_inside(b, a)

    def apply_properties_to_polygons(self):
        assert self.building_type is not None, f"building type for polyhedron {self} is not set"
        for polygon in self.polygons:
            polygon.properties = set_properties(self.building_type)

    def __eq__(self, other):
        """Reference :2151-2164."""
        if not isinstance(other, OrientedPolyhedron) or not OrientedGeometry.__eq__(self, other):
            return False
        if self.building_type != other.building_type or len(self.polygons) != len(other.polygons):
            return False
        return all(a == b for a, b in zip(self.polygons, other.polygons))

    __hash__ = object.__hash__

    def to_json(self, filename=None):
        attributes = {k: v for k, v in self.attributes.items() if k != "building_type"}
        data = dict(id=self.id, **attributes, geotype="polyhedron", building_type=self.building_type,
                    polygons=[p.to_json() for p in self.polygons])
        return _json_save(filename, data) if filename is not None else data

    @classmethod
    def from_json(cls, data=None, filename=None):
        data = _json_data(data, filename)
        if data.pop("geotype") != "polyhedron":
            raise ValueError("Cannot cast geotype to polyhedron.")
        polygons = [OrientedPolygon.from_json(p) for p in data.pop("polygons")]
        building_type = data.pop("building_type", None)
        out = cls(polygons, **data)
        out.building_type = building_type
        return out


class Building(OrientedPolyhedron):
    """Footprint extruded along z (reference geometry.py:2305-2415)."""

    @staticmethod
    def by_polygon_and_height(polygon, height, id=None, make_ccw=True, keep_ground=True, flat_roof=True):
        if isinstance(polygon, OrientedPolygon):
            bottom_points = polygon.points
        elif isinstance(polygon, np.ndarray):
            bottom_points = polygon
        else:
            raise ValueError(f"Type {type(polygon)} is not supported for polygon !")
        bottom_points = np.asarray(bottom_points, dtype=float)
        top_points = np.array(bottom_points)
        if flat_roof:
            top_points[:, -1] = float(np.max(bottom_points[:, 2]))
        top_points[:, -1] += float(height)

        if make_ccw:
            if not is_ccw(top_points):
                top_points = top_points[::-1, :]
            else:
                bottom_points = bottom_points[::-1, :]

        top = OrientedPolygon(top_points)
        top.part = "top"
        bottom = OrientedPolygon(bottom_points)
        bottom.part = "bottom"
        if top.get_normal()[2] < 0:
            top.parametric = -top.parametric
        if bottom.get_normal()[2] > 0:
            bottom.parametric = -bottom.parametric

        polygons = [top, bottom] if keep_ground else [top]
        walls_bottom = bottom_points[::-1, :]  # same xy order as the top ring
        n = top_points.shape[0]
        for i in range(n):
            a, b = top_points[i - 1], top_points[i]
            c, d = walls_bottom[i - 1], walls_bottom[i]
            polygons.append(OrientedPolygon(np.array([a, c, d, b])))
        return OrientedPolyhedron(polygons, id=id)

    @staticmethod
    def by_polygon2d_and_height(polygon, height, id=None, make_ccw=True, keep_ground=False, flat_roof=True):
        points_2d = np.asarray(polygon, dtype=float)
        if points_2d.ndim != 2 or points_2d.shape[1] != 2:
            raise ValueError(f"Type {type(polygon)} is not supported for polygon !")
        if np.array_equal(points_2d[0], points_2d[-1]):
            points_2d = points_2d[:-1]  # GeoJSON rings repeat the first vertex
        points_3d = np.hstack((points_2d, np.zeros((len(points_2d), 1))))
        return Building.by_polygon_and_height(OrientedPolygon(points_3d), height, id, make_ccw=make_ccw,
                                              keep_ground=keep_ground, flat_roof=flat_roof)

    @staticmethod
    def building_on_slope(polygon, ground, height, id=None):
        """Building whose footprint vertices are dropped vertically onto the plane of ``ground`` and whose flat roof
        stands ``height`` above the highest of them (reference :2418-2435)."""
        feet = polygon.points.copy()
        for i, vertex in enumerate(polygon.points):
            below = vertex.copy()
            below[-1] = -1000
            feet[i] = polygon_line_intersection(ground, np.array([below, vertex]))
        return Building.by_polygon_and_height(OrientedPolygon(feet), height, id, make_ccw=True, keep_ground=True,
                                              flat_roof=True)

    @staticmethod
    def rebuild_building(polygon, grounds, height):
        """The building(s) of footprint ``polygon`` standing on ``grounds``: one building when a single ground holds
        every vertex, otherwise one per part of the footprint on each ground that holds some (reference :2438-2471)."""
        if not isinstance(grounds, list):
            return [Building.building_on_slope(polygon, grounds, height)]
        sharing = []
        for ground in grounds:
            held = ground.points_contained(polygon)
            if held.all():
                return [Building.building_on_slope(polygon, ground, height)]
            if held.any():
                sharing.append(ground)
        buildings = []
        for ground, part in zip(sharing, polygon.split(sharing)):
            for piece in (part if isinstance(part, list) else [part]):
                buildings.append(Building.building_on_slope(piece, ground, height))
        return buildings

    @staticmethod
    def create_tree_trunk(trunk_spot, trunk_size, trunk_height, rotate, crown_size=None, crown_height=None):
        """Square column of type "tree_trunk" (reference :2475-2486); the crown it will carry is noted in its
        attributes.  ``trunk_height=None`` draws it with ``random.uniform`` between the tree height bounds."""
        if trunk_height is None:
            trunk_height = random.uniform(MIN_TREE_HEIGHT, MAX_TREE_HEIGHT)
        trunk_top = Square.by_center_and_side(trunk_spot, trunk_size, rotate=rotate).get_points()
        trunk = Building.by_polygon_and_height(trunk_top, trunk_height, keep_ground=False)
        trunk.building_type = "tree_trunk"
        trunk.apply_properties_to_polygons()
        if crown_size is not None and crown_height is not None:
            trunk.attributes["crown_size"] = crown_size
            trunk.attributes["crown_height"] = crown_height
        return trunk

    @staticmethod
    def create_tree_crown(tree_trunk, crown_size, crown_height, rotate=False):
        """Box of type "tree_crown" standing on the trunk's top face (reference :2488-2496)."""
        center = Square.center(tree_trunk.get_top_face().points)
        crown_footprint = Square.by_center_and_side(center, crown_size, rotate=rotate)
        crown = Building.by_polygon_and_height(crown_footprint, crown_height, keep_ground=True)
        crown.building_type = "tree_crown"
        crown.apply_properties_to_polygons()
        return crown


class Cube(OrientedPolyhedron):
    """Axis-aligned cube around a point; its 6 faces are the emitter's screens (reference :2502-2531)."""

    @staticmethod
    def by_point_and_side_length(point, side_length):
        r = side_length / 2
        polygon = np.array([[r, r, -r], [-r, r, -r], [-r, -r, -r], [r, -r, -r]], dtype=float)
        polygon += np.asarray(point, dtype=float).reshape(1, 3)
        building = Building.by_polygon_and_height(polygon, side_length)
        return Cube(building.polygons)


class OrientedPlace(OrientedGeometry):
    """Ground surface + polyhedra + receiver points (reference geometry.py:2534-2663)."""

    def __init__(self, surface, polyhedra=None, set_of_points=None, **attributes):
        super().__init__(**attributes)
        if isinstance(surface, OrientedSurface):
            self.surface = surface
        elif type(surface) == list or type(surface) == np.ndarray:
            self.surface = OrientedSurface(surface)
        else:
            raise ValueError("OrientedPlace needs an OrientedSurface or a numpy.ndarray as surface input")
        if polyhedra is not None:
            if type(polyhedra) != list:
                polyhedra = [polyhedra]
            if len(polyhedra) == 0 or isinstance(polyhedra[0], OrientedPolyhedron):
                self.polyhedra = polyhedra
            else:
                self.polyhedra = [OrientedPolyhedron(p) for p in polyhedra]
        else:
            self.polyhedra = []
        if set_of_points is None:
            set_of_points = np.empty((0, 3))
        if type(set_of_points) == np.ndarray and set_of_points.ndim == 2 and set_of_points.shape[1] == 3:
            self.set_of_points = set_of_points
        else:
            raise ValueError(f"OrientedPlace has an invalid set_of_points as input: {set_of_points}, "
                             f"type {type(set_of_points)}")
        self._scene_arrays = None

    def __str__(self):
        return f"Place({len(self.polyhedra)} polyhedra, {self.set_of_points.shape[0]} points)"

    def add_set_of_points(self, points):
        self.set_of_points = np.concatenate([self.set_of_points, np.asarray(points, dtype=float).reshape(-1, 3)])

    def get_polygons_iter(self):
        """Ground polygons first, then each polyhedron's polygons in its own order (reference :2650-2656)."""
        return itertools.chain(self.surface.get_polygons_iter(),
                               itertools.chain.from_iterable(p.get_polygons_iter() for p in self.polyhedra))

    def apply_on_points(self, func, *args, **kwargs):
        surface = self.surface.apply_on_points(func, *args, **kwargs)
        polyhedra = [p.apply_on_points(func, *args, **kwargs) for p in self.polyhedra]
        return OrientedPlace(surface, polyhedra=polyhedra, set_of_points=func(self.set_of_points, *args, **kwargs))

    def obstructs_line_path(self, points):
        return polygons_obstruct_line_path(self.get_polygons_list(), points)

    def overlap_place(self, polyhedron):
        """True when the polyhedron's top face, seen from above, shares area with the top face of a polyhedron of the
        place (reference :2731-2740)."""
        return any(mine.overlap(polyhedron) for mine in self.polyhedra)

    def add_polyhedron(self, polyhedron, allow_overlap=False):
        """Appends a polyhedron; unless ``allow_overlap``, one that overlaps a polyhedron of the place is refused
        (reference :2742-2748)."""
        assert isinstance(polyhedron, OrientedPolyhedron), "given polyhedron is not of type OrientedPolyhedron"
        if not allow_overlap:
            assert not self.overlap_place(polyhedron), "the polyhedron overlaps another one in the place"
        self.polyhedra.append(polyhedron)
        self._scene_arrays = None

    def add_tree(self, tree_size, how_close_to_buildings=2):
        """Plants a tree trunk (random height, randomly turned) at a random spot of the first ground polygon that is
        at least ``how_close_to_buildings + tree_size / 2`` away from every polyhedron's top face, seen from above
        (reference :2753-2787).  Draws spots with ``random.uniform`` over the ground's box until one fits, as the
        reference does over the box of the free zone; the reference grows the footprints with mitred corners, here
        the distance is the plain one, so a spot diagonal to a corner can be accepted that the reference refuses.
        Returns the spot."""
        margin = how_close_to_buildings + tree_size / 2
        ground = self.surface.polygons[0].points[:, :2]
        lo, hi = ground.min(axis=0).tolist(), ground.max(axis=0).tolist()
        tops = [p.get_top_face().points[:, :2] for p in self.polyhedra]
        for _ in range(100000):
            x, y = random.uniform(lo[0], hi[0]), random.uniform(lo[1], hi[1])
            if not ring_contains(ground, x, y)[0]:
                continue
            if any(ring_contains(t, x, y)[0] or ring_distance(t, x, y)[0] <= margin for t in tops):
                continue
            spot = np.array([x, y, 0.0])
            self.add_polyhedron(Building.create_tree_trunk(spot, tree_size, trunk_height=None, rotate=True))
            return spot
        raise RuntimeError("no room left for a tree")

    def add_tree_crowns(self):
        """Puts a crown on every tree trunk that carries crown attributes (reference :2789-2802).  Called after the
        place has been solved, so that rays pass through the crowns and are only attenuated by them.  Returns the
        number of polygons added (they come last in the polygon order)."""
        crowns = []
        for polyhedron in self.polyhedra:
            if polyhedron.building_type == "tree_trunk":
                crowns.append(Building.create_tree_crown(polyhedron, crown_size=polyhedron.attributes["crown_size"],
                                                         crown_height=polyhedron.attributes["crown_height"],
                                                         rotate=False))
        self.polyhedra.extend(crowns)
        if crowns:
            self._scene_arrays = None
        return sum(len(c.polygons) for c in crowns)

    def __eq__(self, other):
        """Reference :2577-2593."""
        if not isinstance(other, OrientedPlace) or not OrientedGeometry.__eq__(self, other):
            return False
        if not np.array_equal(self.set_of_points, other.set_of_points) or self.surface != other.surface:
            return False
        return all(a == b for a, b in zip(self.polyhedra, other.polyhedra))

    __hash__ = object.__hash__

    def to_json(self, filename=None):
        data = dict(id=self.id, **self.attributes, geotype="place", surface=self.surface.to_json(),
                    polyhedra=[p.to_json() for p in self.polyhedra], set_of_points=self.set_of_points.tolist())
        if filename is not None:
            with open(filename, "w") as f:
                json.dump(data, f)
        return data

    @classmethod
    def from_json(cls, data=None, filename=None):
        if data is None:
            with open(filename) as f:
                data = json.load(f)
        data = dict(data)
        if data.pop("geotype") != "place":
            raise ValueError("Cannot cast geotype to place.")
        surface = OrientedSurface.from_json(data.pop("surface"))
        polyhedra = [OrientedPolyhedron.from_json(p) for p in data.pop("polyhedra")]
        points = np.array(data.pop("set_of_points"), dtype=float).reshape(-1, 3)
        return cls(surface, polyhedra, points, **data)


# --------------------------------------------------------------------------------------------------
# flat (structure-of-arrays) view of a polygon list: what the HIP library consumes
# --------------------------------------------------------------------------------------------------

class SceneArrays:
    """Contiguous arrays describing N polygons.

    vert_off  int32  [N+1]     CSR offsets into ``verts``
    verts     float64[Vtot,3]  vertices, polygon after polygon, in each polygon's own order
    planes    float64[N,4]     [nx, ny, nz, d] exactly as ``get_parametric`` returns them (tops/bottoms may carry
                               a flipped sign, see ``Building.by_polygon_and_height``)
    material  int32  [N]       index into ``materials_properties.MATERIAL_NAMES``
    building  int32  [N]       dense id of the polyhedron / surface the polygon belongs to
    """

    def __init__(self, polygons):
        polygons = list(polygons)
        self.n = len(polygons)
        counts = np.array([len(p) for p in polygons], dtype=np.int64)
        self.vert_off = np.zeros(self.n + 1, dtype=np.int32)
        np.cumsum(counts, out=self.vert_off[1:])
        self.verts = np.ascontiguousarray(np.concatenate([p.points for p in polygons]), dtype=np.float64)
        self.planes = np.ascontiguousarray(np.array([p.get_parametric() for p in polygons]), dtype=np.float64)
        self.material = np.zeros(self.n, dtype=np.int32)
        self.building = np.zeros(self.n, dtype=np.int32)
        building_index = {}
        for i, p in enumerate(polygons):
            material = "concrete"
            if p.properties is not None:
                material = p.properties.get("material", "concrete")
            self.material[i] = MATERIAL_ID[material]
            self.building[i] = building_index.setdefault(p.building_id, len(building_index))
        self._device_scene = None

    def device_scene(self):
        """The library-side scene handle for these polygons (created on first use, then reused)."""
        from . import _lib

        if self._device_scene is None:
            self._device_scene = _lib.Scene(self)
        return self._device_scene


class LoadedScene:
    """A rooftop file as the kernels need it — ``arrays`` (SceneArrays), the buildings' ``footprints`` (list of (n, 2)
    rings) and the scene's box — without the thousands of Python objects of a place.  ``place`` builds those when asked
    for (``load_place``).  Stands in for a place where only footprints and the box are read (``get_domain``)."""

    def __init__(self, filename, seed, arrays, footprints, domain):
        self.filename, self.seed = filename, seed
        self.arrays, self.footprints, self.domain = arrays, footprints, domain
        self._place = None

    def get_domain(self):
        return self.domain[0].copy(), self.domain[1].copy()

    @property
    def place(self):
        if self._place is None:
            self._place = load_place(self.filename, seed=self.seed)
        return self._place


def _scene_cache_dir():
    """Where loaded scenes are kept: ``data/.scene_cache`` beside the package when it can be written (the files then
    travel with the tree, like built libraries), else the user's cache directory."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for d in (os.path.join(root, "data", ".scene_cache"), os.path.join(os.path.expanduser("~"), ".cache", "raytracing_amd")):
        try:
            os.makedirs(d, exist_ok=True)
            if os.access(d, os.W_OK):
                return d
        except OSError:
            pass
    return None


def _loader_fingerprint():
    """Changes whenever the code that turns a file into arrays does (this module and the material table)."""
    global _LOADER_FINGERPRINT
    if _LOADER_FINGERPRINT is None:
        import hashlib

        h = hashlib.sha1()
        here = os.path.dirname(os.path.abspath(__file__))
        for name in ("geometry.py", "materials_properties.py", "array_utils.py"):
            with open(os.path.join(here, name), "rb") as f:
                h.update(f.read())
        h.update(np.__version__.encode())
        _LOADER_FINGERPRINT = h.hexdigest()
    return _LOADER_FINGERPRINT


_LOADER_FINGERPRINT = None


def load_scene(filename, seed=0, use_cache=True):
    """``load_place`` for callers that need the arrays, not the objects: the SceneArrays of the file's polygon list, the
    footprints and the box, read from a cache file when this very file (by content), seed and loader code have been
    loaded before, and made through ``load_place`` (and cached) otherwise.  The cached arrays are the arrays
    ``SceneArrays(load_place(filename, seed).get_polygons_list())`` gives, bit for bit: they are those arrays, stored."""
    import hashlib

    path = None
    if use_cache:
        cache_dir = _scene_cache_dir()
        if cache_dir is not None:
            with open(filename, "rb") as f:
                digest = hashlib.sha1(f.read() + f"|{seed}|{_loader_fingerprint()}".encode()).hexdigest()
            path = os.path.join(cache_dir, f"{os.path.basename(filename)}-{digest[:20]}.npz")
            if os.path.exists(path):
                try:
                    with np.load(path) as z:
                        arrays = SceneArrays.__new__(SceneArrays)
                        arrays.vert_off, arrays.verts, arrays.planes = z["vert_off"], z["verts"], z["planes"]
                        arrays.material, arrays.building = z["material"], z["building"]
                        arrays.n = len(arrays.material)
                        arrays._device_scene = None
                        off, xy, domain = z["footprint_off"], z["footprint_xy"], z["domain"]
                    footprints = [xy[off[i]:off[i + 1]] for i in range(len(off) - 1)]
                    return LoadedScene(filename, seed, arrays, footprints, (domain[0], domain[1]))
                except (OSError, ValueError, KeyError):
                    pass  # an unreadable cache file is made again
    place = load_place(filename, seed=seed)
    arrays = scene_arrays_of(place.get_polygons_list())
    footprints = [p.get_top_face().points[:, :2].copy() for p in place.polyhedra]
    lo, hi = place.get_domain()
    scene = LoadedScene(filename, seed, arrays, footprints, (np.asarray(lo, float), np.asarray(hi, float)))
    scene._place = place
    if path is not None:
        off = np.zeros(len(footprints) + 1, dtype=np.int64)
        np.cumsum([len(f) for f in footprints], out=off[1:])
        tmp = f"{path}.{os.getpid()}.tmp.npz"
        try:
            np.savez(tmp, vert_off=arrays.vert_off, verts=arrays.verts, planes=arrays.planes, material=arrays.material,
                     building=arrays.building, footprint_off=off,
                     footprint_xy=np.concatenate(footprints) if footprints else np.zeros((0, 2)),
                     domain=np.stack([scene.domain[0], scene.domain[1]]))
            os.replace(tmp, path)  # (several ranks may load the same scene at once: each writes its own file, then renames)
        except OSError:
            pass
    return scene


def scene_arrays_of(polygons):
    """SceneArrays for a polygon list, cached on the list's first polygon so repeated calls reuse the upload."""
    polygons = list(polygons)
    key = tuple(id(p) for p in polygons)
    cache = getattr(polygons[0], "_scene_cache", None)
    if cache is not None and cache[0] == key:
        return cache[1]
    arrays = SceneArrays(polygons)
    polygons[0]._scene_cache = (key, arrays)
    return arrays


# --------------------------------------------------------------------------------------------------
# geometry entry points (GPU)
# --------------------------------------------------------------------------------------------------

def polygons_obstruct_line_path(polygons, points, return_polygons=False):
    """True when any polygon cuts the open segment ``points[0] -> points[1]`` (reference geometry.py:836-898)."""
    from . import _lib

    points = np.asarray(points, dtype=float)
    assert len(points) == 2, "wrong argument"
    polygons = list(polygons)
    if len(polygons) == 0:
        return [] if return_polygons else False
    scene = scene_arrays_of(polygons).device_scene()
    if return_polygons:
        hits = np.flatnonzero(scene.segment_hits(points.reshape(1, 2, 3))[0])
        groups, current = [], []
        for i in hits:  # grouping rule of the reference (:886-895)
            if not current or current[0].building_id == polygons[i].building_id:
                current.append(polygons[i])
            else:
                groups.append(current)
                current = []
        if current:
            groups.append(current)
        return groups
    return bool(scene.obstructed(points.reshape(1, 2, 3))[0])


def polygon_visibility_vector(polygon_A, polygons, out=None, strict=False):
    """Which polygons a ray leaving the front of ``polygon_A`` may hit (reference geometry.py:924-1019).
    ``out`` is OR-ed into when given; a fresh vector starts at all-False (the reference leaves it uninitialised)."""
    polygons = list(polygons)
    scene = scene_arrays_of(polygons).device_scene()
    self_index = -1
    for i, p in enumerate(polygons):
        if p is polygon_A:
            self_index = i
            break
    vis = scene.face_visibility(polygon_A.points, polygon_A.get_parametric(), self_index, strict)
    if out is None:
        return vis
    out |= vis
    return out


def polygons_visibility_matrix(polygons, strict=False):
    """Symmetric N×N visibility matrix (reference geometry.py:1022-1048)."""
    polygons = list(polygons)
    return scene_arrays_of(polygons).device_scene().visibility_matrix(strict)


def polygons_sharp_edges_iter(polygons, min_angle=10.0):
    """Yields ``(edge (2,3), i, j)`` for every convex shared edge, pairs in ascending (i, j) order
    (reference geometry.py:482-565)."""
    polygons = list(polygons)
    pairs, edges = scene_arrays_of(polygons).device_scene().sharp_edges(min_angle)
    for (i, j), edge in zip(pairs, edges):
        yield edge, int(i), int(j)


def sharp_edges_visibility_dict(sharp_edges):
    d = container_utils.ManyToOneDict()
    for edge, i, j in sharp_edges:
        d[i, j] = edge
    return d


class SolveResult:
    """What the three solver entry points return beside the points: the fields of SciPy's result object that the
    reference's callers read (``success``, ``x``; ``nfev`` when the reference's own solver was asked for)."""

    def __init__(self, x, success, nfev=None):
        self.x = x
        self.success = bool(success)
        self.status = 1 if success else 0
        self.message = "The solution converged." if success else "No specular solution exists."
        if nfev is not None:
            self.nfev = int(nfev) + 2  # SciPy's count: the method's evaluations and the two of its wrapper's shape check
            self.message = "The solution converged." if success else "The solver did not report convergence."


def _reference_solver_asked(kwargs):
    """``reference_solver=True`` (or ``ray_tracing.REFERENCE_SOLVER``) makes the three solver functions answer as the
    reference's own solver does — SciPy's hybr from the reference's first guess, with its misses — instead of with the
    exact root."""
    asked = kwargs.pop("reference_solver", None)
    if asked is None:
        from . import ray_tracing

        asked = ray_tracing.REFERENCE_SOLVER
    return bool(asked)


def _solve(origin, destination, planes, edge, reference):
    from . import _lib

    origin = np.asarray(origin, dtype=float).reshape(3)
    destination = np.asarray(destination, dtype=float).reshape(1, 3)
    if reference:
        x, ok, nfev = _lib.solve_like_reference(origin, destination, planes, edge)
        # sol.x is the solver's vector of unknowns: the reflection points, then the diffraction unknown in the edge's
        # frame; the points are what the reference's functions return beside it
        return x[0], SolveResult(x[0].reshape(-1), ok[0], nfev[0])
    if edge is None:
        x, ok = _lib.solve_reflections(origin, destination, planes)
    else:
        x, ok = _lib.solve_reflections_and_diffraction(origin, destination, planes, edge)
    return x[0], SolveResult(x[0].reshape(-1), ok[0])


def reflexion_points_from_origin_destination_and_planes(origin, destination, planes_parametric, **kwargs):
    """Reflection point on each plane (reference geometry.py:660-687).  ``success`` is False when no specular path
    exists over the infinite planes (an image segment misses its plane); with ``reference_solver=True`` it is what
    SciPy's hybr reports from the reference's first guess, and the points are that solver's."""
    planes = np.asarray(planes_parametric, dtype=float).reshape(-1, 4)
    return _solve(origin, destination, planes, None, _reference_solver_asked(kwargs))


def diffraction_point_from_origin_destination_and_edge(origin, destination, edge, **kwargs):
    """Keller-cone point on the edge's line (reference geometry.py:718-748)."""
    return _solve(origin, destination, np.empty((0, 4)), np.asarray(edge, dtype=float).reshape(2, 3),
                  _reference_solver_asked(kwargs))


def reflexion_points_and_diffraction_point_from_origin_destination_planes_and_edge(origin, destination,
                                                                                   planes_parametric, edge, **kwargs):
    """Reflections then one diffraction (reference geometry.py:800-833), including its rule that the reflection
    points are aimed at the edge's first endpoint."""
    if len(planes_parametric) == 0:
        return diffraction_point_from_origin_destination_and_edge(origin, destination, edge, **kwargs)
    if edge is None:
        return reflexion_points_from_origin_destination_and_planes(origin, destination, planes_parametric, **kwargs)
    planes = np.asarray(planes_parametric, dtype=float).reshape(-1, 4)
    return _solve(origin, destination, planes, np.asarray(edge, dtype=float).reshape(2, 3), _reference_solver_asked(kwargs))


def point_on_edge_(point, edge, tol=1e-8):
    """Point within a ±tol box of the edge in the edge-aligned frame (reference geometry.py:456-479)."""
    edge = np.asarray(edge, dtype=float)
    matrix = projection_matrix_from_vector(edge[1] - edge[0]).T
    pe = edge @ matrix.T
    pp = np.asarray(point, dtype=float).reshape(3) @ matrix.T
    a, b = pe[0], pe[1]
    return bool(a[0] - tol <= pp[0] <= a[0] + tol and a[1] - tol <= pp[1] <= a[1] + tol
                and a[2] - tol <= pp[2] <= b[2] + tol)


# --------------------------------------------------------------------------------------------------
# GeoJSON -> place
# --------------------------------------------------------------------------------------------------

_GRS80_A = 6378137.0
_GRS80_F = 1.0 / 298.257222101


def lonlat_to_epsg3035(lon_deg, lat_deg):
    """Forward ETRS89-LAEA Europe (EPSG:3035): Lambert azimuthal equal-area on GRS80, origin 52°N 10°E,
    false easting 4 321 000 m, false northing 3 210 000 m (EPSG guidance note 7-2, method 9820).
    The reference gets this from PROJ through ``GeoDataFrame.to_crs(epsg=3035)`` (geometry.py:2916)."""
    lon = np.radians(np.asarray(lon_deg, dtype=float))
    lat = np.radians(np.asarray(lat_deg, dtype=float))
    e2 = 2 * _GRS80_F - _GRS80_F ** 2
    e = math.sqrt(e2)
    lat0, lon0 = math.radians(52.0), math.radians(10.0)

    def q(phi):
        s = np.sin(phi)
        return (1 - e2) * (s / (1 - e2 * s * s) - (1 / (2 * e)) * np.log((1 - e * s) / (1 + e * s)))

    qp = q(math.pi / 2)
    q0 = q(lat0)
    beta = np.arcsin(q(lat) / qp)
    beta0 = math.asin(q0 / qp)
    rq = _GRS80_A * math.sqrt(qp / 2)
    d = _GRS80_A * (math.cos(lat0) / math.sqrt(1 - e2 * math.sin(lat0) ** 2)) / (rq * math.cos(beta0))
    b = rq * np.sqrt(2 / (1 + math.sin(beta0) * np.sin(beta) + math.cos(beta0) * np.cos(beta) * np.cos(lon - lon0)))
    east = 4321000.0 + b * d * np.cos(beta) * np.sin(lon - lon0)
    north = 3210000.0 + (b / d) * (math.cos(beta0) * np.sin(beta) - math.sin(beta0) * np.cos(beta) * np.cos(lon - lon0))
    return east, north


def read_geojson_tolerant(path):
    """Parse a GeoJSON file, accepting the trailing commas GDAL accepts (``place_levant*.geojson`` have them)."""
    with open(path, encoding="utf-8") as f:
        text = f.read()
    try:
        return json.loads(text)
    except json.JSONDecodeError:
        return json.loads(re.sub(r",(\s*[}\]])", r"\1", text))


def preprocess_geojson(filename, drop_missing_heights=False, seed=0, project=True):
    """Footprints of a rooftop file as a list of dicts ``{ring (n,2) metres, height, building_type, id}``.

    Steps of the reference's ``preprocess_geojson`` (geometry.py:2867-2920): keep Polygon features only, fill
    missing heights with uniform(8, 15) rounded to 0.1 m, default building type "appartments", one id per
    building, project lon/lat to EPSG:3035.  Differences, all for determinism: the random heights come from
    ``numpy.random.default_rng(seed)`` (one draw per feature, used where the file has none), ids are derived from
    the seed, and the result stays in memory instead of being written next to the input.
    """
    data = read_geojson_tolerant(filename)
    features = [f for f in data["features"] if f.get("geometry") and f["geometry"].get("type") == "Polygon"]
    rng = np.random.default_rng(seed)
    random_heights = np.round(rng.uniform(8, 15, size=len(features)), 1)
    id_rng = np.random.default_rng([seed, 1])
    out = []
    for k, feature in enumerate(features):
        props = feature.get("properties") or {}
        height = props.get("height", None)
        try:
            height = float(height) if height is not None else float("nan")
        except (TypeError, ValueError):
            height = float("nan")
        if math.isnan(height):
            if drop_missing_heights:
                continue
            height = float(random_heights[k])
        ring = np.asarray(feature["geometry"]["coordinates"][0], dtype=float)[:, :2]  # exterior only (:2404)
        if project:
            x, y = lonlat_to_epsg3035(ring[:, 0], ring[:, 1])
            ring = np.column_stack([x, y])
        building_id = str(uuid.UUID(int=int(id_rng.integers(0, 2 ** 63)) << 64 | int(id_rng.integers(0, 2 ** 63))))
        out.append(dict(ring=ring, height=height, building_type=props.get("building_type", "appartments"),
                        id=building_id))
    return out


def generate_place_from_rooftops_file(roof_top_file, center=True, drop_missing_heights=True, default_height=10,
                                      seed=0):
    """Place made of one flat ground quad over the footprints' bounding box and one extruded building per
    footprint (reference geometry.py:2824-2863).  Accepts either a path or the list ``preprocess_geojson``
    returns."""
    footprints = roof_top_file
    if isinstance(roof_top_file, (str, bytes)) or hasattr(roof_top_file, "__fspath__"):
        footprints = preprocess_geojson(roof_top_file, seed=seed)
    all_xy = np.concatenate([f["ring"] for f in footprints])
    lo, hi = all_xy.min(axis=0), all_xy.max(axis=0)
    shift = (lo + hi) / 2 if center else np.zeros(2)
    lo, hi = lo - shift, hi - shift

    polyhedra = []
    for f in footprints:
        building = Building.by_polygon2d_and_height(f["ring"] - shift, f["height"], id=f["id"], keep_ground=False,
                                                    flat_roof=True)
        building.building_type = f["building_type"]
        building.apply_properties_to_polygons()
        polyhedra.append(building)

    ground = np.zeros((4, 3))  # corner order of the reference's get_bounds (:2815-2822)
    ground[0::3, 0] = lo[0]
    ground[1:3, 0] = hi[0]
    ground[:2, 1] = lo[1]
    ground[2:, 1] = hi[1]
    place = OrientedPlace(OrientedSurface(ground, building_type="ground"))
    place.polyhedra = polyhedra
    return place


def load_place(filename, seed=0):
    """``preprocess_geojson`` + ``generate_place_from_rooftops_file`` in one deterministic call."""
    return generate_place_from_rooftops_file(preprocess_geojson(filename, seed=seed))
