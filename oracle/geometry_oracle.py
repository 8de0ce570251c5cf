"""CPU ORACLE — test infrastructure only.  Never imported by the product path.

Plain NumPy/SciPy restatement of the geometry half of the reference's path-solve path
(``raytracing/geometry.py`` of MaximeLeu/RayTracing), one function per reference function, each citing the
lines it follows.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.

What replaces third-party natives the reference calls:
  * ``scipy.optimize.root`` (MINPACK hybr) — called as the reference calls it (SciPy is present here);
  * Shapely/GEOS ``Polygon.intersects(Point)`` — ``pip_inclusive`` below (crossing number, boundary counts
    as inside);
  * Shapely/GEOS ``is_valid`` / ``difference`` / ``is_empty`` / ``union`` in the painter's algorithm —
    ``oracle.visibility_oracle`` (exact slab decomposition).
Pinning: solver, plane, frame, extrusion and sharp-edge functions are checked against vectors produced by
the reference's own code (``tests/golden/make_golden.py``); the GEOS-backed predicates have no reference
fixture: parity unpinned for those.
"""
import itertools

import numpy as np
from numpy.linalg import norm
from scipy.optimize import root


# ---------------------------------------------------------------------------------------------
# polygon attributes — geometry.py:1437-1465, 1747-1853
# ---------------------------------------------------------------------------------------------

def parametric(points):
    """geometry.py:1757-1764"""
    normal = np.cross(points[1, :] - points[0, :], points[2, :] - points[1, :])
    normal = normal / norm(normal)
    a, b, c = normal
    d = -np.dot(np.array([a, b, c]), points[2, :])
    return np.array([a, b, c, d])


def frame_matrix(points, normal):
    """geometry.py:1839-1851; returned as the reference stores it (columns are the axes)."""
    matrix = np.empty((3, 3), dtype=float)
    matrix[0, :] = points[1, :] - points[0, :]
    matrix[1, :] = np.cross(normal, matrix[0, :])
    matrix[2, :] = normal
    for i in range(2):
        matrix[i, :] /= norm([matrix[i, :]])
    return matrix.T


def domain(points):
    """geometry.py:1446-1450"""
    return np.array([points.min(axis=0), points.max(axis=0)])


def centroid(points):
    """geometry.py:1462-1463 (bounding-box centre)"""
    return domain(points).mean(axis=0)


def enclosed_area(points):
    """geometry.py:163-167"""
    x = points[:, 0]
    y = points[:, 1]
    s = np.sum((x[1:] - x[:-1]) * (y[1:] + y[:-1]))
    s += (x[0] - x[-1]) * (y[0] + y[-1])
    return s / 2


def is_ccw(points):
    """geometry.py:181"""
    return enclosed_area(points) < 0


class Poly:
    """A polygon as the oracle sees it: vertices, plane (possibly sign-fixed) and labels."""

    __slots__ = ("points", "parametric", "part", "building", "material")

    def __init__(self, points, part="side", building=0, material="concrete"):
        self.points = np.asarray(points, dtype=float)
        self.parametric = parametric(self.points)
        self.part = part
        self.building = building
        self.material = material

    @property
    def normal(self):
        return self.parametric[:3]

    def matrix(self):
        return frame_matrix(self.points, self.normal)


def building_by_polygon_and_height(bottom_points, height, keep_ground=True, flat_roof=True, building=0,
                                   material="concrete"):
    """geometry.py:2313-2385: [top, (bottom), walls...] with walls [A, C, D, B]."""
    bottom_points = np.asarray(bottom_points, dtype=float)
    top_points = np.array(bottom_points)
    if flat_roof:
        top_points[:, -1] = float(np.max(bottom_points[:, 2]))
        top_points[:, -1] += float(height)
    else:
        top_points[:, -1] += float(height)
    if not is_ccw(top_points):
        top_points = top_points[::-1, :]
    else:
        bottom_points = bottom_points[::-1, :]
    top = Poly(top_points, "top", building, material)
    bottom = Poly(bottom_points, "bottom", building, material)
    if top.normal[2] < 0:
        top.parametric = -top.parametric
    if bottom.normal[2] > 0:
        bottom.parametric = -bottom.parametric
    polys = [top, bottom] if keep_ground else [top]
    bottom_points = bottom_points[::-1, :]
    n = top_points.shape[0]
    for i in range(n):
        A = top_points[i - 1, :]
        B = top_points[i, :]
        C = bottom_points[i - 1, :]
        D = bottom_points[i, :]
        polys.append(Poly(np.vstack([A, C, D, B]), "side", building, material))
    return polys


def building_by_polygon2d_and_height(points_2d, height, building=0, material="concrete"):
    """geometry.py:2388-2415 with keep_ground=False as generate_place_from_rooftops_file calls it (:2846)."""
    points_2d = np.asarray(points_2d, dtype=float)
    if np.array_equal(points_2d[0], points_2d[-1]):
        points_2d = points_2d[:-1]
    points_3d = np.hstack((points_2d, np.zeros((len(points_2d), 1))))
    return building_by_polygon_and_height(points_3d, height, keep_ground=False, building=building,
                                          material=material)


def cube_by_point_and_side_length(point, side_length):
    """geometry.py:2508-2531"""
    r = side_length / 2
    polygon = np.array([[r, r, -r], [-r, r, -r], [-r, -r, -r], [r, -r, -r]], dtype=float)
    polygon += np.asarray(point, dtype=float).reshape(1, 3)
    return building_by_polygon_and_height(polygon, side_length)


def place_polygons(footprints):
    """Ground quad + buildings in file order: geometry.py:2815-2822, 2845-2853, 2650-2656.
    ``footprints`` = list of dicts with ``ring`` (n,2) already centred, ``height``."""
    all_xy = np.concatenate([f["ring"] for f in footprints])
    lo, hi = all_xy.min(axis=0), all_xy.max(axis=0)
    ground = np.zeros((4, 3))
    ground[0::3, 0] = lo[0]
    ground[1:3, 0] = hi[0]
    ground[:2, 1] = lo[1]
    ground[2:, 1] = hi[1]
    polys = [Poly(ground, "ground", 0, "medium_dry_ground")]
    for b, f in enumerate(footprints):
        polys.extend(building_by_polygon2d_and_height(f["ring"], f["height"], building=b + 1, material="brick"))
    return polys


# ---------------------------------------------------------------------------------------------
# frames and projections — geometry.py:236-291, 414-479
# ---------------------------------------------------------------------------------------------

def projection_matrix_from_vector(vector):
    """geometry.py:425-435"""
    z = vector / norm(vector)
    y = np.ones_like(z)
    y -= y.dot(z) * z
    y /= norm(y)
    x = np.cross(y, z)
    matrix = np.empty((3, 3))
    matrix[:, 0] = x
    matrix[:, 1] = y
    matrix[:, 2] = z
    return matrix


def projection_matrix_from_line_path(points):
    """geometry.py:449-452"""
    return projection_matrix_from_vector(points[1, :] - points[0, :])


def project_points(points, matrix):
    """geometry.py:236-291 without around_point: points @ matrix.T"""
    return np.atleast_2d(points) @ matrix.T


def point_on_edge_(point, edge, tol=1e-8):
    """geometry.py:457-479"""
    v = edge[1, :] - edge[0, :]
    matrix = projection_matrix_from_vector(v).T
    projected_edge = project_points(edge, matrix)
    projected_point = project_points(point, matrix).reshape(-1)
    A = projected_edge[0, :]
    B = projected_edge[1, :]
    return bool(A[0] - tol <= projected_point[0] <= A[0] + tol and
                A[1] - tol <= projected_point[1] <= A[1] + tol and
                A[2] - tol <= projected_point[2] <= B[2] + tol)


def intersection_of_2d_lines(points_A, points_B):
    """geometry.py:123-132"""
    h = np.ones((4, 3))
    h[:2, :2] = points_A
    h[2:, :2] = points_B
    l1 = np.cross(h[0], h[1])
    l2 = np.cross(h[2], h[3])
    x, y, z = np.cross(l1, l2)
    if z == 0:
        return np.full(2, np.nan)
    return np.array([x / z, y / z])


# ---------------------------------------------------------------------------------------------
# point in polygon — stands for shapely ``Polygon(points[:, :2]).intersects(Point(p))``
# ---------------------------------------------------------------------------------------------

def pip_inclusive(poly2d, p):
    """Crossing-number point-in-polygon on the ring ``poly2d`` (n,2); points on the boundary are inside."""
    x, y = float(p[0]), float(p[1])
    n = len(poly2d)
    inside = False
    for k in range(n):
        ax, ay = poly2d[k - 1]
        bx, by = poly2d[k]
        cross = (bx - ax) * (y - ay) - (by - ay) * (x - ax)
        if cross == 0.0 and min(ax, bx) <= x <= max(ax, bx) and min(ay, by) <= y <= max(ay, by):
            return True
        if (ay > y) != (by > y):
            # edge straddles the horizontal line through p: is the crossing to the right of p?
            if (cross > 0.0) == (by > ay):
                inside = not inside
    return inside


# ---------------------------------------------------------------------------------------------
# obstruction and containment — geometry.py:836-898, 1778-1803
# ---------------------------------------------------------------------------------------------

def polygons_obstruct_line_path(polys, points, return_indices=False):
    """geometry.py:850-898: first polygon whose projection along the segment contains the segment's foot point,
    whose extent along the segment overlaps it and whose plane is crossed at 1e-8 < t < 1 - 1e-8."""
    A = points[0, :]
    B = points[1, :]
    V = B - A
    matrix = projection_matrix_from_line_path(points).T
    projected_points = project_points(points, matrix)
    pA = projected_points[0, :]
    pB = projected_points[1, :]
    z_min = pA[2]
    z_max = pB[2]
    tol = 1e-8
    hits = []
    for idx, poly in enumerate(polys):
        pp = poly.points @ matrix.T
        if pp[:, 2].max() > z_min and pp[:, 2].min() < z_max:
            if pip_inclusive(pp[:, :2], pA):
                normal = poly.normal
                d = poly.parametric[3]
                with np.errstate(divide="ignore", invalid="ignore"):
                    t = - (d + np.dot(A, normal)) / np.dot(V, normal)
                if tol < t < 1 - tol:
                    if not return_indices:
                        return True
                    hits.append(idx)
    return hits if return_indices else False


def contains_point(poly, point, check_in_plane=False, plane_tol=1e-8):
    """geometry.py:1792-1803"""
    point = point.reshape(3)
    if check_in_plane:
        d = poly.parametric[3]
        if not np.allclose(np.dot(poly.normal, point), -d, atol=plane_tol):
            return False
    matrix = poly.matrix().T
    projected = poly.points @ matrix.T
    projected_point = (point.reshape(1, 3) @ matrix.T).reshape(3)
    return pip_inclusive(projected[:, :2], projected_point)


# ---------------------------------------------------------------------------------------------
# min-path-tracing zero functions and solvers — geometry.py:592-833
# ---------------------------------------------------------------------------------------------

def gamma(v):
    """geometry.py:593-605"""
    n = v.shape[0]
    norms = np.empty(n)
    for i in range(n):
        norms[i] = norm(v[i, :])
    g = np.empty((n - 1, 1))
    for i in range(n - 1):
        g[i, 0] = norms[i] / norms[i + 1]
    return g


def generate(origin, destination, planes_parametric):
    """geometry.py:610-632"""
    n = len(planes_parametric)
    normal = np.empty((n, 3))
    d = np.empty((n, 1))
    points = np.empty((n + 2, 3))
    points[0, :] = origin.ravel()
    points[-1, :] = destination.ravel()
    for i in range(n):
        normal[i, :] = planes_parametric[i][:3]
        d[i, 0] = planes_parametric[i][3]
        new_point = 0.5 * (points[i, :] + destination
                           - 2 * (d[i, 0] + np.dot(points[i, :], normal[i, :])) * normal[i, :])
        points[i + 1, :] = new_point.ravel()
    return normal, d, points


def reflexion_zero_func(x, normal, d, points):
    """geometry.py:638-657"""
    n = d.size
    for i in range(1, n + 1):
        j = 3 * (i - 1)
        points[i, 0] = x[j]
        points[i, 1] = x[j + 1]
        points[i, 2] = x[j + 2]
    v = points[1:, :] - points[:-1, :]
    g = gamma(v)
    dot_product = np.empty((n, 1))
    for i in range(n):
        dot_product[i, 0] = np.dot(points[i, :], normal[i, :])
    return (g * v[1:, :] - v[:-1, :] - 2 * (d + dot_product) * normal).reshape(-1)


def reflexion_points_from_origin_destination_and_planes(origin, destination, planes_parametric, **kwargs):
    """geometry.py:676-687"""
    A = origin.reshape(1, 3).astype(float)
    B = destination.reshape(1, 3).astype(float)
    normal, d, points = generate(A, B, np.asarray(planes_parametric))
    n = d.size
    with np.errstate(all="ignore"):
        sol = root(reflexion_zero_func, x0=points[1:n + 1, :].reshape(-1), args=(normal, d, points), **kwargs)
    return sol.x.reshape(-1, 3), sol


def diffraction_zero_func(x, ze, Xe, projected_A, projected_B):
    """geometry.py:694-715"""
    t = x - ze
    x = np.copy(Xe)
    x[2] += t[0]
    i = x - projected_A
    d = projected_B - x
    num1 = (ze - projected_A[2] + t)
    num2 = (projected_B[2] - ze - t)
    den1 = norm(i)
    den2 = norm(d)
    f = num1 / den1 - num2 / den2
    den1_prime = num1 / den1
    den2_prime = - num2 / den2
    jac = (-num1 - den1_prime) / (den1 * den1) - (num2 - den2_prime) / (den2 * den2)
    return f, jac


def diffraction_point_from_origin_destination_and_edge(origin, destination, edge, **kwargs):
    """geometry.py:733-748"""
    A = origin.reshape(1, 3).astype(float)
    B = destination.reshape(1, 3).astype(float)
    matrix = projection_matrix_from_line_path(edge).T
    projected_A = project_points(A, matrix)[0, :]
    projected_B = project_points(B, matrix)[0, :]
    projected_edge = project_points(edge, matrix)
    Xe = projected_edge[0, :]
    ze = np.atleast_1d(Xe[2])
    with np.errstate(all="ignore"):
        sol = root(diffraction_zero_func, x0=ze, jac=True, args=(ze, Xe, projected_A, projected_B), **kwargs)
    Xe[2] = sol.x[0]
    return project_points(Xe, matrix.T), sol


def reflexion_and_diffraction_zero_func(x, normal, d, points, matrix, ze, Xe, projected_B):
    """geometry.py:756-797 — note ``points[-1]`` stays at the edge's first endpoint for the whole solve."""
    n = d.size
    for i in range(1, n + 1):
        j = 3 * (i - 1)
        points[i, 0] = x[j]
        points[i, 1] = x[j + 1]
        points[i, 2] = x[j + 2]
    r = np.empty_like(x)
    v = points[1:, :] - points[:-1, :]
    g = gamma(v)
    dot_product = np.empty((n, 1))
    for i in range(n):
        dot_product[i] = np.dot(points[i, :], normal[i, :])
    projected_A = project_points(points[-2, :], matrix).ravel()
    r[:-1] = (g * v[1:, :] - v[:-1, :] - 2 * (d + dot_product) * normal).reshape(-1)
    t = x[-1] - ze
    xn = np.copy(Xe)
    xn[2] += t[0]
    i_ = xn - projected_A
    d_ = projected_B - xn
    num1 = (ze - projected_A[2] + t)
    num2 = (projected_B[2] - ze - t)
    den1 = norm(i_)
    den2 = norm(d_)
    r[-1] = (num1 / den1 - num2 / den2)[0]
    return r


def reflexion_points_and_diffraction_point_from_origin_destination_planes_and_edge(origin, destination,
                                                                                   planes_parametric, edge, **kwargs):
    """geometry.py:802-833"""
    if len(planes_parametric) == 0:
        return diffraction_point_from_origin_destination_and_edge(origin, destination, edge, **kwargs)
    elif edge is None:
        return reflexion_points_from_origin_destination_and_planes(origin, destination, planes_parametric, **kwargs)
    A = origin.reshape(1, 3).astype(float)
    B = destination.reshape(1, 3).astype(float)
    normal, d, points = generate(A, B, np.asarray(planes_parametric))
    n = len(planes_parametric)
    matrix = projection_matrix_from_line_path(edge).T
    projected_B = project_points(B, matrix)[0, :]
    projected_edge = project_points(edge, matrix)
    Xe = projected_edge[0, :]
    ze = np.atleast_1d(Xe[2])
    points[-1, :] = edge[0, :]
    with np.errstate(all="ignore"):
        sol = root(reflexion_and_diffraction_zero_func, x0=points[1:, :].flat[:-2],
                   args=(normal, d, points, matrix, ze, Xe, projected_B), **kwargs)
    x = sol.x
    points[1:n + 1, :] = x[:-1].reshape(-1, 3)
    points[-1, :] = project_points(np.array([Xe[0], Xe[1], x[-1]]), matrix.T)
    return points[1:, :], sol


# ---------------------------------------------------------------------------------------------
# sharp edges — geometry.py:482-565, array_utils.py:9-41
# ---------------------------------------------------------------------------------------------

def sort_by_columns(array):
    """array_utils.py:9-41 (lexicographic row sort, first column most significant)"""
    return array[np.lexsort([array[:, c] for c in reversed(range(array.shape[1]))])]


def polygons_sharp_edges(polys, min_angle=10.0):
    """geometry.py:499-565 → list of (edge (2,3), i, j) in pair order."""
    out = []
    max_cos = np.cos(np.deg2rad(min_angle))
    normals = [p.normal for p in polys]
    centroids = [centroid(p.points) for p in polys]
    domains = [domain(p.points) for p in polys]
    # The proximity test of geometry.py:518-521 is evaluated for all j > i of one i at once (same subtractions,
    # additions and comparisons as the pair-by-pair form, so the same pairs pass); the rest runs pair by pair.
    all_centroids = np.array(centroids).reshape(len(polys), 3)
    all_extents = np.array([np.diff(d, axis=0).reshape(3) for d in domains]).reshape(len(polys), 3)

    def near_pairs():
        for i in range(len(polys) - 1):
            far = np.any(np.abs(all_centroids[i] - all_centroids[i + 1:]) > all_extents[i] + all_extents[i + 1:], axis=1)
            for j in np.flatnonzero(~far):
                yield i, i + 1 + int(j)

    for i, j in near_pairs():
        normal_A, centroid_A = normals[i], centroids[i]
        normal_B, centroid_B = normals[j], centroids[j]
        if abs(np.dot(normal_A, normal_B)) >= max_cos:
            continue
        line_A = np.vstack([centroid_A, centroid_A + normal_A])
        line_B = np.vstack([centroid_B, centroid_B + normal_B])
        x = np.cross(normal_A, normal_B)
        matrix = projection_matrix_from_vector(x)
        projected_line_A = project_points(line_A, matrix.T)
        projected_line_B = project_points(line_B, matrix.T)
        intersection_point = intersection_of_2d_lines(projected_line_A[:, :2], projected_line_B[:, :2])
        if np.any(np.isnan(intersection_point)):
            continue
        A = projected_line_A[0, :2]
        V = projected_line_A[1, :2] - A
        if V[0] != 0:
            t = (intersection_point[0] - A[0]) / V[0]
        elif V[1] != 0:
            t = (intersection_point[1] - A[1]) / V[1]
        else:
            continue
        if t < 0:
            points_A = polys[i].points
            points_B = polys[j].points
            edges_A = [sort_by_columns(np.vstack([points_A[k - 1, :], points_A[k, :]]))
                       for k in range(points_A.shape[0])]
            edges_B = [sort_by_columns(np.vstack([points_B[m - 1, :], points_B[m, :]]))
                       for m in range(points_B.shape[0])]
            found = False
            for k in range(points_A.shape[0]):
                for m in range(points_B.shape[0]):
                    if np.allclose(edges_A[k], edges_B[m]):
                        out.append((edges_A[k], i, j))
                        found = True
                        break
                if found:
                    break
    return out


# ---------------------------------------------------------------------------------------------
# exact roots of the same systems (no iteration)
# ---------------------------------------------------------------------------------------------
# SciPy's hybr, as the reference calls it, does not always report the root of these systems even when one exists:
# on chains whose planes meet near the path it can stall on the planes' common line and still return success, and
# with the hand-written "jac" of geometry.py:710-713 it often reaches the Keller root to 1e-16 and returns
# success=False.  The functions below give the roots themselves; tests use them to decide whether a candidate that
# hybr rejected has a root of the reference's own zero function.

class ExactSolution:
    def __init__(self, x, success):
        self.x = x
        self.success = bool(success)


def exact_reflexion_points(origin, destination, planes_parametric):
    """Root of reflexion_zero_func: mirror the origin through the planes, then intersect backwards.  success=False
    when an image segment does not cross its plane strictly inside (then the system has no root)."""
    origin = np.asarray(origin, dtype=float).reshape(3)
    target = np.asarray(destination, dtype=float).reshape(3)
    planes = [np.asarray(p, dtype=float) for p in planes_parametric]
    images = []
    prev = origin
    for p in planes:
        prev = prev - 2 * (p[3] + prev @ p[:3]) * p[:3]
        images.append(prev)
    ok = True
    points = [None] * len(planes)
    for i in reversed(range(len(planes))):
        p = planes[i]
        direction = target - images[i]
        with np.errstate(divide="ignore", invalid="ignore"):
            t = -(p[3] + images[i] @ p[:3]) / (direction @ p[:3])
        if not (0.0 < t < 1.0):
            ok = False
        points[i] = images[i] + t * direction
        target = points[i]
    x = np.array(points).reshape(-1, 3)
    return x, ExactSolution(x.reshape(-1), ok)


def exact_diffraction_point(origin, destination, edge):
    """Root of diffraction_zero_func in the edge frame: z = zA + (zB - zA) rhoA / (rhoA + rhoB)."""
    A = np.asarray(origin, dtype=float).reshape(3)
    B = np.asarray(destination, dtype=float).reshape(3)
    matrix = projection_matrix_from_line_path(edge).T
    pA, pB, Xe = matrix @ A, matrix @ B, matrix @ edge[0]
    rho_a = np.hypot(Xe[0] - pA[0], Xe[1] - pA[1])
    rho_b = np.hypot(Xe[0] - pB[0], Xe[1] - pB[1])
    ok = rho_a + rho_b > 0
    with np.errstate(divide="ignore", invalid="ignore"):
        z = pA[2] + (pB[2] - pA[2]) * (rho_a / (rho_a + rho_b))
    x = (np.array([Xe[0], Xe[1], z]) @ matrix).reshape(1, 3)
    return x, ExactSolution(np.array([z]), ok)


def exact_reflexion_points_and_diffraction_point(origin, destination, planes_parametric, edge):
    """Root of reflexion_and_diffraction_zero_func: its reflection block sees the edge's first endpoint as its
    destination (geometry.py:821), its last equation the last reflection point and the receiver."""
    if len(planes_parametric) == 0:
        return exact_diffraction_point(origin, destination, edge)
    if edge is None:
        return exact_reflexion_points(origin, destination, planes_parametric)
    x, sol = exact_reflexion_points(origin, edge[0], planes_parametric)
    d, sol_d = exact_diffraction_point(x[-1], destination, edge)
    return np.vstack([x, d]), ExactSolution(np.concatenate([x.reshape(-1), sol_d.x]), sol.success and sol_d.success)
