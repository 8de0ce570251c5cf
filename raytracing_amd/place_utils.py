"""Builders of the scenes the reference's studies are run on (electromagnetism_fun/place_utils.py): the small and
dummy places, the two-ray place, Place du Levant flat and with its slanted street, Place Saint-Jean flat and on its
slanted ground, and the receiver points of a street.  Host code: NumPy, the tolerant GeoJSON reader and the projection
of ``raytracing_amd.geometry``; no GPU call is made here.

What the reference does through GeoPandas, PROJ and Shapely is done here with the package's own reader, EPSG:3035
formulas and ring predicates, so coordinates agree with the reference's to the centimetre and random draws
(heights of buildings without one, tree heights and turns, receiver points) come from seeded generators.
"""
import json
import os
import random

import numpy as np

from . import geometry as geom
from . import scenarios
from .materials_properties import set_properties

DATA_DIR = scenarios.DATA_DIR


def _data_file(*parts):
    return os.path.join(DATA_DIR, *parts)


def _up(z):
    return np.array([0.0, 0.0, float(z)])


# ------------------------------------------------------------------------------------------------------------------
# the three hand-made places (place_utils.py:53-123)
# ------------------------------------------------------------------------------------------------------------------
def create_small_place(npoints=3):
    """small.geojson with ``npoints`` receivers 4 m apart down the street (place_utils.py:53-73)."""
    place = geom.load_place(_data_file("small.geojson"))
    rx = np.array([5.0, 15.0, 1.5])
    for k in range(npoints):
        place.add_set_of_points((rx + k * np.array([0.0, -4.0, 0.0])).reshape(-1, 3))
    return place, np.array([3.0, 38.0, 18.0]).reshape(-1, 3), "small"


def create_dummy_place():
    """Three towers beside a 70 m x 24 m road at 10 m elevation (place_utils.py:76-103)."""
    z0 = 10
    ground = geom.Square.by_2_corner_points(np.array([[0, 0, z0], [70, 24, z0]])).rotate(axis=np.array([0, 1, 0]),
                                                                                         angle_deg=180)
    ground.properties = set_properties("road")
    buildings = []
    for (x0, y0), (x1, y1) in (((13, 22), (17, 24)), ((33, 0), (37, 2)), ((53, 22), (57, 24))):
        footprint = geom.Square.by_2_corner_points(np.array([[x0, y0, z0], [x1, y1, z0]]))
        tower = geom.Building.by_polygon_and_height(polygon=footprint, height=10)
        tower.building_type = "appartments"
        tower.apply_properties_to_polygons()
        buildings.append(tower)
    place = geom.OrientedPlace(geom.OrientedSurface(ground), buildings)
    place.add_set_of_points(np.array([65.0, 12.0, z0 + 5.0]).reshape(-1, 3))
    return place, np.array([5.0, 12.0, z0 + 5.0]).reshape(-1, 3), "dummy"


def create_two_rays_place(npoints=20):
    """A road and nothing else: line of sight plus one ground reflection (place_utils.py:106-123)."""
    step = 1
    ground = geom.Square.by_2_corner_points(np.array([[0, 0, 0], [step * npoints + 50, 24, 0]]))
    ground = ground.rotate(axis=np.array([0, 1, 0]), angle_deg=180)
    ground.properties = set_properties("road")
    place = geom.OrientedPlace(geom.OrientedSurface(ground))
    tx = np.array([5.0, 12.0, 30.0]).reshape(-1, 3)
    first = tx + np.array([50, 0, -20])
    for k in range(npoints):
        place.add_set_of_points(first + np.array([k * step, 0, 0]))
    return place, tx, "two_rays"


def rebuild_buildings_on_slanted_grounds(place, grounds):
    """Every building of ``place`` rebuilt so that it stands on ``grounds`` with its roof where it was; a building
    over several grounds becomes one building per ground (place_utils.py:126-136)."""
    buildings = []
    for polyhedron in place.polyhedra:
        top = polyhedron.get_top_face()
        for building in geom.Building.rebuild_building(top, grounds, top.points[0][2]):
            building.building_type = polyhedron.building_type
            building.apply_properties_to_polygons()
            buildings.append(building)
    return buildings


# ------------------------------------------------------------------------------------------------------------------
# nodes of a GeoJSON file in the coordinates of the place built from it
# ------------------------------------------------------------------------------------------------------------------
def _projected_features(filename):
    """[(id, type, xy array)] of every Point and Polygon feature, projected to EPSG:3035 and centred on the middle of
    the bounding box of ALL of them: ``GeoDataFrame.to_crs`` followed by ``center_gdf`` on the unfiltered file
    (place_utils.py:143-145, geometry.py:2806-2812)."""
    data = geom.read_geojson_tolerant(filename)
    out = []
    for feature in data["features"]:
        g = feature.get("geometry") or {}
        props = feature.get("properties") or {}
        ident = props.get("id") or feature.get("id") or props.get("@id")
        if g.get("type") == "Point":
            lonlat = np.asarray(g["coordinates"], dtype=float)[None, :2]
        elif g.get("type") == "Polygon":
            lonlat = np.asarray(g["coordinates"][0], dtype=float)[:, :2]
        else:
            continue
        x, y = geom.lonlat_to_epsg3035(lonlat[:, 0], lonlat[:, 1])
        out.append((ident, g["type"], np.column_stack([x, y])))
    every = np.concatenate([xy for _, _, xy in out])
    middle = (every.min(axis=0) + every.max(axis=0)) / 2
    return [(ident, kind, xy - middle) for ident, kind, xy in out]


def _bounds_corners(features):
    """The four corners of the features' bounding box at z = 0 in the order of ``get_bounds`` (geometry.py:2815-2822)."""
    every = np.concatenate([xy for _, _, xy in features])
    lo, hi = every.min(axis=0), every.max(axis=0)
    return np.array([[lo[0], lo[1], 0.0], [hi[0], lo[1], 0.0], [hi[0], hi[1], 0.0], [lo[0], hi[1], 0.0]])


class Place_du_levant:
    """Place du Levant, Louvain-la-Neuve (place_utils.py:139-327)."""

    FILE = "place_levant_edited.geojson"

    @staticmethod
    def levant_find_crucial_coordinates(filename=None):
        """(entrances [Maxwell, Stevin, Réaumur], bottom of the street, corners of the file's bounding box, tree spots
        [Vinci, Maxwell, Stevin]) from the nodes of the edited file (place_utils.py:141-169)."""
        features = _projected_features(filename or _data_file(Place_du_levant.FILE))
        nodes = {ident: np.array([xy[0, 0], xy[0, 1], 0.0]) for ident, kind, xy in features if kind == "Point"}
        maxwell, stevin, reaumur = nodes["node/2368705254"], nodes["node/8804430834"], nodes["node/2034353808"]
        levant_bottom = maxwell + np.array([-94, 0, 0])  # the February measurements
        tree_spots = [nodes["node/5710653390"], nodes["node/5710653391"], nodes["node/5710653392"]]
        return [maxwell, stevin, reaumur], levant_bottom, _bounds_corners(features), tree_spots

    @staticmethod
    def levant_add_trees(tree_spots):
        """Trunks and crowns of the three trees of the square (place_utils.py:172-196)."""
        sizes = [((0.5, 4), (2, 5)), ((0.5, 4.5), (2, 4.5)), ((0.5, 6.5), (2, 2))]  # (trunk, crown) x (side, height)
        trunks, crowns = [], []
        for spot, ((trunk_side, trunk_height), (crown_side, crown_height)) in zip(tree_spots, sizes):
            trunk = geom.Building.create_tree_trunk(spot, trunk_side, trunk_height, False, crown_side, crown_height)
            trunks.append(trunk)
            crowns.append(geom.Building.create_tree_crown(trunk, crown_side, crown_height, False))
        return trunks, crowns

    @staticmethod
    def _street_points(place, top_of_street, levant_bottom, npoints, on_ground=None):
        """Emitter on the Maxwell's roof and ``npoints`` receivers down the street from its entrance
        (place_utils.py:208-219, 293-309)."""
        rx_height = tx_height = 1.2
        maxwell_height = 17.5
        tx = (top_of_street + np.array([23, 6, maxwell_height + tx_height])).reshape(-1, 3)
        first = top_of_street + np.array([-34, 0, rx_height])
        step = np.linalg.norm(levant_bottom - first) / npoints
        for k in range(npoints):
            rx = first + np.array([-step * k, 0, 0])
            if on_ground is not None:
                rx = on_ground(rx, rx_height)
            place.add_set_of_points(rx.reshape(-1, 3))
        return tx

    @staticmethod
    def create_flat_levant(npoints=15, seed=0):
        """(place, tx, "flat_levant"): the square on flat ground with its three tree trunks; the crowns are added
        after the solve by ``my_field_computation`` (place_utils.py:199-236)."""
        filename = _data_file(Place_du_levant.FILE)
        place = geom.load_place(filename, seed=seed)
        entrances, levant_bottom, _, tree_spots = Place_du_levant.levant_find_crucial_coordinates(filename)
        trunks, _ = Place_du_levant.levant_add_trees(tree_spots)
        place.polyhedra.extend(trunks)
        tx = Place_du_levant._street_points(place, entrances[0], levant_bottom, npoints)
        return place, tx, "flat_levant"

    @staticmethod
    def create_slanted_levant(npoints=15, seed=0):
        """(place, tx, "slanted_levant"): three grounds — flat, a slope of 3 m between the Réaumur's and the Stevin's
        entrances, flat again — with every building rebuilt on them (place_utils.py:239-327)."""
        filename = _data_file(Place_du_levant.FILE)
        flat = geom.load_place(filename, seed=seed)
        entrances, levant_bottom, corners, tree_spots = Place_du_levant.levant_find_crucial_coordinates(filename)
        maxwell, stevin, reaumur = entrances
        rise = 3
        tree_spots = [spot + _up(rise) for spot in tree_spots]
        bottom_left = corners[0]
        top_right = corners[2] + _up(rise)
        low_right = np.array([reaumur[0], top_right[1], 0.0])      # where the slope starts
        high_left = np.array([stevin[0], bottom_left[1], rise])    # where it ends
        ground1 = geom.Square.by_2_corner_points(np.array([bottom_left, low_right]))
        ground2 = geom.Square.by_2_corner_points(np.array([low_right, high_left]))
        ground3 = geom.Square.by_2_corner_points(np.array([high_left, top_right]))
        # normals up
        ground1 = ground1.rotate(axis=np.array([0, 1, 0]), angle_deg=180)
        ground3 = ground3.rotate(axis=np.array([0, 1, 0]), angle_deg=180)
        grounds = [ground1, ground2, ground3]
        for ground in grounds:
            ground.properties = set_properties("road")
        maxwell = geom.polygon_line_intersection(grounds[2], [maxwell, maxwell + _up(100)])
        levant_bottom = geom.polygon_line_intersection(grounds[0], [levant_bottom, levant_bottom + _up(100)])
        buildings = rebuild_buildings_on_slanted_grounds(flat, grounds)
        trunks, _ = Place_du_levant.levant_add_trees(tree_spots)
        buildings.extend(trunks)
        place = geom.OrientedPlace(geom.OrientedSurface(grounds), buildings)

        def on_ground(rx, rx_height):
            if rx[0] < high_left[0]:  # on the slope
                rx = geom.polygon_line_intersection(grounds[1], [rx, rx + _up(100)]) + _up(rx_height)
            if rx[0] < low_right[0]:  # on flat ground again
                rx[2] = rx_height
            return rx

        tx = Place_du_levant._street_points(place, maxwell, levant_bottom, npoints, on_ground)
        return place, tx, "slanted_levant"


class Place_saint_jean:
    """Place Saint-Jean, Brussels (place_utils.py:394-531)."""

    FILE = os.path.join("place_saint_jean", "place_saint_jean.geojson")

    @staticmethod
    def gen_place_saint_jean(seed=0):
        return geom.load_place(_data_file(Place_saint_jean.FILE), seed=seed)

    @staticmethod
    def load_points_from_file(filename):
        with open(filename) as f:
            return [tuple(point) for point in json.load(f)]

    @staticmethod
    def _points(points):
        if isinstance(points, (str, bytes)) or hasattr(points, "__fspath__"):
            points = Place_saint_jean.load_points_from_file(points)
        return np.asarray(points, dtype=float).reshape(-1, 3)

    @staticmethod
    def create_place_saint_jean(points_filename, tx=None, seed=0):
        """(place, tx, "place_saint_jean") on flat ground; ``points_filename``: a file ``create_points`` wrote, or the
        points themselves (place_utils.py:403-417)."""
        place = Place_saint_jean.gen_place_saint_jean(seed)
        place.add_set_of_points(Place_saint_jean._points(points_filename))
        if tx is None:
            tx = np.array([72, 40, 2])
        return place, np.asarray(tx, dtype=float).reshape(-1, 3), "place_saint_jean"

    @staticmethod
    def create_slanted_place_saint_jean(points_filename, tx=None, tx_on_ground=True, seed=0):
        """(place, tx, "slanted_place_saint_jean"): one ground tilted from 16 m to 34 m and turned by -80 degrees, the
        buildings rebuilt on it, the receivers 1.2 m and the emitter 3 m above what lies below them
        (place_utils.py:420-482)."""
        rx_height, tx_height = 1.2, 3
        flat = Place_saint_jean.gen_place_saint_jean(seed)
        ground = geom.Square.by_2_corner_points(np.array([[-170, -150, 16], [150, 150, 34]]))
        ground = ground.rotate(axis=np.array([0, 1, 0]), angle_deg=180)
        ground = ground.rotate(axis=np.array([0, 0, 1]), angle_deg=-80)
        ground.properties = set_properties("road")
        ground.part = "ground"
        grounds = [ground]
        place = geom.OrientedPlace(geom.OrientedSurface(grounds), rebuild_buildings_on_slanted_grounds(flat, grounds))
        # (the reference adds the flat points to the place it then throws away, and the lifted ones to the new place)
        points = Place_saint_jean._points(points_filename).copy()
        for g in place.surface.polygons:
            over = geom.ring_contains(g.points[:, :2], points[:, 0], points[:, 1]) & \
                   (geom.ring_distance(g.points[:, :2], points[:, 0], points[:, 1]) > 0)
            for i in np.flatnonzero(over):
                points[i] = geom.polygon_line_intersection(g, [points[i], points[i] + _up(1)]) + _up(rx_height)
        place.add_set_of_points(points)
        if tx is None:
            tx = np.array([25, 2, 18])
        tx = np.asarray(tx, dtype=float).reshape(3)
        base = place.surface.polygons[0] if tx_on_ground else place.polyhedra[29].get_top_face()
        tx = geom.polygon_line_intersection(base, [tx, tx + _up(1)]) + _up(tx_height)
        return place, tx.reshape(-1, 3), "slanted_place_saint_jean"

    @staticmethod
    def generate_spaced_points_in_polygon(polygon, n, d):
        """``n`` random points of the 2-D ring ``polygon`` more than 1 m from its boundary and at least ``d`` apart:
        ``random.uniform`` over the ring's box, kept or thrown away one by one (place_utils.py:485-503).  Seed
        Python's ``random`` for a reproducible set.  Loops for ever when the ring has no room for them."""
        ring = np.asarray(polygon, dtype=float)[:, :2]
        lo, hi = ring.min(axis=0).tolist(), ring.max(axis=0).tolist()
        points = []
        while len(points) < n:
            x, y = random.uniform(lo[0], hi[0]), random.uniform(lo[1], hi[1])
            if not geom.ring_contains(ring, x, y)[0] or geom.ring_distance(ring, x, y)[0] <= 1:
                continue
            if points and np.min(np.hypot(np.array(points)[:, 0] - x, np.array(points)[:, 1] - y)) < d:
                continue
            points.append((x, y))
        return points

    @staticmethod
    def create_points(street_polygon, npoints=16 * 5, d=4, filename=None):
        """``npoints`` receivers 1.2 m above a street given as a 2-D ring, written to ``filename`` when given
        (place_utils.py:506-516).  The reference cuts its street polygon out of a Shapely union of the roofs whose
        vertex numbering is GEOS's own (place_utils.py:534-589); here the street is the caller's."""
        points = [(x, y, 1.2) for x, y in Place_saint_jean.generate_spaced_points_in_polygon(street_polygon, npoints, d)]
        if filename is not None:
            with open(filename, "w") as f:
                json.dump(points, f)
        return points, filename
