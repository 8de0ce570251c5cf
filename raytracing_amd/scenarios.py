"""Deterministic benchmark / test scenarios of the path-solve path (scene file, emitter, receiver lattice).

The reference builds its study scenes by hand in ``electromagnetism_fun/place_utils.py`` (e.g. ``create_small_place``
:53-73: TX (3, 38, 18), receivers (5, 15 - 4 i, 1.5)); the lattices below are synthetic inputs of the same shape for
the larger scenes, all derived from the scene file and seed 0.
"""
import os

import numpy as np

from . import geometry as geom

DATA_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")


def _footprints(place):
    """Rings of the buildings' footprints, of a place or of a ``geometry.LoadedScene`` (which holds them ready)."""
    if hasattr(place, "footprints"):
        return place.footprints
    return [p.get_top_face().points[:, :2] for p in place.polyhedra]


def _inside_any(footprints, boxes, x, y):
    for ring, (lo, hi) in zip(footprints, boxes):
        if lo[0] <= x <= hi[0] and lo[1] <= y <= hi[1]:
            inside = False
            n = len(ring)
            for k in range(n):
                ax, ay = ring[k - 1]
                bx, by = ring[k]
                cr = (bx - ax) * (y - ay) - (by - ay) * (x - ax)
                if cr == 0 and min(ax, bx) <= x <= max(ax, bx) and min(ay, by) <= y <= max(ay, by):
                    return True
                if (ay > y) != (by > y) and ((cr > 0) == (by > ay)):
                    inside = not inside
            if inside:
                return True
    return False


def _inside_any_many(footprints, boxes, xs, ys):
    """``_inside_any`` for arrays of points (the same predicate, evaluated footprint by footprint on the points inside
    its box)."""
    inside = np.zeros(len(xs), dtype=bool)
    for ring, (lo, hi) in zip(footprints, boxes):
        sel = np.flatnonzero((lo[0] <= xs) & (xs <= hi[0]) & (lo[1] <= ys) & (ys <= hi[1]) & ~inside)
        if len(sel) == 0:
            continue
        inside[sel] = geom.ring_contains(ring, xs[sel], ys[sel])  # the same crossing-number and on-edge tests, all sides at once
    return inside


def lattice_receivers(place, n_side, z, inset=0.05, count=None):
    """n_side × n_side lattice over the ground box shrunk by ``inset``, at height z; lattice points inside a
    footprint (boundary included) are dropped and replaced by points of the same lattice shifted by half a pitch,
    taken in row order, until ``count`` (default n_side²) receivers exist."""
    count = n_side * n_side if count is None else count
    lo, hi = place.get_domain()
    cx, cy = (lo[0] + hi[0]) / 2, (lo[1] + hi[1]) / 2
    hx, hy = (hi[0] - lo[0]) / 2 * (1 - inset), (hi[1] - lo[1]) / 2 * (1 - inset)
    xs = np.linspace(cx - hx, cx + hx, n_side)
    ys = np.linspace(cy - hy, cy + hy, n_side)
    fps = _footprints(place)
    boxes = [(f.min(axis=0), f.max(axis=0)) for f in fps]
    out = []
    px, py = (xs[1] - xs[0]) / 2 if n_side > 1 else 0.0, (ys[1] - ys[0]) / 2 if n_side > 1 else 0.0
    for shift in range(0, 64):
        # pass 0 is the lattice itself; pass s >= 1 is the lattice moved by (half pitch) / s
        ox, oy = (px / shift, py / shift) if shift else (0.0, 0.0)
        gx, gy = np.meshgrid(xs + ox, ys + oy)  # row order: y outer, x inner
        gx, gy = gx.ravel(), gy.ravel()
        ok = ~((np.abs(gx - cx) > hx) | (np.abs(gy - cy) > hy))
        ok[ok] = ~_inside_any_many(fps, boxes, gx[ok], gy[ok])
        for k in np.flatnonzero(ok):
            out.append((gx[k], gy[k], z))
            if len(out) == count:
                return np.array(out, dtype=float)
    raise RuntimeError("could not place the requested number of receivers outside the footprints")


def emitter_outside(place, x, y, z):
    """(x, y, z) moved in +x by 1 m steps until it is outside every footprint."""
    fps = _footprints(place)
    boxes = [(f.min(axis=0), f.max(axis=0)) for f in fps]
    while _inside_any(fps, boxes, x, y):
        x += 1.0
    return np.array([x, y, z], dtype=float)


def small_place(n_rx=16):
    """Config 1: data/small.geojson, TX (3, 38, 18), receivers (5, 15 - 4 i, 1.5)."""
    place = geom.load_place(os.path.join(DATA_DIR, "small.geojson"))
    tx = np.array([3.0, 38.0, 18.0])
    rx = np.array([[5.0, 15.0 - 4.0 * i, 1.5] for i in range(n_rx)])
    return place, tx, rx


def dummy_place():
    """The reference's create_dummy_place (electromagnetism_fun/place_utils.py:76-103): a 70 m x 24 m road at 10 m
    elevation with three 4 m x 2 m x 10 m blocks along its sides, TX (5, 12, 15), one receiver at (65, 12, 15)."""
    from .materials_properties import set_properties

    z0 = 10.0
    ground = geom.Square.by_2_corner_points(np.array([[0.0, 0.0, z0], [70.0, 24.0, z0]]))
    ground = ground.rotate(axis=np.array([0, 1, 0]), angle_deg=180)
    ground.properties = set_properties("road")
    buildings = []
    for (x0, y0), (x1, y1) in (((13.0, 22.0), (17.0, 24.0)), ((33.0, 0.0), (37.0, 2.0)), ((53.0, 22.0), (57.0, 24.0))):
        square = geom.Square.by_2_corner_points(np.array([[x0, y0, z0], [x1, y1, z0]]))
        building = geom.Building.by_polygon_and_height(polygon=square, height=10)
        building.building_type = "appartments"
        building.apply_properties_to_polygons()
        buildings.append(building)
    place = geom.OrientedPlace(geom.OrientedSurface(ground), buildings)
    tx = np.array([5.0, 12.0, z0 + 5.0])
    rx = np.array([[65.0, 12.0, z0 + 5.0]])
    return place, tx, rx


def two_rays_place(npoints=20):
    """The reference's create_two_rays_place (electromagnetism_fun/place_utils.py:106-123): a concrete ground of
    (npoints + 50) m x 24 m turned to face up, TX (5, 12, 30), receivers (55 + i, 12, 10)."""
    from .materials_properties import set_properties

    ground = geom.Square.by_2_corner_points(np.array([[0.0, 0.0, 0.0], [1.0 * npoints + 50.0, 24.0, 0.0]]))
    ground = ground.rotate(axis=np.array([0, 1, 0]), angle_deg=180)
    ground.properties = set_properties("road")
    place = geom.OrientedPlace(geom.OrientedSurface(ground))
    tx = np.array([5.0, 12.0, 30.0])
    rx = np.array([tx + np.array([50.0 + i, 0.0, -20.0]) for i in range(npoints)])
    return place, tx, rx


def levant_grid(n_side=32):
    """Config 2: data/place_levant.geojson, TX above the scene centre at 18.7 m, n_side² receivers at 1.2 m."""
    place = geom.load_place(os.path.join(DATA_DIR, "place_levant.geojson"))
    lo, hi = place.get_domain()
    tx = emitter_outside(place, (lo[0] + hi[0]) / 2, (lo[1] + hi[1]) / 2, 18.7)
    return place, tx, lattice_receivers(place, n_side, 1.2)


def lln_place():
    """The scene and the emitter of config 3 (data/lln.geojson, TX above the scene centre at 30 m), without receivers."""
    place = geom.load_place(os.path.join(DATA_DIR, "lln.geojson"))
    lo, hi = place.get_domain()
    return place, emitter_outside(place, (lo[0] + hi[0]) / 2, (lo[1] + hi[1]) / 2, 30.0)


def lln_scene():
    """Config 3 for callers that need arrays only: (geometry.LoadedScene of data/lln.geojson, emitter), the same scene
    and emitter as ``lln_place`` without the place's objects (they are made if ``.place`` is asked for)."""
    scene = geom.load_scene(os.path.join(DATA_DIR, "lln.geojson"))
    lo, hi = scene.get_domain()
    return scene, emitter_outside(scene, (lo[0] + hi[0]) / 2, (lo[1] + hi[1]) / 2, 30.0)


def lln_grid(n_side=100):
    """Config 3: data/lln.geojson, TX above the scene centre at 30 m, n_side² receivers at 1.5 m."""
    place, tx = lln_place()
    return place, tx, lattice_receivers(place, n_side, 1.5)


def ny_tiled(tiles=8, gap=10.0, n_rx=256, seed=0):
    """Config 4: data/ny.geojson repeated ``tiles`` x ``tiles`` times with a ``gap`` m street between the copies
    (92 polygons -> 5 825 for 8 x 8), TX above the centre at 20 m, ``n_rx`` receivers on a lattice at 1.5 m."""
    footprints = geom.preprocess_geojson(os.path.join(DATA_DIR, "ny.geojson"), seed=seed)
    all_xy = np.concatenate([f["ring"] for f in footprints])
    lo, hi = all_xy.min(axis=0), all_xy.max(axis=0)
    pitch = (hi - lo) + gap
    tiled = []
    for ty in range(tiles):
        for tx_ in range(tiles):
            for f in footprints:
                g = dict(f)
                g["ring"] = f["ring"] + np.array([tx_ * pitch[0], ty * pitch[1]])
                g["id"] = f"{f['id']}-{ty}-{tx_}"
                tiled.append(g)
    place = geom.generate_place_from_rooftops_file(tiled)
    lo3, hi3 = place.get_domain()
    tx = emitter_outside(place, (lo3[0] + hi3[0]) / 2, (lo3[1] + hi3[1]) / 2, 20.0)
    n_side = int(np.ceil(np.sqrt(n_rx)))
    return place, tx, lattice_receivers(place, n_side, 1.5, count=n_rx)


# emitter positions of the reference's measurement campaign on place_saint_jean (scenario_measures.py:157-167), lifted
# to 3 m above the ground
SAINT_JEAN_EMITTERS = np.array([[35.0, -6.0, 3.0], [69.0, -54.0, 3.0], [53.0, -93.0, 3.0], [70.0, 41.5, 3.0],
                                [-96.0, 92.0, 3.0]])
SWEEP_FREQUENCIES_HZ = np.array([3.5e9, 6e9, 12.5e9, 28e9, 30e9, 39e9, 60e9])


def saint_jean_sweep(n_rx=5000, seed=0, min_spacing=1.0):
    """Config 5: data/place_saint_jean, the five emitters above (moved out of any footprint), ``n_rx`` receivers drawn
    uniformly in the ground box outside the footprints (seeded rejection sampling, at least ``min_spacing`` apart on a
    grid of that pitch, z = 1.2 m), and the sweep frequencies."""
    place = geom.load_place(os.path.join(DATA_DIR, "place_saint_jean", "place_saint_jean.geojson"))
    lo, hi = place.get_domain()
    fps = _footprints(place)
    boxes = [(f.min(axis=0), f.max(axis=0)) for f in fps]
    rng = np.random.default_rng(seed)
    taken = set()
    out = []
    while len(out) < n_rx:
        x, y = rng.uniform(lo[0], hi[0]), rng.uniform(lo[1], hi[1])
        cell = (int(np.floor(x / min_spacing)), int(np.floor(y / min_spacing)))
        if cell in taken or _inside_any(fps, boxes, x, y):
            continue
        taken.add(cell)
        out.append((x, y, 1.2))
    emitters = np.array([emitter_outside(place, *e) for e in SAINT_JEAN_EMITTERS])
    return place, emitters, np.array(out), SWEEP_FREQUENCIES_HZ.copy()
