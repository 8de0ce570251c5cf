"""Generates tests/golden/trees.npz from the REFERENCE's tree builders (raytracing/geometry.py: Square :1942-2002,
OrientedPolygon.rotate :1900-1920, Building.create_tree_trunk / create_tree_crown :2475-2496); authoring container
only, see make_golden.py.  Python's ``random`` is seeded before every call that draws from it."""
import os
import random
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
warnings.filterwarnings("ignore")

from _ref_import import import_reference_geometry  # noqa: E402

geom = import_reference_geometry()


def main():
    out = {}
    cases = [((4.0, -3.0, 0.0), 0.6, 7.0, False, 5.0, 6.0, 1), ((-10.5, 22.25, 1.5), 2.0, None, True, 4.0, 3.5, 2),
             ((0.0, 0.0, 0.0), 1.0, 12.0, True, 7.0, 8.0, 3)]
    for k, (spot, size, height, rotate, crown_size, crown_height, seed) in enumerate(cases):
        random.seed(seed)
        trunk = geom.Building.create_tree_trunk(np.array(spot), size, height, rotate, crown_size=crown_size,
                                                crown_height=crown_height)
        crown = geom.Building.create_tree_crown(trunk, crown_size, crown_height, rotate=False)
        out[f"case{k}_args"] = np.array(list(spot) + [size, -1.0 if height is None else height, float(rotate), crown_size,
                                                       crown_height, seed])
        out[f"case{k}_trunk"] = np.array([p.points for p in trunk.polygons])
        out[f"case{k}_crown"] = np.array([p.points for p in crown.polygons])
        out[f"case{k}_trunk_planes"] = np.array([p.get_parametric() for p in trunk.polygons])
        out[f"case{k}_crown_planes"] = np.array([p.get_parametric() for p in crown.polygons])
        assert trunk.building_type == "tree_trunk" and crown.building_type == "tree_crown"
        out[f"case{k}_materials"] = np.array([trunk.polygons[0].properties["material"], crown.polygons[0].properties["material"]])
    sq = geom.Square.by_2_corner_points(np.array([[3.0, 4.0, 1.0], [-1.0, -2.0, 1.0]]))
    out["square_points"] = sq.points
    out["square_center"] = geom.Square.center(sq.points)
    np.savez_compressed(os.path.join(HERE, "trees.npz"), **out)
    print({k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    main()
