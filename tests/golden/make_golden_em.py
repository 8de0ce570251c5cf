"""Generates tests/golden/em.npz from the REFERENCE's field code (authoring container only; see make_golden.py).

Two reference functions cannot run here because they go through Shapely:
  * ElectromagneticField.account_for_trees (a full-scene obstruction scan that changes nothing for an unobstructed
    segment) is replaced by the identity for the vectors of my_reflect / my_diff;
  * OrientedPolygon.contains_point, which my_diff uses only to orient the edge, is answered by the footprint test
    written out below for the one rectangular face involved.
Everything else (antenna frames, source field, Fresnel coefficients, dyadics, transition function, wedge diffraction
coefficients, spreading and phase factors) is the reference's arithmetic untouched.
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
warnings.filterwarnings("ignore")

from _ref_import import import_reference_electromagnetism  # noqa: E402

geom, em = import_reference_electromagnetism()
from electromagnetism_fun.materials_properties import set_properties  # noqa: E402


class FakeProblem:
    def __init__(self, polygons):
        self.polygons = polygons
        self.place = None


def main():
    import contextlib
    import io

    rng = np.random.default_rng(77)
    out = {}
    em.ElectromagneticField.account_for_trees = staticmethod(lambda E, points, place: E)

    # antennas and source field
    pos = rng.uniform(-20, 20, (10, 3))
    tgt = rng.uniform(-20, 20, (10, 3))
    probe = rng.uniform(-30, 30, (10, 3))
    mats, angles, fields = [], [], []
    for p, t, q in zip(pos, tgt, probe):
        a = em.Antenna(p)
        a.align_antenna_to_point(t)
        mats.append(a.mat)
        angles.append(a.incident_angles(q))
        fields.append(em.ElectromagneticField.from_path([p, q], a))
    out.update(ant_pos=pos, ant_target=tgt, ant_probe=probe, ant_mat=np.array(mats), ant_angles=np.array(angles),
               src_field=np.array(fields))
    # vertical target: the frame degenerates (x = 0), kept as the reference produces it
    a = em.Antenna(np.array([1.0, 2.0, 3.0]))
    a.align_antenna_to_point(np.array([1.0, 2.0, 9.0]))
    out["ant_vertical_mat"] = a.mat

    # Fresnel coefficients per material and angle
    types = ["office", "appartments", "garage", "ground", "tree_trunk"]
    thetas = np.array([0.05, 0.3, 0.8, 1.2, 1.5])
    coeffs = []
    for t in types:
        poly = geom.OrientedPolygon(np.array([[0.0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]]))
        poly.properties = set_properties(t)
        with contextlib.redirect_stdout(io.StringIO()):
            coeffs.append([em.Reflection.fresnel_coeffs(th, poly) for th in thetas])
    out.update(fresnel_types=np.array(types), fresnel_thetas=thetas, fresnel_coeffs=np.array(coeffs))

    # reflections on the faces of a 20 m cube around the origin (outward normals): paths outside the cube
    cube = geom.Cube.by_point_and_side_length(np.zeros(3), 20.0)
    for q in cube.polygons:
        q.properties = set_properties("appartments")
    rtp = FakeProblem(cube.polygons)
    tx = np.array([30.0, 3.0, 4.0])
    ant = em.Antenna(tx)
    ant.align_antenna_to_point(np.array([10.0, 1.0, 2.0]))
    refl_paths, refl_R, refl_out = [], [], []
    for _ in range(12):
        hit = np.array([10.0, rng.uniform(-9, 9), rng.uniform(-9, 9)])
        rx = np.array([rng.uniform(15, 40), rng.uniform(-20, 20), rng.uniform(-5, 15)])
        path = np.array([tx, hit, rx])
        with contextlib.redirect_stdout(io.StringIO()):
            R, e_per, ei_par, er_par = em.Reflection.dyadic_ref_coeff(path, cube.polygons[2])
            E_i = em.ElectromagneticField.from_path([tx, hit], ant)
            try:
                E = em.Reflection.my_reflect(E_i, path, [2], rtp)
            except AssertionError:
                continue
        refl_paths.append(path)
        refl_R.append(R)
        refl_out.append(E)
    out.update(refl_tx=tx, refl_target=np.array([10.0, 1.0, 2.0]), refl_paths=np.array(refl_paths),
               refl_R=np.array(refl_R), refl_E=np.array(refl_out), refl_polygon=cube.polygons[2].points,
               refl_parametric=cube.polygons[2].get_parametric())

    # transition function and A±
    xs = np.concatenate([np.logspace(-6, 4.5, 60), [0.0]])
    out["F_x"] = xs
    out["F_val"] = np.array([em.Diffraction.F(x) for x in xs])
    ax = rng.uniform(-2 * np.pi, 4 * np.pi, 30)
    an = rng.uniform(1.0, 2.0, 30)
    out.update(A_x=ax, A_n=an, A_plus=np.array([em.Diffraction.A(x, n, "plus") for x, n in zip(ax, an)]),
               A_minus=np.array([em.Diffraction.A(x, n, "minus") for x, n in zip(ax, an)]))

    # wedge diffraction on the cube's vertical edge x = 10, y = 10 (faces 2: x = 10 and 3: y = 10)
    corner = np.array([[10.0, 10.0, -10.0], [10.0, 10.0, 10.0]])
    face0 = cube.polygons[2]

    def contains_point(self, point, check_in_plane=False, plane_tol=1e-8):
        # face x = 10 of the cube: y, z in [-10, 10]
        assert self is face0
        p = np.asarray(point).reshape(3)
        return bool(abs(p[0] - 10.0) <= plane_tol + 1e-5 * 10.0 and -10 <= p[1] <= 10 and -10 <= p[2] <= 10)

    geom.OrientedPolygon.contains_point = contains_point
    diff_paths, diff_D, diff_E, diff_Ei = [], [], [], []
    for _ in range(40):
        a = np.array([rng.uniform(12, 40), rng.uniform(-30, 9), rng.uniform(-8, 8)])
        b = np.array([rng.uniform(-30, 9), rng.uniform(12, 40), rng.uniform(-8, 8)])
        d, sol = geom.diffraction_point_from_origin_destination_and_edge(a, b, corner.copy())
        d = np.asarray(d).reshape(3)
        if not (-10 < d[2] < 10):
            continue
        path = np.array([a, d, b])
        antd = em.Antenna(a)
        antd.align_antenna_to_point(d)
        E_i = em.ElectromagneticField.from_path([a, d], antd)
        e = np.array([0.0, 0.0, 1.0])
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                comps = em.Diffraction.diff_dyadic_components(path, cube.polygons[2], cube.polygons[3], e)
                E = em.Diffraction.my_diff(E_i, path, (2, 3), corner, rtp, 0)
        except AssertionError:
            continue
        diff_paths.append(path)
        diff_D.append(comps)
        diff_E.append(E)
        diff_Ei.append(E_i)
    out.update(diff_paths=np.array(diff_paths), diff_D=np.array(diff_D), diff_E=np.array(diff_E),
               diff_Ei=np.array(diff_Ei), diff_corner=corner, diff_face0=cube.polygons[2].points,
               diff_face1=cube.polygons[3].points, diff_par0=cube.polygons[2].get_parametric(),
               diff_par1=cube.polygons[3].get_parametric())
    np.savez_compressed(os.path.join(HERE, "em.npz"), **out)
    print("em.npz written:", len(refl_paths), "reflections,", len(diff_paths), "diffractions")


if __name__ == "__main__":
    main()
