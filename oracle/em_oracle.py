"""CPU ORACLE — test infrastructure only.  Never imported by the product path.

Restatement of the field step of the reference: ``electromagnetism_fun/electromagnetism.py`` (Antenna :50-148,
ElectromagneticField :154-215, Reflection :218-349, Diffraction :353-507, find_shortest_path :610-644,
my_field_computation :648-802) and ``electromagnetism_utils.py`` (to_spherical :146-161, vv_normalize / cot
:170-177, to_db :180-188).  SciPy's ``fresnel`` is called as the reference calls it.  The carrier frequency is a
parameter here (the reference fixes it in ``materials_properties.py:23``); with the default it is the reference's.

Works on the oracle's containers: ``polys`` (list of geometry_oracle.Poly with ``material``), and the ``los`` /
``reflections`` / ``diffractions`` dictionaries of ``ray_tracing_oracle.RayTracingOracle``.
"""
from math import pi

import numpy as np
import scipy.special
from scipy.constants import c, epsilon_0

from . import geometry_oracle as go

Z_0 = 120 * pi                       # materials_properties.py:29
P_IN = 1                             # materials_properties.py:32
TX_GAIN, RX_GAIN = 1, 1              # materials_properties.py:41
ISOTROPIC_ANTENNA = True             # materials_properties.py:36
ALPHA = 10                           # materials_properties.py:37


def material_properties(material, frequency):
    """materials_properties.py:47-109: (epsilon_eff complex, roughness) of a material at a carrier frequency."""
    f = frequency / 1e9
    table = {
        "concrete": (5.24, 0.0462 * f ** 0.7822, 1e-4),
        "brick": (3.91, 0.0238 * f ** 0.16, 1e-3),
        "wood": (1.99, 0.0047 * f ** 1.0718, 1e-4),
        "glass": (6.31, 0.0036 * f ** 1.3394, 1e-6),
        "metal": (1, 1e7, 1e-6),
        "medium_dry_ground": (15 * f ** (-0.1), 0.035 * f ** 1.63, 1e-2),
    }
    eps, sigma, rough = table[material]
    return eps - 1j * sigma / (epsilon_0 * 2 * pi * frequency), rough


def vv_normalize(vv):
    """electromagnetism_utils.py:173-177"""
    n = np.linalg.norm(vv)
    if n == 0:
        return np.array([0, 0, 0])
    return vv / n


def cot(x):
    return 1 / np.tan(x)


def to_spherical(point):
    """electromagnetism_utils.py:146-161"""
    x, y, z = point
    assert not np.all(point == 0), "spherical coordinates system not defined for point at the origin."
    if x == 0 and y == 0:
        return z, 0, 0
    hxy = np.sqrt(x ** 2 + y ** 2)
    r = np.sqrt(hxy ** 2 + z ** 2)
    el = np.arctan2(hxy, z)
    az = np.arctan2(y, x)
    return r, el, az


def to_db(field_power):
    """electromagnetism_utils.py:180-188 (np.NINF spelled -inf)"""
    P = np.atleast_1d(field_power).astype(float)
    db = np.full(P.shape, -np.inf)
    ind = P != 0
    db[ind] = 10 * np.log10(P[ind] / P_IN)
    return db[0] if np.isscalar(field_power) else db


class Antenna:
    """electromagnetism.py:50-148"""

    def __init__(self, position):
        self.position = position
        self.basis = None
        self.polarisation = None
        self.mat = None

    def align_antenna_to_point(self, point):
        new_z = vv_normalize(point - self.position)
        new_x = vv_normalize(np.cross(new_z, np.array([0, 0, 1])))
        new_y = vv_normalize(np.cross(new_x, new_z))
        self.basis = [new_x, new_y, new_z]
        self.polarisation = new_x
        x, y, z = np.eye(3)
        self.mat = np.array([[np.dot(new_x, x), np.dot(new_x, y), np.dot(new_x, z)],
                             [np.dot(new_y, x), np.dot(new_y, y), np.dot(new_y, z)],
                             [np.dot(new_z, x), np.dot(new_z, y), np.dot(new_z, z)]])

    def pp_transform(self, point, direction):
        if direction == "A2W":
            return self.mat.T @ point + self.position
        return (point - self.position) @ (self.mat.T)

    def vv_transform(self, vector, direction):
        return vector @ self.mat if direction == "A2W" else vector @ (self.mat.T)

    def incident_angles(self, point):
        return to_spherical(self.pp_transform(point, "W2A"))


class FieldOracle:
    def __init__(self, polys, frequency=12.5e9):
        self.polys = polys
        self.frequency = frequency
        self.lam = c / frequency
        self.K = 2 * pi / self.lam

    # ---- electromagnetism.py:154-215 ---------------------------------------------------------------
    @staticmethod
    def radiation_pattern(theta, phi=0):
        F = np.cos(theta / 2) ** (2 * ALPHA)
        return 1 if ISOTROPIC_ANTENNA else F

    def from_path(self, path, tx_antenna):
        r, theta, phi = tx_antenna.incident_angles(path[1])
        r_vv = vv_normalize(tx_antenna.vv_transform(path[1] - path[0], "W2A"))
        E = -1j * self.K * Z_0 / (4 * pi) * np.exp(-1j * self.K * r) / r * self.radiation_pattern(theta, phi)
        E = E * vv_normalize(np.cross(np.cross(r_vv, tx_antenna.vv_transform(tx_antenna.polarisation, "W2A")), r_vv))
        return tx_antenna.vv_transform(E, "A2W")

    def account_for_trees(self, E, points):
        """electromagnetism.py:179-199 with geometry.py:868-921: whatever the segment crosses is treated as a tree."""
        hits = go.polygons_obstruct_line_path(self.polys, np.asarray(points), return_indices=True)
        groups, pol = [], []
        for idx in hits:  # geometry.py:886-895
            if len(pol) == 0:
                pol.append(idx)
            elif self.polys[pol[0]].building == self.polys[idx].building:
                pol.append(idx)
            else:
                groups.append(pol)
                pol = []
        if len(pol) > 0:
            groups.append(pol)
        for g in groups:
            if len(g) == 1:
                continue
            assert len(g) == 2, "Cannot handle concave geometries"
            pts_in = self._line_plane(self.polys[g[0]], points)
            pts_out = self._line_plane(self.polys[g[1]], points)
            d = np.linalg.norm(pts_in - pts_out)
            if d > 0:
                L_db = 0.39 * ((self.frequency / 1e6) ** 0.39) * d ** 0.25
                E = E * 10 ** (-L_db / 10)
        return E

    @staticmethod
    def _line_plane(poly, points):
        """geometry.py:89-106"""
        normal = poly.normal
        plane_point = poly.points[1, :]
        direction = points[1] - points[0]
        direction = direction / np.linalg.norm(direction)
        ndotu = np.dot(direction, normal)
        w = points[0] - plane_point
        si = -np.dot(w, normal) / ndotu
        return w + si * direction + plane_point

    # ---- electromagnetism.py:218-349 ---------------------------------------------------------------
    def fresnel_coeffs(self, theta_i, poly):
        assert 0 < theta_i < pi / 2
        eps2, roughness = material_properties(poly.material, self.frequency)
        sqrt = np.sqrt(eps2 - (np.sin(theta_i)) ** 2)
        cosi = np.cos(theta_i)
        r_par = abs((eps2 * cosi - sqrt) / (eps2 * cosi + sqrt))
        r_per = abs((cosi - sqrt) / (cosi + sqrt))
        assert r_par <= 1 and r_per <= 1
        if roughness > self.lam / (8 * cosi):  # :236-244
            chi = np.exp(-2 * (self.K ** 2) * (roughness ** 2) * cosi ** 2)
        else:
            chi = 1
        return r_par * chi, r_per * chi

    def dyadic_ref_coeff(self, path, poly):
        si, _, sr, _ = self._normalize_path(path)
        theta_i = np.arccos(np.dot(poly.normal, -si))
        assert 0 < theta_i < pi / 2, f"incident angle {theta_i}"
        r_par, r_per = self.fresnel_coeffs(theta_i, poly)
        e_per = vv_normalize(np.cross(si, sr))
        ei_par = vv_normalize(np.cross(e_per, si))
        er_par = vv_normalize(np.cross(e_per, sr))
        return e_per.reshape(3, 1) @ e_per.reshape(1, 3) * r_per + ei_par.reshape(3, 1) @ er_par.reshape(1, 3) * r_par

    @staticmethod
    def _normalize_path(points):
        """geometry.py:79-87"""
        vectors = np.diff(points, axis=0)
        norms = np.linalg.norm(vectors, axis=1)
        return vectors[0] / norms[0], norms[0], vectors[1] / norms[1], norms[1]

    def my_reflect(self, E_i, reflections_path, surfaces_ids):
        E_r = E_i
        for ind, surf_id in enumerate(surfaces_ids):
            the_path = reflections_path[ind:ind + 3]
            _, Si, _, Sr = self._normalize_path(the_path)
            R = self.dyadic_ref_coeff(the_path, self.polys[surf_id])
            E_r = self.account_for_trees(E_r, np.array([the_path[0], the_path[1]]))  # electromagnetism.py:338
            field = (E_r.reshape(1, 3) @ R) * np.exp(-1j * self.K * Sr) * (Si / (Si + Sr))
            E_r = field.reshape(3,)
            assert np.linalg.norm(E_i) >= np.linalg.norm(field), "reflected field should be less powerful"
        last = reflections_path[len(surfaces_ids) - 1:len(surfaces_ids) + 2]
        return self.account_for_trees(E_r, np.array([last[1], last[2]]))  # electromagnetism.py:346-348

    # ---- electromagnetism.py:353-507 ---------------------------------------------------------------
    @staticmethod
    def F(x):
        factor = np.sqrt(np.pi / 2)
        sqrtx = np.sqrt(x)
        S, C = scipy.special.fresnel(sqrtx / factor)
        return 2j * sqrtx * np.exp(1j * x) * (factor * ((1 - 1j) / 2 - C + 1j * S))

    @staticmethod
    def A(x, n, sign):
        if sign == "minus":
            N = round((-pi + x) / (2 * n * pi))
        else:
            N = round((pi + x) / (2 * n * pi))
        return 2 * (np.cos(n * pi * N - x / 2)) ** 2

    def diff_dyadic_components(self, diff_path, polygon1, polygon2, e):
        si, Si, sd, Sd = self._normalize_path(diff_path)
        n_0 = polygon1.normal
        n_1 = polygon2.normal
        t_0 = vv_normalize(np.cross(n_0, e))
        alpha = pi - np.arccos(np.dot(n_0, n_1))
        n = 2 - alpha / pi
        assert 0 < alpha < pi and 1 <= n <= 2
        beta_0 = np.arccos(np.dot(si, e))
        assert 0 <= beta_0 <= pi
        si_t = vv_normalize(si - np.dot(si, e) * e)
        sd_t = vv_normalize(sd - np.dot(sd, e) * e)
        phi_i = pi - (pi - np.arccos(np.dot(-si_t, t_0))) * np.sign(np.dot(-si_t, n_0))
        phi_d = pi - (pi - np.arccos(np.dot(sd_t, t_0))) * np.sign(np.dot(sd_t, n_0))
        phi_i = phi_i % (2 * pi)
        phi_d = phi_d % (2 * pi)
        assert 0 <= phi_i <= n * pi, "phi_i inside the wedge"
        assert 0 <= phi_d <= n * pi, "phi_d inside the wedge"
        common_factor = -np.exp(-1j * pi / 4) / (2 * n * np.sqrt(2 * pi * self.K) * np.sin(beta_0))
        L = ((np.sin(beta_0)) ** 2) * Si * Sd / (Si + Sd)
        cot1 = cot((pi + (phi_d - phi_i)) * 1 / (2 * n))
        cot2 = cot((pi - (phi_d - phi_i)) * 1 / (2 * n))
        cot3 = cot((pi + (phi_d + phi_i)) * 1 / (2 * n))
        cot4 = cot((pi - (phi_d + phi_i)) * 1 / (2 * n))
        F1 = self.F(self.K * L * self.A(phi_d - phi_i, n, "plus"))
        F2 = self.F(self.K * L * self.A(phi_d - phi_i, n, "minus"))
        F3 = self.F(self.K * L * self.A(phi_d + phi_i, n, "plus"))
        F4 = self.F(self.K * L * self.A(phi_d + phi_i, n, "minus"))
        return common_factor * cot1 * F1, common_factor * cot2 * F2, common_factor * cot3 * F3, common_factor * cot4 * F4

    def my_diff(self, E_i, diff_path, diff_surfaces_ids, corner_points):
        si, Si, sd, Sd = self._normalize_path(diff_path)
        polygon_1 = self.polys[diff_surfaces_ids[0]]
        polygon_2 = self.polys[diff_surfaces_ids[1]]
        n_0 = polygon_1.normal
        e = vv_normalize(np.diff(corner_points, axis=0).reshape(-1))
        t_0 = vv_normalize(np.cross(n_0, e))
        mid_edge = (corner_points[0] + corner_points[1]) / 2
        if not go.contains_point(polygon_1, mid_edge + t_0 * 1e-6, check_in_plane=True, plane_tol=1e-8):
            e = -e
            t_0 = vv_normalize(np.cross(n_0, e))
        assert go.contains_point(polygon_1, mid_edge + t_0 * 1e-6, check_in_plane=True, plane_tol=1e-8), \
            "edge is wrongly oriented"
        D1, D2, D3, D4 = self.diff_dyadic_components(diff_path, polygon_1, polygon_2, e)
        D_per = D1 + D2 + (D3 + D4)
        D_par = D1 + D2 + (D3 + D4)
        phi_i_vv = vv_normalize(np.cross(-e, si))
        phi_d_vv = vv_normalize(np.cross(e, sd))
        beta_0_i_vv = vv_normalize(np.cross(phi_i_vv, si))
        beta_0_d_vv = vv_normalize(np.cross(phi_d_vv, sd))
        D = beta_0_i_vv.reshape(3, 1) @ beta_0_d_vv.reshape(1, 3) * (-D_per) - \
            phi_i_vv.reshape(3, 1) @ phi_d_vv.reshape(1, 3) * D_par
        field = (E_i.reshape(1, 3) @ D) * np.sqrt(Si / (Sd * (Sd + Si))) * np.exp(-1j * self.K * Sd)
        assert np.linalg.norm(E_i) >= np.linalg.norm(field), "diffracted field should be less powerful"
        return self.account_for_trees(field.reshape(3,), np.array([diff_path[1], diff_path[2]]))  # :506

    # ---- electromagnetism.py:610-644 ---------------------------------------------------------------
    @staticmethod
    def find_shortest_path(rt, receiver):
        if rt.los[receiver] != []:
            return rt.los[receiver][0]
        shortest, best = np.inf, None
        for order in rt.diffractions[receiver]:
            for item in rt.diffractions[receiver][order]:
                length = np.sum(np.linalg.norm(np.diff(item[0], axis=0), axis=1))
                if length < shortest:
                    shortest, best = length, item[0]
        for order in rt.reflections[receiver]:
            for item in rt.reflections[receiver][order]:
                length = np.sum(np.linalg.norm(np.diff(item[0], axis=0), axis=1))
                if length < shortest:
                    shortest, best = length, item[0]
        return best

    # ---- electromagnetism.py:648-802 ---------------------------------------------------------------
    def field_computation(self, rt, receivers):
        """Rows (dict) in the reference's order: per receiver LOS, reflections (orders as stored), diffractions."""
        tx = np.asarray(rt.emitter, dtype=float).reshape(3)
        rows = []
        for receiver in range(len(receivers)):
            tx_antenna = Antenna(tx)
            rx_antenna = Antenna(receivers[receiver])
            best_path = self.find_shortest_path(rt, receiver)
            if best_path is None:
                rows.append(dict(rx_id=receiver, receiver=receivers[receiver], path_type="Impossible", tx_angles=None,
                                 rx_angles=None, time_to_receiver=None, field_strength=np.array([0, 0, 0]),
                                 path_power=0))
                continue
            tx_antenna.align_antenna_to_point(best_path[1])
            rx_antenna.align_antenna_to_point(best_path[-2])
            sols = []
            if rt.los[receiver] != []:
                path = rt.los[receiver][0]
                E = self.from_path(path, tx_antenna)
                E = self.account_for_trees(E, path)
                sols.append(("LOS", path, E))
            for order in rt.reflections[receiver]:
                for the_path, ids in rt.reflections[receiver][order]:
                    E_i = self.from_path([tx, the_path[1]], tx_antenna)
                    sols.append(("R" * order, the_path, self.my_reflect(E_i, the_path, ids)))
            for order in rt.diffractions[receiver]:
                for the_path, ids, diff_ids, corner_points in rt.diffractions[receiver][order]:
                    E_0 = self.from_path([tx, the_path[1]], tx_antenna)
                    if len(the_path) == 3:  # electromagnetism.py:724-725
                        E_i = self.account_for_trees(E_0, np.array([the_path[0], the_path[1]]))
                    else:
                        E_i = self.my_reflect(E_0, the_path[0:-1], ids)
                    E = self.my_diff(E_i, the_path[-3:], diff_ids, corner_points)
                    sols.append(("R" * (order - 1) + "D", the_path, E))
            for path_type, path, field in sols:
                _, tx_el, tx_az = tx_antenna.incident_angles(path[1])
                _, rx_el, rx_az = rx_antenna.incident_angles(path[-2])
                Ae = (self.lam ** 2 / (4 * pi)) * RX_GAIN * self.radiation_pattern(rx_el, rx_az)
                field_A = rx_antenna.vv_transform(field, "W2A")
                field_polarisation = vv_normalize(field_A)
                pol_eff = (np.linalg.norm(np.dot(field_polarisation,
                                                 rx_antenna.vv_transform(rx_antenna.polarisation, "W2A")))) ** 2
                power = 1 / (2 * Z_0) * Ae * (np.linalg.norm(np.real(field_A))) ** 2
                power = power * pol_eff * TX_GAIN
                length = np.sum(np.linalg.norm(np.diff(path, axis=0), axis=1))
                rows.append(dict(rx_id=receiver, receiver=path[-1], path_type=path_type,
                                 tx_angles=np.degrees([tx_el, tx_az]), rx_angles=np.degrees([rx_el, rx_az]),
                                 time_to_receiver=length / c, field_strength=field, path_power=power))
        return rows


def power_db_each_receiver(rows, n_receivers):
    """electromagnetism_utils.py:104-115"""
    total = np.zeros(n_receivers)
    for row in rows:
        total[row["rx_id"]] += row["path_power"]
    return to_db(total)
