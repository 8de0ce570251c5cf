"""Field step with the reference's entry points (``electromagnetism_fun/electromagnetism.py``), computed on the GPU.

``my_field_computation(rtp, save_name)`` (:648-802) turns the paths of a solved ``RayTracingProblem`` into the
reference's result table (columns :600-607) with one call of the library's field kernel.  ``Antenna``,
``ElectromagneticField``, ``Reflection`` and ``Diffraction`` keep the reference's names; their per-path methods build a
one-path batch for the same kernel, so every number comes from the same device code.
"""
from math import pi

import numpy as np

from . import _lib
from . import electromagnetism_utils
from . import geometry as geom
from . import materials_properties as mat
from .electromagnetism_utils import vv_normalize

STATUS_MESSAGES = {
    1: "incident angle should be in 0,90 deg",
    2: "fresnel coeffs are greater than 1",
    3: "reflected field should be less powerful than incident field",
    4: "we restrict ourselves to wedges with interior angles in 0,180 deg",
    5: "beta_0 outside [0, pi]",
    6: "a diffracted ray cannot be inside the wedge (the RX or TX may be inside a building)",
    7: "edge is wrongly oriented",
    8: "diffracted field should be less powerful than incident field",
    9: "Cannot handle concave geometries",
    10: "spherical coordinates system not defined for point at the origin",
}


class Antenna:
    """Antenna frame: z towards the aimed point, x horizontal, polarisation along x (reference :50-148)."""

    def __init__(self, position):
        self.position = np.asarray(position, dtype=float)
        self.basis = None
        self.radiation_efficiency = 1
        self.polarisation = None
        self.mat = None

    def align_antenna_to_point(self, point):
        new_z = vv_normalize(np.asarray(point, dtype=float) - self.position)
        new_x = vv_normalize(np.cross(new_z, np.array([0, 0, 1])))
        new_y = vv_normalize(np.cross(new_x, new_z))
        self.basis = [new_x, new_y, new_z]
        self.polarisation = new_x
        self.mat = np.array([new_x, new_y, new_z], dtype=float)

    def pp_transform(self, point, direction):
        assert direction in ["A2W", "W2A"], "direction must be either 'A2W' or 'W2A'"
        if direction == "A2W":
            return self.mat.T @ point + self.position
        return (point - self.position) @ self.mat.T

    def vv_transform(self, vector, direction):
        assert direction in ["A2W", "W2A"], "direction must be either 'A2W' or 'W2A'"
        return vector @ self.mat if direction == "A2W" else vector @ self.mat.T

    def incident_angles(self, point):
        return electromagnetism_utils.to_spherical(self.pp_transform(np.asarray(point, dtype=float), "W2A"))

    def path_W2A(self, path):
        return [self.pp_transform(p, "W2A") for p in path]

    @staticmethod
    def compute_Ae(theta_rx, phi_rx=0):
        return (mat.LAMBDA ** 2 / (4 * pi)) * mat.RX_GAIN * ElectromagneticField.radiation_pattern(theta_rx, phi_rx)


class PathBatch:
    """Per-path description arrays the field kernel consumes."""

    def __init__(self):
        self.rx, self.kind, self.chain, self.edge_polys, self.edge_points, self.points, self.labels = [], [], [], [], [], [], []

    def add(self, rx, interaction_points, chain=(), edge_polys=None, edge_points=None):
        k = len(chain)
        has_edge = edge_polys is not None
        assert len(interaction_points) == k + (1 if has_edge else 0) <= _lib.RT_MAX_ORDER
        self.rx.append(rx)
        self.kind.append(k + (256 if has_edge else 0))
        self.chain.append(list(chain) + [-1] * (_lib.RT_MAX_ORDER - k))
        self.edge_polys.append(list(edge_polys) if has_edge else [-1, -1])
        self.edge_points.append(np.asarray(edge_points, dtype=float).reshape(6) if has_edge else np.zeros(6))
        pts = np.zeros((_lib.RT_MAX_ORDER, 3))
        if len(interaction_points):
            pts[:len(interaction_points)] = np.asarray(interaction_points, dtype=float).reshape(-1, 3)
        self.points.append(pts)
        self.labels.append("LOS" if k == 0 and not has_edge else "R" * k + ("D" if has_edge else ""))

    def __len__(self):
        return len(self.rx)

    def run(self, scene, tx, receivers, frequencies=None, **options):
        frequencies = [mat.FREQUENCY] if frequencies is None else frequencies
        params = dict(tx_gain=mat.TX_GAIN, rx_gain=mat.RX_GAIN, p_in=mat.P_IN, isotropic=mat.ISOTROPIC_ANTENNA,
                      alpha=mat.ALPHA)
        params.update(options)
        n = len(self)
        return scene.fields(tx, receivers, np.array(self.rx, dtype=np.int32), np.array(self.kind, dtype=np.int32),
                            np.array(self.chain, dtype=np.int32).reshape(n, _lib.RT_MAX_ORDER),
                            np.array(self.edge_polys, dtype=np.int32).reshape(n, 2),
                            np.array(self.edge_points, dtype=float).reshape(n, 6),
                            np.array(self.points, dtype=float).reshape(n, _lib.RT_MAX_ORDER * 3), frequencies, **params)


def _raise_on_status(status, labels=None):
    bad = np.flatnonzero(status)
    if len(bad):
        i = int(bad[0])
        what = f" (path {i}{', ' + labels[i] if labels else ''})"
        raise AssertionError(STATUS_MESSAGES.get(int(status[i]), f"status {int(status[i])}") + what)


def _helper_scene(polygons=None):
    """A device scene for calls that carry no problem: the given polygons or one far-away dummy polygon."""
    if polygons is None:
        far = geom.OrientedPolygon(np.array([[1e7, 1e7, -1e7], [1e7 + 1, 1e7, -1e7], [1e7, 1e7 + 1, -1e7]]))
        far.properties = mat.set_properties("road")
        far.building_id = "far"
        polygons = [far]
    return geom.scene_arrays_of(polygons).device_scene()


class ElectromagneticField:
    @staticmethod
    def radiation_pattern(theta, phi=0):
        F = np.cos(theta / 2) ** (2 * mat.ALPHA)
        return 1 if mat.ISOTROPIC_ANTENNA else F

    @staticmethod
    def path_loss(d):
        pr_pt = ((mat.LAMBDA / (4 * pi * d)) ** 2) * mat.RX_GAIN * mat.TX_GAIN * \
            (ElectromagneticField.radiation_pattern(theta=0, phi=0)) ** 2
        return 10 * np.log10(pr_pt)

    @staticmethod
    def from_path(path, tx_antenna):
        """Far field at path[1] of the aligned antenna at path[0], world frame (reference :156-176)."""
        assert len(path) == 2, "given path is strange"
        assert len(path[0]) == 3 and len(path[1]) == 3, "given path does not have the right shape"
        assert tx_antenna.basis is not None, "tx antenna is not aligned"
        a, b = np.asarray(path[0], dtype=float), np.asarray(path[1], dtype=float)
        batch = PathBatch()
        batch.add(0, [])
        aim = (tx_antenna.position + tx_antenna.basis[2]).reshape(1, 3)
        res = batch.run(_helper_scene(), tx_antenna.position, b.reshape(1, 3), aim_tx=aim, aim_rx=a.reshape(1, 3),
                        path_origin=a.reshape(1, 3))
        _raise_on_status(res["status"])
        return res["field"][0, 0]


def find_shortest_path(rtp, receiver):
    """Line of sight if there is one, else the shortest path, diffractions scanned before reflections (reference
    :610-644)."""
    if rtp.los[receiver] != []:
        return rtp.los[receiver][0]
    shortest, best = np.inf, None
    for container in (rtp.diffractions, rtp.reflections):
        for order in container[receiver]:
            for item in container[receiver][order]:
                length = geom.path_length(item[0])
                if length < shortest:
                    shortest, best = length, item[0]
    return best


def paths_of_problem(rtp):
    """All paths of a solved problem as a PathBatch, per receiver in the reference's row order: line of sight,
    reflections, diffractions (:776-778)."""
    n_rx = len(rtp.solved_receivers) if rtp.solved_receivers is not None else 0
    batch = PathBatch()
    lines = []
    for r in range(n_rx):
        for path in rtp.los[r][:1]:
            batch.add(r, [])
            lines.append(path)
        for order in rtp.reflections[r]:
            for path, chain in rtp.reflections[r][order]:
                batch.add(r, path[1:-1], chain)
                lines.append(path)
        for order in rtp.diffractions[r]:
            for path, chain, edge_polys, edge in rtp.diffractions[r][order]:
                batch.add(r, path[1:-1], chain, edge_polys, edge)
                lines.append(path)
    return batch, lines


TABLE_COLUMNS = ["rx_id", "receiver", "path_type", "tx_angles", "rx_angles", "time_to_receiver", "field_strength",
                 "path_power"]


def path_labels(kind):
    """The reference's path_type strings ('LOS', 'R' * k, 'R' * k + 'D') of the kernel's path kinds."""
    kind = np.asarray(kind, dtype=np.int64)
    names = {}
    out = []
    for k in kind.tolist():
        if k not in names:
            n_refl, has_edge = k & 255, k >> 8
            names[k] = "LOS" if k == 0 else "R" * n_refl + ("D" if has_edge else "")
        out.append(names[k])
    return out


def table_from_columns(n_rx, receivers, rx, labels, angles, time, field, power):
    """The reference's result table (electromagnetism.py:577-607) from per-path columns whose rows are already in its
    order (by receiver; line of sight, reflections, diffractions): one row per path, and one 'Impossible' row for every
    receiver without a path (:594-598).  Built column by column instead of row by row."""
    import pandas as pd

    rx = np.asarray(rx, dtype=np.int64)
    receivers = np.asarray(receivers, dtype=float).reshape(-1, 3)
    has_path = np.zeros(n_rx, dtype=bool)
    has_path[rx] = True
    missing = np.flatnonzero(~has_path)
    n, m = len(rx), len(missing)
    # where each row goes: the paths keep their order, a receiver without paths takes the place its index gives it
    all_rx = np.concatenate([rx, missing])
    place = np.argsort(all_rx, kind="stable")
    angles = np.asarray(angles, dtype=float).reshape(n, 4)
    field = np.asarray(field).reshape(n, 3)

    def column(values, filler):
        col = np.empty(n + m, dtype=object)
        col[:n] = values
        for k in range(n, n + m):
            col[k] = filler() if callable(filler) else filler
        return col[place]

    def rows_of(a):
        out = np.empty(len(a), dtype=object)
        for i, row in enumerate(a):  # (a slice assignment would try to make one 2-D array of the rows)
            out[i] = row
        return out

    rx_rows = rows_of(receivers[rx]) if n else np.empty(0, dtype=object)
    data = {
        "rx_id": all_rx[place],
        "receiver": None,
        "path_type": column(np.asarray(labels, dtype=object) if n else np.empty(0, dtype=object), "Impossible"),
        "tx_angles": column(rows_of(angles[:, 0:2].copy()), None),
        "rx_angles": column(rows_of(angles[:, 2:4].copy()), None),
        "time_to_receiver": column(np.asarray(time, dtype=float).astype(object) if n else np.empty(0, dtype=object), None),
        "field_strength": column(rows_of(field.copy()), lambda: np.array([0, 0, 0])),
        "path_power": column(np.asarray(power, dtype=float).astype(object) if n else np.empty(0, dtype=object), 0),
    }
    receiver_col = np.empty(n + m, dtype=object)
    receiver_col[:n] = rx_rows
    for k, r in enumerate(missing):
        receiver_col[n + k] = receivers[r]
    data["receiver"] = receiver_col[place]
    df = pd.DataFrame(data, columns=TABLE_COLUMNS)
    for name in ("time_to_receiver", "path_power"):  # plain numbers (None -> NaN), as a table made row by row has them
        df[name] = df[name].infer_objects()
    return df


def my_field_computation(rtp, save_name="problem_fields", frequency=None, save=True):
    """Field and power of every path of every solved receiver; returns the reference's DataFrame and writes the CSV.

    As in the reference the problem must have been solved with ``max_order >= 2`` (``solved_receivers`` is only set
    then, ray_tracing.py:352-353, and this function iterates over it, :763)."""
    import pandas as pd

    tx = np.asarray(rtp.emitter, dtype=float)[0]
    rtp.place.add_tree_crowns()
    if rtp.solved_receivers is None:
        raise TypeError("object of type 'NoneType' has no len(): solve(max_order >= 2) must run first")
    receivers = np.asarray(rtp.receivers, dtype=float).reshape(-1, 3)
    n_rx = len(rtp.solved_receivers)
    batch, lines = paths_of_problem(rtp)
    # the crowns come last in the place's polygon order, behind the polygons the problem was solved with
    all_polygons = rtp.place.get_polygons_list()
    n_late = len(all_polygons) - len(rtp.polygons)
    scene = geom.scene_arrays_of(all_polygons).device_scene() if n_late > 0 else rtp._device_scene()
    if len(batch):
        res = batch.run(scene, tx, receivers, [mat.FREQUENCY if frequency is None else frequency],
                        n_late_polygons=max(n_late, 0))
        _raise_on_status(res["status"], batch.labels)
        df = table_from_columns(n_rx, receivers, np.asarray(batch.rx, dtype=np.int64), batch.labels, res["angles"],
                                res["time"], res["field"][:, 0], res["power"][:, 0])
    else:
        df = table_from_columns(n_rx, receivers, np.zeros(0, dtype=np.int64), [], np.zeros((0, 4)), np.zeros(0),
                                np.zeros((0, 3), dtype=complex), np.zeros(0))
    if save:
        electromagnetism_utils.save_df(df, save_name)
    return df


class Reflection:
    @staticmethod
    def split_reflections_path(reflections_path):
        """[[TX, R1, P1], ..., [PN, RN, RX]] (reference :220-233)."""
        assert len(reflections_path) >= 3, f"wrong path shape given: {reflections_path}"
        return [reflections_path[i:i + 3] for i in range(len(reflections_path) - 2)]

    @staticmethod
    def my_reflect(E_i, reflections_path, surfaces_ids, rtp):
        """Field at the end of a chain of reflections given the field E_i arriving at the first one (reference
        :314-349)."""
        path = np.asarray(reflections_path, dtype=float)
        batch = PathBatch()
        batch.add(0, path[1:-1], [int(i) for i in surfaces_ids])
        res = batch.run(rtp._device_scene(), path[0], path[-1].reshape(1, 3), aim_tx=path[1].reshape(1, 3),
                        aim_rx=path[-2].reshape(1, 3), path_origin=path[0].reshape(1, 3),
                        path_e_in=np.asarray(E_i, dtype=complex).reshape(1, 3))
        _raise_on_status(res["status"])
        return res["field"][0, 0]


class Diffraction:
    @staticmethod
    def my_diff(E_i, diff_path, diff_surfaces_ids, corner_points, rtp, receiver=None):
        """Field after one wedge diffraction given the field E_i arriving at the edge (reference :453-507)."""
        path = np.asarray(diff_path, dtype=float)
        assert len(path) == 3
        batch = PathBatch()
        batch.add(0, path[1:2], [], [int(i) for i in diff_surfaces_ids], corner_points)
        res = batch.run(rtp._device_scene(), path[0], path[2].reshape(1, 3), aim_tx=path[1].reshape(1, 3),
                        aim_rx=path[1].reshape(1, 3), path_origin=path[0].reshape(1, 3),
                        path_e_in=np.asarray(E_i, dtype=complex).reshape(1, 3))
        _raise_on_status(res["status"])
        return res["field"][0, 0]
