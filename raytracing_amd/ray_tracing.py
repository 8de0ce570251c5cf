"""``RayTracingProblem`` with the reference's construction and solve interface, computed on the GPU.

Mirrors ``raytracing/ray_tracing.py`` of the reference: constructor and attributes (:32-63), ``precompute`` (:201-237),
``solve`` (:281-390), ``to_json`` / ``from_json`` (:66-132).  The result containers have the reference's shapes:
    los[r]                   list of (2,3) arrays
    reflections[r][order]    list of (lines (order+2,3), [polygon indices])
    diffractions[r][order]   list of (lines (order+2,3), [polygon indices], (i, j), edge (2,3))
in the order the reference's depth-first loops produce them.
"""
import json
from collections import defaultdict

import numpy as np

from . import _lib
from . import geometry as geom


# Which candidates count as solved.  False (default): every candidate whose system has a root that passes the checks.
# True: only those for which the reference's own solver, SciPy's hybr as it calls it (geometry.py:682, 744, 823), would
# have reported that root — the reference loses a few valid paths to its solver, and with this switch on the solve
# loses the same ones (every found path is re-judged on the GPU by a restatement of that solver).  Read by
# ``RayTracingProblem.solve`` (and ``multithread_solve_place``) when they run; a problem's own ``reference_solver``
# attribute, when set, takes precedence.
REFERENCE_SOLVER = False


class RayTracingProblem:
    reference_solver = None

    def __init__(self, emitter, place, receivers=None, n_screens=6):
        assert 0 < n_screens <= 6
        self.emitter = emitter
        self.n_screens = n_screens
        self.place = place
        self.polygons = None
        self.receivers = np.array([])
        if self.place is not None:
            polygon_list = place.get_polygons_list()
            self.polygons = np.empty(len(polygon_list), dtype=object)
            for i, p in enumerate(polygon_list):
                self.polygons[i] = p
            self.receivers = self.place.set_of_points
        if receivers is not None:
            # as in the reference (:42-43) the extra receivers reach the place, not this problem's receiver list
            self.place.add_set_of_points(receivers)
        self.solved_receivers = None
        self.visibility_matrix = None
        self.distance_to_screen = None
        self.emitter_visibility = None
        self.sharp_edges = None
        self.path_solves_attempted = 0
        self._scene = None
        # paths handed to store_paths that have not been put into the containers yet (that happens when somebody reads
        # `reflections` or `diffractions`; to_json writes them from the arrays)
        self._pending = []
        self._edge_pairs = None
        self._edge_array = None
        n = self.receivers.shape[0]
        self.los = {r: list() for r in range(n)}
        self.reflections = {r: defaultdict(list) for r in range(n)}
        self.diffractions = {r: defaultdict(list) for r in range(n)}
        if self.place is not None:
            self.precompute()

    # ---- the result containers: the reference's dictionaries, filled when first looked at ---------------------
    @property
    def reflections(self):
        self._fill_containers()
        return self._reflections

    @reflections.setter
    def reflections(self, value):
        self._reflections = value

    @property
    def diffractions(self):
        self._fill_containers()
        return self._diffractions

    @diffractions.setter
    def diffractions(self, value):
        self._diffractions = value

    # ------------------------------------------------------------------------------------------------
    def _device_scene(self):
        if self._scene is None:
            self._scene = geom.scene_arrays_of(list(self.polygons)).device_scene()
        return self._scene

    def _solving_scene(self):
        """The device scene with the solver switch of this problem applied."""
        scene = self._device_scene()
        scene.set_reference_solver(REFERENCE_SOLVER if self.reference_solver is None else self.reference_solver)
        return scene

    def precompute(self):
        """Visibility matrix, emitter visibility through the screens of a 0.2 m cube, sharp edges, lines of sight."""
        scene = self._device_scene()
        emitter = np.asarray(self.emitter, dtype=float)
        cube = geom.Cube.by_point_and_side_length(emitter, 2 * 0.1)
        self.distance_to_screen = cube.polygons[0].distance_to_point(emitter)

        screens = [(cube.polygons[i].points, cube.polygons[i].get_parametric(), -1) for i in range(self.n_screens)]
        ahead = getattr(self.place, "_screens_ahead", None)  # painted beside the matrix rows by the sharded driver
        if ahead is not None and ahead[0] == (emitter.tobytes(), self.n_screens):
            screen_vis = ahead[1]
            self.visibility_matrix = self.place.get_visibility_matrix(strict=False)
        elif self.place.visibility_matrix is None:
            # the matrix and the screens in one call: the screens are painted while the matrix is (same results)
            self.place.visibility_matrix, screen_vis = scene.visibility_with_faces(screens)
            self.visibility_matrix = self.place.visibility_matrix
        else:
            self.visibility_matrix = self.place.get_visibility_matrix(strict=False)
            screen_vis = scene.faces_visibility(screens, False)
        visibilities = [(int(np.sum(vis)), vis) for vis in screen_vis]
        visibilities.sort(key=lambda x: x[0], reverse=True)
        self.emitter_visibility = visibilities[0][1].copy()
        for _, vis in visibilities[1:self.n_screens]:
            self.emitter_visibility |= vis

        self.sharp_edges = self.place.get_sharp_edges()
        keys = list(self.sharp_edges.keys())
        self._edge_pairs = np.array(keys, dtype=np.int32).reshape(-1, 2)
        self._edge_array = np.array([dict.__getitem__(self.sharp_edges, k) for k in keys], dtype=float).reshape(-1, 2, 3)

        if len(self.receivers) > 0:
            tx = emitter.reshape(3)
            segs = np.concatenate([np.broadcast_to(tx, (len(self.receivers), 3)), self.receivers], axis=1)
            blocked = scene.obstructed(segs, mask=self.emitter_visibility)
            for r, receiver in enumerate(self.receivers):
                if not blocked[r]:
                    self.los[r].append(np.vstack([emitter.reshape(1, 3), receiver.reshape(1, 3)]))

    def get_visible_polygons_indices(self, index):
        return np.where(self.visibility_matrix[index, :])[0]

    # ------------------------------------------------------------------------------------------------
    def check_reflections(self, lines, polygons_indices):
        """Validity of one reflection path (reference :243-274)."""
        scene = self._device_scene()
        lines = np.asarray(lines, dtype=float)
        for i, index in enumerate(polygons_indices):
            if not scene.contains_points(index, lines[i + 1], True, 1e-8)[0]:
                return False
        segs = np.concatenate([lines[:-1], lines[1:]], axis=1)[:len(polygons_indices) + 1]
        if scene.obstructed(segs).any():
            return False
        return bool((np.linalg.norm(np.diff(lines, axis=0), axis=1) >= 0.005).all())

    def check_reflections_and_diffraction(self, lines, polygons_indices, edge):
        return self.check_reflections(lines[:-1, :], polygons_indices) and geom.point_on_edge_(lines[-2, :], edge) \
            and not self._device_scene().obstructed(np.concatenate([lines[-2], lines[-1]]).reshape(1, 6))[0]

    # ------------------------------------------------------------------------------------------------
    def solve(self, max_order=2, receivers_indexs=None):
        """Finds every reflection / diffraction path up to ``max_order`` for all (or the listed) receivers.
        Results are appended to ``reflections`` / ``diffractions`` (calling twice appends twice, as the reference)."""
        emitter = np.asarray(self.emitter, dtype=float)
        if receivers_indexs is not None:
            assert all(i < len(self.receivers) for i in receivers_indexs), \
                f"provided RX indexs {receivers_indexs} but only {len(self.receivers)} RX exist"
            rx_ids = [int(i) for i in receivers_indexs]
            receivers = [self.receivers[i] for i in rx_ids]
        else:
            rx_ids = list(range(len(self.receivers)))
            receivers = self.receivers
        if max_order < 1 or len(rx_ids) == 0:
            if max_order >= 2:
                self.solved_receivers = receivers
            return

        scene = self._solving_scene()
        cand = scene.enumerate(self.visibility_matrix, self.emitter_visibility, self._edge_pairs, self._edge_array,
                               max_order, with_diffraction=True)
        rx = np.asarray([self.receivers[i] for i in rx_ids], dtype=float).reshape(-1, 3)
        res = scene.solve_paths(cand, emitter.reshape(3), rx)
        self.path_solves_attempted += res["n_attempted"]
        self.store_paths(np.asarray(rx_ids, dtype=np.int64)[res["rx"]], res["group"], res["cand"], res["points"], cand)
        cand.close()
        if max_order >= 2:
            self.solved_receivers = receivers  # the reference only gets here for max_order >= 2 (:352-353, 390)

    def store_paths(self, rx, group, cand_index, points, cand):
        """Takes solved paths for ``reflections`` / ``diffractions`` (ray_tracing.py:314, 350, 376).  ``rx``: receiver
        index of each path, ``group`` / ``cand_index``: its candidate (lines of sight, group -1, are skipped:
        precompute stored them), ``points`` [P, RT_MAX_ORDER, 3]; the paths must come sorted by (receiver, group,
        candidate), which is the reference's visiting order.  Only arrays are kept here: a coverage map has hundreds
        of thousands of paths, and the per-path tuples of the reference's containers are made when the containers
        are first read (``to_json`` writes the file from the arrays)."""
        rx = np.asarray(rx, dtype=np.int64)
        group = np.asarray(group, dtype=np.int64)
        keep = group >= 0
        rx, group = rx[keep], group[keep]
        cand_index = np.asarray(cand_index, dtype=np.int64)[keep]
        points = np.asarray(points, dtype=float).reshape(len(keep), _lib.RT_MAX_ORDER, 3)[keep]
        if len(rx) == 0:
            return
        kind, chain, edge_polys, edge_points = cand.describe(group, cand_index)
        n_refl = (kind & 255).astype(np.int32)
        has_edge = (kind >> 8) != 0
        self._pending.append(dict(rx=rx, order=(n_refl + has_edge).astype(np.int32), n_refl=n_refl, has_edge=has_edge,
                                  points=points, chain=chain.astype(np.int32), edge_key=edge_polys.astype(np.int32),
                                  edge_points=edge_points.reshape(-1, 2, 3)))

    def _lines_of(self, batch, sel):
        """[len(sel), RT_MAX_ORDER + 2, 3]: emitter, interaction points, receiver of the listed paths of a batch (a
        path of n points uses the first n + 2 rows)."""
        n_pts = batch["order"][sel]
        lines = np.zeros((len(sel), _lib.RT_MAX_ORDER + 2, 3))
        lines[:, 0] = np.asarray(self.emitter, dtype=float).reshape(1, 3)
        lines[:, 1:_lib.RT_MAX_ORDER + 1] = batch["points"][sel]
        receivers = np.asarray(self.receivers, dtype=float).reshape(-1, 3)
        lines[np.arange(len(sel)), n_pts + 1] = receivers[batch["rx"][sel]]
        return lines, n_pts

    def _fill_containers(self):
        """Puts the paths store_paths has taken into the dictionaries, in the reference's item shapes: the point
        arrays of all paths of a kind are put together at once; per path only the tuple is made."""
        pending, self._pending = self._pending, []
        for batch in pending:
            everything = np.arange(len(batch["rx"]))
            lines, n_pts = self._lines_of(batch, everything)
            chains = batch["chain"].tolist()
            keys = [tuple(k) for k in batch["edge_key"].tolist()]
            rx, order, n_refl, has_edge = batch["rx"].tolist(), batch["order"].tolist(), batch["n_refl"].tolist(), batch["has_edge"]
            for i in range(len(rx)):
                path = lines[i, :n_pts[i] + 2]
                if has_edge[i]:
                    # (a copy of the edge per path, as the reference's containers hold)
                    item = (path, chains[i][:n_refl[i]], keys[i], batch["edge_points"][i].copy())
                    self._diffractions[rx[i]][order[i]].append(item)
                else:
                    self._reflections[rx[i]][order[i]].append((path, chains[i][:n_refl[i]]))

    # ------------------------------------------------------------------------------------------------
    def path_set(self):
        """Set of (rx, kind, order, polygon chain, edge key): the surviving path set."""
        out = set()
        for r, lst in self.los.items():
            for _ in lst:
                out.add((r, "LOS", 0, (), ()))
        for r, by_order in self.reflections.items():
            for order, lst in by_order.items():
                for _, chain in lst:
                    out.add((r, "R", order, tuple(chain), ()))
        for r, by_order in self.diffractions.items():
            for order, lst in by_order.items():
                for item in lst:
                    out.add((r, "D", order, tuple(item[1]), tuple(item[2])))
        return out

    def __str__(self):
        return (f"RayTracingProblem object containing: {self.place}, emitter: {self.emitter}, "
                f"number of solved_receivers: {len(self.solved_receivers) if self.solved_receivers is not None else 0}, "
                f"los: {len(self.los)}, reflections: {len(self.reflections)}, diffractions : {len(self.diffractions)}")

    # ---- on-disk format of the reference (:66-132) -------------------------------------------------------
    def __eq__(self, other):
        """The reference's comparison (ray_tracing.py:146-200): emitter, screens, emitter visibility, receivers, solved
        receivers, screen distance, sharp edges, and the receiver keys of the three result dictionaries (it does not
        look inside them)."""
        if not isinstance(other, RayTracingProblem):
            return False
        if not np.array_equal(self.emitter, other.emitter) or self.n_screens != other.n_screens:
            return False
        if not np.array_equal(self.emitter_visibility, other.emitter_visibility):
            return False
        if not np.array_equal(self.receivers, other.receivers):
            return False
        if not np.array_equal(np.asarray(self.solved_receivers, dtype=object if self.solved_receivers is None else float),
                              np.asarray(other.solved_receivers, dtype=object if other.solved_receivers is None else float)):
            return False
        if self.distance_to_screen != other.distance_to_screen or not self.sharp_edges == other.sharp_edges:
            return False
        return all(all(a == b for a, b in zip(mine, theirs)) for mine, theirs in
                   ((self.los, other.los), (self.reflections, other.reflections), (self.diffractions, other.diffractions)))

    __hash__ = object.__hash__

    def to_json(self, filename):
        def plain(obj):
            """Nested containers of arrays as nested lists (one C-level tolist() per array)."""
            if isinstance(obj, np.ndarray):
                return obj.tolist()
            if isinstance(obj, dict):
                return {k: plain(v) for k, v in obj.items()}
            if isinstance(obj, (list, tuple)):
                return [plain(v) for v in obj]
            if isinstance(obj, np.integer):
                return int(obj)
            if isinstance(obj, np.floating):
                return float(obj)
            return obj

        # The two path containers are most of the file.  When the paths are still the arrays of one store_paths call
        # and the dictionaries hold nothing else, their text is written from the arrays (the same characters json.dumps
        # gives for the dictionaries, tests/test_paths_json.py); otherwise the dictionaries are walked.
        from_arrays = None
        if len(self._pending) == 1 and not any(by_order for c in (self._reflections, self._diffractions)
                                               for by_order in c.values()):
            batch = self._pending[0]
            from_arrays = {}
            for name, container, is_diff in (("reflections", self._reflections, False), ("diffractions", self._diffractions, True)):
                sel = np.flatnonzero(batch["has_edge"] == is_diff)
                # the encoder takes the paths grouped by receiver in the order of the container's keys (ascending) and by
                # order within a receiver: a solve of receivers listed out of order (solve(receivers_indexs=[2, 0])) or
                # twice is not, and goes the dictionaries' way
                key = batch["rx"][sel] * (2 * _lib.RT_MAX_ORDER + 2) + batch["order"][sel]
                if len(key) > 1 and (np.diff(key) < 0).any():
                    from_arrays = None
                    break
                lines, n_pts = self._lines_of(batch, sel)
                from_arrays[name] = _lib.format_paths_json(
                    list(container.keys()), batch["rx"][sel], batch["order"][sel], n_pts + 2, lines, batch["n_refl"][sel],
                    batch["chain"][sel], batch["edge_key"][sel] if is_diff else None,
                    batch["edge_points"][sel] if is_diff else None)
        data = {
            "place": self.place.to_json(),
            "receivers": plain(self.receivers),
            "solved_receivers": plain(self.solved_receivers),
            "emitter": plain(self.emitter),
            "n_screens": self.n_screens,
            "los": plain(self.los),
        }
        if from_arrays is None:
            data["reflections"] = plain(self.reflections)
            data["diffractions"] = plain(self.diffractions)
        text = json.dumps(data)  # one pass of the C encoder (json.dump would iterate in Python)
        if from_arrays is not None:
            text = f'{text[:-1]}, "reflections": {from_arrays["reflections"]}, "diffractions": {from_arrays["diffractions"]}}}'
        filename = filename if ".json" in filename else f"{filename}.json"
        with open(filename, "w") as f:
            f.write(text)

    @staticmethod
    def from_json(filename):
        with open(filename) as f:
            data = json.load(f)
        place = geom.OrientedPlace.from_json(data=data["place"])
        problem = RayTracingProblem(np.array(data["emitter"]), place, n_screens=data["n_screens"])
        solved = data.get("solved_receivers")
        problem.solved_receivers = [np.array(s) for s in solved] if solved is not None else None
        problem.reflections = {int(r): defaultdict(list) for r in data["reflections"]}
        for r, by_order in data["reflections"].items():
            for k, items in by_order.items():
                for points, indices in items:
                    problem.reflections[int(r)][int(k)].append((np.array(points), indices))
        problem.diffractions = {int(r): defaultdict(list) for r in data["diffractions"]}
        for r, by_order in data["diffractions"].items():
            for k, items in by_order.items():
                for points, chain, indices, edge_points in items:
                    problem.diffractions[int(r)][int(k)].append((np.array(points), chain, tuple(indices),
                                                                 np.array(edge_points)))
        return problem
