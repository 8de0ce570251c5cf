"""CPU ORACLE — test infrastructure only.  Never imported by the product path.

Restatement of ``raytracing/ray_tracing.py`` of the reference (``RayTracingProblem``: precompute :201-237,
validity checks :243-279, depth-first candidate enumeration and solve :281-390).  Same call structure: one
SciPy hybr solve per (receiver, candidate), result gated on ``sol.success``, then the checks.

``counts`` records how many (receiver, candidate) solves were attempted per path type — the unit of the
path-solves/s metric.
"""
from collections import defaultdict

import numpy as np

from . import geometry_oracle as go
from . import visibility_oracle as vo


class RayTracingOracle:
    def __init__(self, emitter, polys, receivers, n_screens=6, visibility_matrix=None, sharp_edges=None,
                 precompute=True, solver="hybr"):
        assert 0 < n_screens <= 6
        assert solver in ("hybr", "exact")
        # "hybr": SciPy's root exactly as the reference calls it.  "exact": the roots of the same systems in closed
        # form (geometry_oracle.exact_*), i.e. what a solver that always finds the root when one exists returns.
        if solver == "hybr":
            self.solve_reflections = go.reflexion_points_from_origin_destination_and_planes
            self.solve_mixed = go.reflexion_points_and_diffraction_point_from_origin_destination_planes_and_edge
        else:
            self.solve_reflections = go.exact_reflexion_points
            self.solve_mixed = go.exact_reflexion_points_and_diffraction_point
        self.emitter = np.asarray(emitter, dtype=float)
        self.polys = list(polys)
        self.receivers = np.asarray(receivers, dtype=float).reshape(-1, 3)
        self.n_screens = n_screens
        self.visibility_matrix = visibility_matrix
        self.sharp_edges = sharp_edges  # list of ((i, j), edge) in pair order
        self.emitter_visibility = None
        n = len(self.receivers)
        self.los = {r: [] for r in range(n)}
        self.reflections = {r: defaultdict(list) for r in range(n)}
        self.diffractions = {r: defaultdict(list) for r in range(n)}
        self.solved_receivers = None
        self.counts = defaultdict(int)
        if precompute:
            self.precompute()

    # ray_tracing.py:201-237
    def precompute(self):
        if self.visibility_matrix is None:
            self.visibility_matrix = vo.polygons_visibility_matrix(self.polys)
        cube = go.cube_by_point_and_side_length(self.emitter, 2 * 0.1)
        emitter_visibilities = []
        for i in range(self.n_screens):
            # the reference passes out=None, i.e. an uninitialised vector that only ever receives True;
            # the evident intent (all-False start) is what is restated here
            ev = vo.polygon_visibility_vector(cube[i], self.polys, self_index=-1)
            emitter_visibilities.append((int(np.sum(ev)), ev))
        emitter_visibilities.sort(key=lambda x: x[0], reverse=True)
        self.emitter_visibility = emitter_visibilities[0][1].copy()
        for _, ev in emitter_visibilities[1:self.n_screens]:
            self.emitter_visibility |= ev
        if self.sharp_edges is None:
            self.sharp_edges = [((i, j), edge) for edge, i, j in go.polygons_sharp_edges(self.polys)]
        visible = [p for p, v in zip(self.polys, self.emitter_visibility) if v]
        for r, receiver in enumerate(self.receivers):
            line = np.vstack([self.emitter.reshape(3), receiver])
            self.counts["LOS"] += 1
            if not go.polygons_obstruct_line_path(visible, line):
                self.los[r].append(line)

    def get_visible_polygons_indices(self, index):
        return np.where(self.visibility_matrix[index, :])[0]

    # ray_tracing.py:243-274
    def check_reflections(self, lines, polygons_indices):
        min_distance_tol = 0.005
        for i, index in enumerate(polygons_indices):
            if not go.contains_point(self.polys[index], lines[i + 1, :], check_in_plane=True):
                return False
        if go.polygons_obstruct_line_path(self.polys, lines[:2, :]):
            return False
        for i, _ in enumerate(polygons_indices):
            if go.polygons_obstruct_line_path(self.polys, lines[i + 1:i + 3, :]):
                return False
        for i in range(len(lines) - 1):
            if np.linalg.norm(lines[i] - lines[i + 1]) < min_distance_tol:
                return False
        return True

    # ray_tracing.py:276-279
    def check_reflections_and_diffraction(self, lines, polygons_indices, edge):
        return self.check_reflections(lines[:-1, :], polygons_indices) and \
            go.point_on_edge_(lines[-2, :], edge) and \
            not go.polygons_obstruct_line_path(self.polys, lines[-2:, :])

    # ray_tracing.py:281-390
    def solve(self, max_order=2, receivers_indexs=None):
        emitter = self.emitter
        indices = np.where(self.emitter_visibility)[0]
        if receivers_indexs is not None:
            assert all(i < len(self.receivers) for i in receivers_indexs)
            rx_ids = list(receivers_indexs)
        else:
            rx_ids = list(range(len(self.receivers)))
        receivers = [self.receivers[i] for i in rx_ids]

        def recursive_reflections(polygons_indices, order):
            planes = [self.polys[i].parametric for i in polygons_indices]
            for r, receiver in zip(rx_ids, receivers):
                self.counts["R" * order] += 1
                points, sol = self.solve_reflections(emitter, receiver, planes)
                if not sol.success:
                    continue
                lines = np.vstack([emitter.reshape(1, 3), points, receiver.reshape(1, 3)])
                if self.check_reflections(lines, polygons_indices):
                    self.reflections[r][order].append((lines, polygons_indices))
            if order == max_order:
                return
            for i in self.get_visible_polygons_indices(polygons_indices[-1]):
                recursive_reflections(polygons_indices + [int(i)], order + 1)

        if max_order >= 1:
            for index in indices:
                recursive_reflections([int(index)], 1)
        if max_order < 1:
            return

        for (i, j), edge in self.sharp_edges:
            if self.emitter_visibility[i] or self.emitter_visibility[j]:
                for r, receiver in zip(rx_ids, receivers):
                    self.counts["D"] += 1
                    points, sol = self.solve_mixed(emitter, receiver, [], edge)
                    if not sol.success:
                        continue
                    lines = np.vstack([emitter.reshape(1, 3), points, receiver.reshape(1, 3)])
                    if self.check_reflections_and_diffraction(lines, [], edge):
                        self.diffractions[r][1].append((lines, [], (i, j), edge))

        if max_order < 2:
            return

        def edges_touching(visible):
            # container_utils.py:41-46: keys containing any of the visible polygons, in insertion order
            vis = set(int(v) for v in visible)
            return [(key, edge) for key, edge in self.sharp_edges if key[0] in vis or key[1] in vis]

        def recursive_reflections_and_diffraction(polygons_indices, order):
            planes = [self.polys[i].parametric for i in polygons_indices]
            visible = self.get_visible_polygons_indices(polygons_indices[-1])
            for edge_polygons, edge in edges_touching(visible):
                for r, receiver in zip(rx_ids, receivers):
                    self.counts["R" * (order - 1) + "D"] += 1
                    points, sol = self.solve_mixed(emitter, receiver, planes, edge)
                    if not sol.success:
                        continue
                    lines = np.vstack([emitter.reshape(1, 3), points, receiver.reshape(1, 3)])
                    if self.check_reflections_and_diffraction(lines, polygons_indices, edge):
                        self.diffractions[r][order].append((lines, polygons_indices, edge_polygons, edge))
            if order == max_order:
                return
            for i in self.get_visible_polygons_indices(polygons_indices[-1]):
                recursive_reflections_and_diffraction(polygons_indices + [int(i)], order + 1)

        for index in indices:
            recursive_reflections_and_diffraction([int(index)], 2)
        self.solved_receivers = receivers

    def path_set(self):
        """Set of (rx, kind, order, chain, edge key) — what 'the surviving path set' means in the parity tests."""
        out = set()
        for r, lst in self.los.items():
            for _ in lst:
                out.add((r, "LOS", 0, (), ()))
        for r, by_order in self.reflections.items():
            for order, lst in by_order.items():
                for _, chain in lst:
                    out.add((r, "R", order, tuple(int(c) for c in chain), ()))
        for r, by_order in self.diffractions.items():
            for order, lst in by_order.items():
                for item in lst:
                    out.add((r, "D", order, tuple(int(c) for c in item[1]), tuple(int(c) for c in item[2])))
        return out
