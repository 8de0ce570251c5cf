"""ctypes binding of ``libraytracing_amd.so`` (C ABI in ``include/rt_pathsolve.h``).

There is no CPU implementation behind these calls: a missing library or a missing GPU raises.
"""
import ctypes
import os

import numpy as np

RT_MAX_ORDER = 4
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libraytracing_amd.so")
_lib = None

c_void_p, c_int, c_int32, c_int64, c_double = ctypes.c_void_p, ctypes.c_int, ctypes.c_int32, ctypes.c_int64, ctypes.c_double


class RtError(RuntimeError):
    pass


class FieldOptions(ctypes.Structure):
    _fields_ = [("aim_tx", c_void_p), ("aim_rx", c_void_p), ("path_origin", c_void_p), ("path_e_in", c_void_p),
                ("n_late_polygons", c_int32)]


class AntennaParams(ctypes.Structure):
    _fields_ = [("tx_gain", c_double), ("rx_gain", c_double), ("p_in", c_double), ("isotropic", c_int32),
                ("alpha", c_int32)]


_SIGNATURES = {
    "rt_last_error": (ctypes.c_char_p, []),
    "rt_device_count": (c_int, []),
    "rt_scene_create": (c_void_p, [c_int, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rt_scene_destroy": (None, [c_void_p]),
    "rt_scene_polygon_count": (c_int32, [c_void_p]),
    "rt_obstruct_segments": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int]),
    "rt_obstruct_segments_noalloc": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "rt_segment_hits": (c_int, [c_void_p, c_void_p, c_void_p]),
    "rt_contains_points": (c_int, [c_void_p, c_int32, c_void_p, c_int64, c_int, c_double, c_void_p, c_int]),
    "rt_face_visibility": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int, c_void_p]),
    "rt_visibility_rows": (c_int, [c_void_p, c_int32, c_void_p, c_int, c_void_p]),
    "rt_faces_visibility": (c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "rt_visibility_with_faces": (c_int, [c_void_p, c_int32, c_void_p, c_int, c_void_p, c_int32, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_int, c_void_p, c_int]),
    "rt_debug_visibility_stats": (c_int, [c_void_p]),
    "rt_debug_set_occlusion_mode": (c_int, [c_void_p, c_int]),
    "rt_debug_set_visibility_tiling": (c_int, [c_void_p, c_int32, c_int32]),
    "rt_scene_set_timing": (c_int, [c_void_p, c_int]),
    "rt_scene_wait_for_stream": (c_int, [c_void_p, c_void_p]),
    "rt_scene_set_deferred_sync": (c_int, [c_void_p, c_int]),
    "rt_scene_synchronize": (c_int, [c_void_p]),
    "rt_debug_set_stored_chain_limit": (c_int, [c_void_p, c_int64]),
    "rt_scene_set_reference_solver": (c_int, [c_void_p, c_int]),
    "rt_debug_set_reflect_mode": (c_int, [c_void_p, c_int]),
    "rt_debug_set_wall_hints": (c_int, [c_void_p, c_int]),
    "rt_debug_set_path_queue_capacity": (c_int, [c_void_p, c_int64]),
    "rt_visibility_matrix": (c_int, [c_void_p, c_int, c_void_p, c_int]),
    "rt_sharp_edges": (c_int, [c_void_p, c_double, c_void_p, c_void_p, c_int64, c_void_p]),
    "rt_solve_reflections": (c_int, [c_int, c_void_p, c_void_p, c_int64, c_void_p, c_int32, c_void_p, c_void_p]),
    "rt_solve_reflections_diffraction": (c_int, [c_int, c_void_p, c_void_p, c_int64, c_void_p, c_int32, c_void_p,
                                                 c_void_p, c_void_p]),
    "rt_solve_like_reference": (c_int, [c_int, c_void_p, c_void_p, c_int64, c_void_p, c_int32, c_void_p, c_void_p, c_void_p,
                                        c_void_p]),
    "rt_enumerate": (c_void_p, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int, c_int]),
    "rt_candidates_destroy": (None, [c_void_p]),
    "rt_candidates_group_count": (c_int32, [c_void_p]),
    "rt_candidates_counts": (c_int, [c_void_p, c_void_p]),
    "rt_candidates_group": (c_int, [c_void_p, c_int32, c_void_p, c_void_p]),
    "rt_solve_paths": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_int64, c_void_p, c_void_p, c_int]),
    "rt_solve_last_timing": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "rt_solve_last_rejudge": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "rt_fields_last_timing": (c_int, [c_void_p, c_void_p]),
    "rt_visibility_last_timing": (c_int, [c_void_p, c_void_p]),
    "rt_describe_paths": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    "rt_receiver_record_bytes": (c_int32, [c_int32]),
    "rt_receiver_records": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rt_format_array_rows": (c_int64, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_int64]),
    "rt_format_paths_json": (c_int64, [c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32,
                                       c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64]),
    "rt_channel_stats": (c_int, [c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_int32, c_double, c_void_p, c_int]),
    "rt_radiomap": (c_int, [c_int, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_double, c_void_p,
                            c_int32, c_void_p, c_void_p]),
    "rt_fields": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                          c_void_p, c_int64, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                          c_void_p, c_void_p, c_int]),
}


def exported_symbols():
    return sorted(_SIGNATURES)


def load():
    """Loads the shared library (no GPU call is made)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RtError(f"{LIB_PATH} is missing: build it with `python -m raytracing_amd.build_ext` "
                          "(there is no CPU implementation to fall back to)")
        # PyTorch ships its own copy of the HIP runtime.  A process that loads this library first (the system runtime)
        # and initialises torch.cuda afterwards holds two runtimes, and the second one finds no GPU; with torch
        # imported first the loader gives this library torch's copy and both work.  So: torch first, when there is one.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RtError(f"{what}: {load().rt_last_error().decode()}")


def require_gpu():
    lib = load()
    if lib.rt_device_count() < 1:
        raise RtError("no HIP device is visible: the path-solve kernels need a GPU (there is no CPU implementation)")
    return lib


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return a.ctypes.data if a is not None else None


def _device_bytes_ptr(t, shape, device):
    """Address of a device-resident byte array of the given shape (a contiguous uint8 or bool torch tensor on GPU
    ``device``)."""
    if not getattr(t, "is_cuda", False) or t.device.index != device:
        raise ValueError(f"the array must be a tensor on GPU {device}")
    if tuple(t.shape) != tuple(shape) or t.element_size() != 1 or not t.is_contiguous():
        raise ValueError(f"the array must be a contiguous tensor of bytes of shape {tuple(shape)}")
    return t.data_ptr()


def _device_matrix_ptr(t, n, device):
    """Address of a device-resident N x N byte matrix."""
    return _device_bytes_ptr(t, (n, n), device)


def default_device():
    return int(os.environ.get("LOCAL_RANK", "0")) % max(1, load().rt_device_count())


class Candidates:
    """Flat candidate lists of one emitter (``rt_enumerate``)."""

    def __init__(self, scene, vis, ev, edge_pairs, edges, max_order, with_diffraction=True):
        lib = require_gpu()
        self.scene = scene
        self.max_order = int(max_order)
        n = scene.n
        ev8 = np.ascontiguousarray(ev, dtype=np.uint8)
        if hasattr(vis, "data_ptr"):
            # a matrix that was painted into device memory (Scene.visibility_with_faces(matrix_out=...)) is read there,
            # on the library's own stream: after whatever the caller's stream still has queued for it
            vis_ptr, on_device = _device_matrix_ptr(vis, n, scene.device), 1
            scene.wait_for_torch_stream()
        else:
            vis = np.asarray(vis)
            # (a bool matrix is passed as it is: of lln's 58 MB a copy takes longer than the enumeration)
            vis8 = vis.view(np.uint8) if vis.dtype == np.bool_ and vis.flags.c_contiguous else np.ascontiguousarray(vis, dtype=np.uint8)
            if vis8.shape != (n, n):
                raise ValueError("visibility matrix does not match the scene")
            vis_ptr, on_device = _ptr(vis8), 0
        if ev8.shape != (n,):
            raise ValueError("emitter visibility does not match the scene")
        self.edge_pairs = np.ascontiguousarray(edge_pairs, dtype=np.int32).reshape(-1, 2)
        self.edges = _f64(edges).reshape(-1, 2, 3)
        self.handle = lib.rt_enumerate(scene.handle, vis_ptr, _ptr(ev8), _ptr(self.edge_pairs), _ptr(self.edges),
                                       len(self.edge_pairs), self.max_order, 1 if with_diffraction else 0, on_device)
        if not self.handle:
            raise RtError(f"rt_enumerate: {lib.rt_last_error().decode()}")
        self.n_groups = lib.rt_candidates_group_count(self.handle)
        self.counts = np.zeros(self.n_groups, dtype=np.int64)
        if self.n_groups:
            _check(lib.rt_candidates_counts(self.handle, _ptr(self.counts)), "rt_candidates_counts")
        self._groups = {}

    def group_kind(self, g):
        """(number of reflections, ends with a diffraction) of a group."""
        if g < self.max_order:
            return g + 1, False
        return g - self.max_order, True

    def group(self, g):
        """(chains int32 [count, n_refl], edge_ids int32 [count]) of a group, fetched once."""
        if g not in self._groups:
            n_refl, _ = self.group_kind(g)
            count = int(self.counts[g])
            chains = np.zeros((count, n_refl), dtype=np.int32)
            edge_ids = np.zeros(count, dtype=np.int32)
            _check(load().rt_candidates_group(self.handle, g, _ptr(chains) if chains.size else None, _ptr(edge_ids)),
                   "rt_candidates_group")
            self._groups[g] = (chains, edge_ids)
        return self._groups[g]

    def describe(self, group, cand):
        """Per-path kind / chain / edge arrays (``rt_describe_paths``) of (group, candidate) pairs."""
        group, cand = encode_group_cand(group, cand)
        n = len(group)
        kind = np.zeros(n, dtype=np.int32)
        chain = np.zeros((n, RT_MAX_ORDER), dtype=np.int32)
        edge_polys = np.zeros((n, 2), dtype=np.int32)
        edge_points = np.zeros((n, 6), dtype=np.float64)
        _check(load().rt_describe_paths(self.handle, _ptr(group), _ptr(cand), n, _ptr(kind), _ptr(chain),
                                        _ptr(edge_polys), _ptr(edge_points), 0), "rt_describe_paths")
        return kind, chain, edge_polys, edge_points

    def describe_device(self, group_ptr, cand_ptr, n, kind_ptr, chain_ptr, edge_polys_ptr, edge_points_ptr):
        """Device-pointer form of ``describe``: (group, candidate) pairs as ``rt_solve_paths`` left them in this GPU's
        memory; kind i32[n], chain i32[n, RT_MAX_ORDER], edge_polys i32[n, 2], edge_points f64[n, 6] written there."""
        _check(load().rt_describe_paths(self.handle, group_ptr, cand_ptr, int(n), kind_ptr, chain_ptr, edge_polys_ptr,
                                        edge_points_ptr, 1), "rt_describe_paths")

    @property
    def total(self):
        return int(self.counts.sum())

    def close(self):
        if getattr(self, "handle", None):
            load().rt_candidates_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Scene:
    """Device-resident polygon list (``rt_scene_create``)."""

    def __init__(self, arrays, device=None):
        lib = require_gpu()
        self.device = default_device() if device is None else int(device)
        self.arrays = arrays
        self.n = arrays.n
        self.handle = lib.rt_scene_create(self.device, arrays.n, _ptr(arrays.vert_off), _ptr(arrays.verts),
                                          _ptr(arrays.planes), _ptr(arrays.material), _ptr(arrays.building))
        if not self.handle:
            raise RtError(f"rt_scene_create: {lib.rt_last_error().decode()}")

    def close(self):
        if getattr(self, "handle", None):
            load().rt_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- obstruction ------------------------------------------------------------------------------
    def obstructed(self, segments, mask=None):
        segs = _f64(segments).reshape(-1, 6)
        out = np.zeros(len(segs), dtype=np.uint8)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        _check(load().rt_obstruct_segments(self.handle, _ptr(segs), len(segs), _ptr(m), _ptr(out), 0),
               "rt_obstruct_segments")
        return out.astype(bool)

    def obstructed_device(self, segs_ptr, n_seg, mask_ptr, out_ptr):
        """Device-pointer form of ``obstructed``: segs f64[n_seg, 6], mask u8[N] or None, out u8[n_seg] in this GPU's
        memory; returns once the flags are written."""
        _check(load().rt_obstruct_segments(self.handle, segs_ptr, int(n_seg), mask_ptr, out_ptr, 1), "rt_obstruct_segments")

    def obstructed_noalloc(self, segs_ptr, n_seg, mask_ptr, out_ptr, stream=None):
        """``rt_obstruct_segments_noalloc``: the same flags without any allocation, queued on ``stream`` (a raw
        hipStream_t, None = the default stream); returns without waiting."""
        _check(load().rt_obstruct_segments_noalloc(self.handle, segs_ptr, int(n_seg), mask_ptr, out_ptr, stream),
               "rt_obstruct_segments_noalloc")

    def segment_hits(self, segments):
        segs = _f64(segments).reshape(-1, 6)
        out = np.zeros((len(segs), self.n), dtype=np.uint8)
        for i in range(len(segs)):
            _check(load().rt_segment_hits(self.handle, _ptr(segs[i]), _ptr(out[i])), "rt_segment_hits")
        return out.astype(bool)

    def contains_points(self, polygon_index, points, check_in_plane=False, plane_tol=1e-8):
        pts = _f64(points).reshape(-1, 3)
        out = np.zeros(len(pts), dtype=np.uint8)
        _check(load().rt_contains_points(self.handle, int(polygon_index), _ptr(pts), len(pts), int(bool(check_in_plane)),
                                         float(plane_tol), _ptr(out), 0), "rt_contains_points")
        return out.astype(bool)

    # ---- visibility / sharp edges ----------------------------------------------------------------------
    def face_visibility(self, face_points, face_parametric, self_index=-1, strict=False):
        pts = _f64(face_points).reshape(-1, 3)
        normal = _f64(np.asarray(face_parametric)[:3])
        out = np.zeros(self.n, dtype=np.uint8)
        _check(load().rt_face_visibility(self.handle, _ptr(pts), len(pts), _ptr(normal), int(self_index),
                                         int(bool(strict)), _ptr(out)), "rt_face_visibility")
        return out.astype(bool)

    def set_occlusion_mode(self, mode):
        """0: obstruction kernel chosen by scene size (default); 1: every-polygon walk; 2: grid walk."""
        _check(load().rt_debug_set_occlusion_mode(self.handle, int(mode)), "rt_debug_set_occlusion_mode")

    def set_reference_solver(self, on):
        """Reference-compatible mode of the solve: every found path is re-judged with a restatement of SciPy's hybr on
        the reference's own systems and dropped when the reference's solver would not have reported it."""
        _check(load().rt_scene_set_reference_solver(self.handle, int(bool(on))), "rt_scene_set_reference_solver")

    def set_visibility_tiling(self, min_candidates, per_tile):
        """Faces with at least ``min_candidates`` candidates are painted in tiles (rectangles of their view) of about ``per_tile``
        candidates (``per_tile`` 0: every face whole)."""
        _check(load().rt_debug_set_visibility_tiling(self.handle, int(min_candidates), int(per_tile)),
               "rt_debug_set_visibility_tiling")

    def set_reflect_mode(self, mode):
        """0: reflection kernel chosen by batch size (default); 1: receivers one by one; 2: receivers in spatial tiles."""
        _check(load().rt_debug_set_reflect_mode(self.handle, int(mode)), "rt_debug_set_reflect_mode")

    def set_wall_hints(self, on):
        """Wall hints around the receivers for the diffraction kernel of a large batch: on (default) or off."""
        _check(load().rt_debug_set_wall_hints(self.handle, int(bool(on))), "rt_debug_set_wall_hints")

    def wait_for_stream(self, stream_handle):
        """Orders the scene's stream after what is queued on the caller's stream (a hipStream_t as an integer, e.g.
        ``torch.cuda.current_stream().cuda_stream``; 0 = the null stream), on the device: the host does not wait."""
        _check(load().rt_scene_wait_for_stream(self.handle, c_void_p(int(stream_handle))), "rt_scene_wait_for_stream")

    def wait_for_torch_stream(self):
        """``wait_for_stream`` on PyTorch's current stream of this scene's GPU: what a caller has queued there for a
        tensor it now hands to the library (which works on a stream of its own) is finished before the library reads or
        writes the tensor."""
        import torch

        self.wait_for_stream(torch.cuda.current_stream(torch.device("cuda", self.device)).cuda_stream)

    def set_deferred_sync(self, on):
        """While on, the device-pointer forms of the path description, the field step and the receiver records only
        queue their work on the scene's stream; ``synchronize`` waits for it."""
        _check(load().rt_scene_set_deferred_sync(self.handle, 1 if on else 0), "rt_scene_set_deferred_sync")

    def synchronize(self):
        """Waits for everything queued on the scene's stream (and reports a launch that failed)."""
        _check(load().rt_scene_synchronize(self.handle), "rt_scene_synchronize")

    def set_stored_chain_limit(self, n_bytes):
        """Largest deepest reflection level (bytes of its chains) that ``enumerate`` stores; above it the level's
        chains are formed in the kernels from the level before (same candidates and indices)."""
        _check(load().rt_debug_set_stored_chain_limit(self.handle, int(n_bytes)), "rt_debug_set_stored_chain_limit")

    def receiver_records_device(self, n_rx, n_freq, power_db_ptr, n_paths, path_rx_ptr, path_kind_ptr, path_status_ptr,
                                out_ptr):
        """One fixed-size record per receiver from the arrays ``fields_device`` left on this GPU (rt_receiver_records)."""
        _check(load().rt_receiver_records(self.handle, int(n_rx), int(n_freq), power_db_ptr, int(n_paths), path_rx_ptr,
                                          path_kind_ptr, path_status_ptr, out_ptr), "rt_receiver_records")

    def set_timing(self, on):
        """Per-kernel-family timing of the solves on (default) or off; ``last_timing`` reports zeros while off."""
        _check(load().rt_scene_set_timing(self.handle, int(bool(on))), "rt_scene_set_timing")

    def set_path_queue_capacity(self, n_paths):
        """First size of the solve's path queue (0 = automatic); too small a queue is grown and the chunk redone."""
        _check(load().rt_debug_set_path_queue_capacity(self.handle, int(n_paths)), "rt_debug_set_path_queue_capacity")

    def faces_visibility(self, faces, strict=False):
        """Visibility vectors of several faces in one launch; ``faces`` = list of (points (V,3), parametric (4,),
        self_index)."""
        off = np.zeros(len(faces) + 1, dtype=np.int32)
        np.cumsum([len(f[0]) for f in faces], out=off[1:])
        verts = _f64(np.concatenate([np.asarray(f[0], dtype=float).reshape(-1, 3) for f in faces]))
        normals = _f64(np.array([np.asarray(f[1], dtype=float)[:3] for f in faces]))
        selfs = np.array([int(f[2]) for f in faces], dtype=np.int32)
        out = np.zeros((len(faces), self.n), dtype=np.uint8)
        _check(load().rt_faces_visibility(self.handle, len(faces), _ptr(off), _ptr(verts), _ptr(normals), _ptr(selfs),
                                          int(bool(strict)), _ptr(out)), "rt_faces_visibility")
        return out.astype(bool)

    def visibility_with_faces(self, faces, row_ids=None, strict=False, faces_strict=False, matrix_out=None, rows_out=None,
                              faces_out=None):
        """The visibility matrix (``row_ids`` None) or the listed unsymmetrised rows, and the visibility vectors of
        ``faces`` (as ``faces_visibility`` takes them), painted side by side: returns (matrix or rows, face vectors).
        ``matrix_out`` (whole matrix) / ``rows_out`` (listed rows) / ``faces_out``: byte tensors on this scene's GPU,
        [N, N] / [len(row_ids), N] / [len(faces), N], that receive the result instead of a host array; each is
        returned in its result's place (``enumerate`` takes such a matrix as it is)."""
        off = np.zeros(len(faces) + 1, dtype=np.int32)
        np.cumsum([len(f[0]) for f in faces], out=off[1:])
        verts = _f64(np.concatenate([np.asarray(f[0], dtype=float).reshape(-1, 3) for f in faces])) if faces else np.zeros((0, 3))
        normals = _f64(np.array([np.asarray(f[1], dtype=float)[:3] for f in faces])) if faces else np.zeros((0, 3))
        selfs = np.array([int(f[2]) for f in faces], dtype=np.int32)
        # (the library copies whole buffers out: nothing of them has to be cleared first)
        on_device = 0
        if faces_out is not None:
            f_ptr, on_device = _device_bytes_ptr(faces_out, (len(faces), self.n), self.device), 2
        else:
            f_out = np.empty((len(faces), self.n), dtype=np.uint8)
            f_ptr = _ptr(f_out)
        if row_ids is None:
            if rows_out is not None:
                raise ValueError("rows_out goes with row_ids; the whole matrix goes into matrix_out")
            rows, n_rows, r_ptr = None, 0, None
            if matrix_out is not None:
                out_ptr, on_device = _device_matrix_ptr(matrix_out, self.n, self.device), on_device | 1
            else:
                r_out = np.empty((self.n, self.n), dtype=np.uint8)
                out_ptr = _ptr(r_out)
        else:
            if matrix_out is not None:
                raise ValueError("matrix_out goes with the whole matrix, not with row_ids")
            rows = np.ascontiguousarray(row_ids, dtype=np.int32)
            n_rows, r_ptr = len(rows), _ptr(rows)
            if rows_out is not None:
                out_ptr, on_device = _device_bytes_ptr(rows_out, (len(rows), self.n), self.device), on_device | 1
            else:
                r_out = np.empty((len(rows), self.n), dtype=np.uint8)
                out_ptr = _ptr(r_out)
        if on_device:
            self.wait_for_torch_stream()  # the caller's tensors: after what its stream still has queued for them
        _check(load().rt_visibility_with_faces(self.handle, n_rows, r_ptr, int(bool(strict)), out_ptr, len(faces),
                                               _ptr(off), _ptr(verts), _ptr(normals), _ptr(selfs), int(bool(faces_strict)),
                                               f_ptr, on_device), "rt_visibility_with_faces")
        first = matrix_out if matrix_out is not None else (rows_out if rows_out is not None else r_out.view(bool))
        return first, (faces_out if faces_out is not None else f_out.view(bool))  # (0 and 1 only: viewed, not copied)

    def visibility_rows(self, row_ids, strict=False):
        """Rows of the visibility matrix before symmetrisation: bool [len(row_ids), N]."""
        rows = np.ascontiguousarray(row_ids, dtype=np.int32)
        out = np.zeros((len(rows), self.n), dtype=np.uint8)
        _check(load().rt_visibility_rows(self.handle, len(rows), _ptr(rows), int(bool(strict)), _ptr(out)),
               "rt_visibility_rows")
        return out.view(bool)

    def visibility_matrix(self, strict=False):
        out = np.zeros((self.n, self.n), dtype=np.uint8)
        _check(load().rt_visibility_matrix(self.handle, int(bool(strict)), _ptr(out), 0), "rt_visibility_matrix")
        return out.view(bool)

    def sharp_edges(self, min_angle=10.0):
        cap = 2 * len(self.arrays.verts) + 16
        pairs = np.zeros((cap, 2), dtype=np.int32)
        edges = np.zeros((cap, 2, 3), dtype=np.float64)
        count = np.zeros(1, dtype=np.int64)
        _check(load().rt_sharp_edges(self.handle, float(min_angle), _ptr(pairs), _ptr(edges), cap, _ptr(count)),
               "rt_sharp_edges")
        k = int(count[0])
        return pairs[:k].copy(), edges[:k].copy()

    # ---- solve ----------------------------------------------------------------------------------------
    def enumerate(self, vis, ev, edge_pairs, edges, max_order, with_diffraction=True):
        return Candidates(self, vis, ev, edge_pairs, edges, max_order, with_diffraction)

    def solve_paths(self, candidates, tx, rx, capacity=None):
        """Host-array form of ``rt_solve_paths``: returns dict(rx, group, cand, points, n_attempted)."""
        lib = load()
        tx = _f64(tx).reshape(3)
        rx = _f64(rx).reshape(-1, 3)
        if capacity is None:
            capacity = max(4096, 64 * len(rx))
        while True:
            o_rx = np.zeros(capacity, dtype=np.int32)
            o_group = np.zeros(capacity, dtype=np.int32)
            o_cand = np.zeros(capacity, dtype=np.int32)
            o_pts = np.zeros((capacity, RT_MAX_ORDER, 3), dtype=np.float64)
            found = np.zeros(1, dtype=np.int64)
            attempted = np.zeros(1, dtype=np.int64)
            _check(lib.rt_solve_paths(self.handle, candidates.handle, _ptr(tx), _ptr(rx), len(rx), _ptr(o_rx),
                                      _ptr(o_group), _ptr(o_cand), _ptr(o_pts), capacity, _ptr(found), _ptr(attempted),
                                      0), "rt_solve_paths")
            k = int(found[0])
            if k <= capacity:
                break
            capacity = k + k // 8 + 16
        g, c = decode_group_cand(o_group[:k], o_cand[:k])
        order = np.lexsort((c, g, o_rx[:k]))
        return dict(rx=o_rx[:k][order], group=g[order], cand=c[order], points=o_pts[:k][order],
                    n_attempted=int(attempted[0]))

    def solve_paths_device(self, candidates, tx, rx_ptr, n_rx, out_rx_ptr, out_group_ptr, out_cand_ptr, out_points_ptr,
                           capacity):
        """Device-pointer form of ``rt_solve_paths`` (receivers and outputs already in this GPU's memory).
        Returns (n_found, n_attempted)."""
        # a solve of a thousand receivers lasts 70 us: the two result words and the emitter's array are kept between
        # calls instead of being made anew (a few microseconds of NumPy per call)
        try:
            res, res_ptr, tx_buf, tx_ptr, fn = self._solve_call
        except AttributeError:
            res, tx_buf = np.zeros(2, dtype=np.int64), np.zeros(3, dtype=np.float64)
            res_ptr, tx_ptr, fn = res.ctypes.data, tx_buf.ctypes.data, load().rt_solve_paths
            self._solve_call = (res, res_ptr, tx_buf, tx_ptr, fn)
        tx_buf[:] = np.asarray(tx, dtype=np.float64).reshape(3)
        if fn(self.handle, candidates.handle, tx_ptr, rx_ptr, n_rx, out_rx_ptr, out_group_ptr, out_cand_ptr,
              out_points_ptr, capacity, res_ptr, res_ptr + 8, 1) != 0:
            _check(-1, "rt_solve_paths")
        return int(res[0]), int(res[1])

    def fields(self, tx, rx, path_rx, path_kind, path_chain, path_edge_polys, path_edge_points, path_points,
               frequencies, tx_gain=1.0, rx_gain=1.0, p_in=1.0, isotropic=True, alpha=10, aim_tx=None, aim_rx=None,
               path_origin=None, path_e_in=None, n_late_polygons=0):
        """Host-array form of ``rt_fields``; returns dict(field complex [P,F,3], power [P,F], angles [P,4], time [P],
        status [P], power_db [R,F]).  ``n_late_polygons``: the scene's last polygons that were added after the paths
        were found (tree crowns): every segment is attenuated by the ones it crosses."""
        tx = _f64(tx).reshape(3)
        rx = _f64(rx).reshape(-1, 3)
        p_rx = np.ascontiguousarray(path_rx, dtype=np.int32)
        n = len(p_rx)
        p_kind = np.ascontiguousarray(path_kind, dtype=np.int32).reshape(n)
        p_chain = np.ascontiguousarray(path_chain, dtype=np.int32).reshape(n, RT_MAX_ORDER)
        p_ep = np.ascontiguousarray(path_edge_polys, dtype=np.int32).reshape(n, 2)
        p_epts = _f64(path_edge_points).reshape(n, 6)
        p_pts = _f64(path_points).reshape(n, RT_MAX_ORDER * 3)
        freqs = _f64(np.atleast_1d(frequencies))
        nf = len(freqs)
        ap = AntennaParams(float(tx_gain), float(rx_gain), float(p_in), int(bool(isotropic)), int(alpha))
        field = np.zeros((n, nf, 6), dtype=np.float64)
        power = np.zeros((n, nf), dtype=np.float64)
        angles = np.zeros((n, 4), dtype=np.float64)
        time = np.zeros(n, dtype=np.float64)
        status = np.zeros(n, dtype=np.int32)
        power_db = np.zeros((len(rx), nf), dtype=np.float64)
        keep = []
        opts = None
        if any(a is not None for a in (aim_tx, aim_rx, path_origin, path_e_in)) or n_late_polygons:
            def dev(a, shape):
                if a is None:
                    return None
                a = _f64(a).reshape(shape)
                keep.append(a)
                return a.ctypes.data
            e_in = None
            if path_e_in is not None:
                e = np.asarray(path_e_in, dtype=np.complex128).reshape(n, 3)
                e_in = np.stack([e.real, e.imag], axis=-1).reshape(n, 6)
            opts = FieldOptions(dev(aim_tx, (len(rx), 3)), dev(aim_rx, (len(rx), 3)), dev(path_origin, (n, 3)),
                                dev(e_in, (n, 6)), int(n_late_polygons))
        _check(load().rt_fields(self.handle, _ptr(tx), _ptr(rx), len(rx), _ptr(p_rx), _ptr(p_kind), _ptr(p_chain),
                                _ptr(p_ep), _ptr(p_epts), _ptr(p_pts), n, _ptr(freqs), nf, ctypes.byref(ap),
                                ctypes.byref(opts) if opts is not None else None, _ptr(field), _ptr(power), _ptr(angles), _ptr(time), _ptr(status), _ptr(power_db), 0),
               "rt_fields")
        cfield = field[..., 0::2] + 1j * field[..., 1::2]
        return dict(field=cfield, power=power, angles=angles, time=time, status=status, power_db=power_db)

    def fields_device(self, tx, rx_ptr, n_rx, path_rx_ptr, kind_ptr, chain_ptr, edge_polys_ptr, edge_points_ptr,
                      points_ptr, n_paths, frequencies, field_ptr, power_ptr, angles_ptr, time_ptr, status_ptr,
                      power_db_ptr, tx_gain=1.0, rx_gain=1.0, p_in=1.0, isotropic=True, alpha=10, n_late_polygons=0):
        """Device-pointer form of ``rt_fields``: the per-path arrays as ``rt_solve_paths`` and ``describe_device`` left
        them in this GPU's memory, the outputs written there (field f64[P, F, 6], power f64[P, F], angles f64[P, 4],
        time f64[P], status i32[P], power_db f64[R, F]); the emitter and the frequencies are host values."""
        tx = _f64(tx).reshape(3)
        freqs = _f64(np.atleast_1d(frequencies))
        ap = AntennaParams(float(tx_gain), float(rx_gain), float(p_in), int(bool(isotropic)), int(alpha))
        opts = FieldOptions(None, None, None, None, int(n_late_polygons)) if n_late_polygons else None
        _check(load().rt_fields(self.handle, _ptr(tx), rx_ptr, int(n_rx), path_rx_ptr, kind_ptr, chain_ptr,
                                edge_polys_ptr, edge_points_ptr, points_ptr, int(n_paths), _ptr(freqs), len(freqs),
                                ctypes.byref(ap), ctypes.byref(opts) if opts is not None else None, field_ptr, power_ptr,
                                angles_ptr, time_ptr, status_ptr, power_db_ptr, 1), "rt_fields")

    def fields_timing(self):
        """Device milliseconds of the last ``rt_fields`` call: dict(kernel, call)."""
        ms = np.zeros(2, dtype=np.float64)
        _check(load().rt_fields_last_timing(self.handle, _ptr(ms)), "rt_fields_last_timing")
        return dict(kernel=float(ms[0]), call=float(ms[1]))

    def visibility_timing(self):
        """Device milliseconds of the last visibility call: dict(rows=dict(prepare, sort, paint, total), faces=...)."""
        ms = np.zeros(8, dtype=np.float64)
        _check(load().rt_visibility_last_timing(self.handle, _ptr(ms)), "rt_visibility_last_timing")
        names = ("prepare", "sort", "paint", "total")
        return dict(rows={k: float(v) for k, v in zip(names, ms[:4])}, faces={k: float(v) for k, v in zip(names, ms[4:])})

    def rejudge_timing(self):
        """The re-judging step of the last solve in reference-compatible mode: dict(ms, n_in, n_kept); zeros when off."""
        ms = np.zeros(1, dtype=np.float64)
        counts = np.zeros(2, dtype=np.int64)
        _check(load().rt_solve_last_rejudge(self.handle, _ptr(ms), counts.ctypes.data, counts.ctypes.data + 8),
               "rt_solve_last_rejudge")
        return dict(ms=float(ms[0]), n_in=int(counts[0]), n_kept=int(counts[1]))

    def last_timing(self):
        ms = np.zeros(4, dtype=np.float64)
        launches = np.zeros(4, dtype=np.int64)
        segs = np.zeros(1, dtype=np.int64)
        _check(load().rt_solve_last_timing(self.handle, _ptr(ms), _ptr(launches), _ptr(segs)), "rt_solve_last_timing")
        return dict(ms=dict(reflect=ms[0], diffract=ms[1], obstruct=ms[2], finalize=ms[3]),
                    launches=dict(reflect=int(launches[0]), diffract=int(launches[1]), obstruct=int(launches[2]),
                                  finalize=int(launches[3])), segments=int(segs[0]))


# ---- the two words that name a path's candidate -----------------------------------------------------------
# (rt_common.h: a candidate index of 2^31 or more keeps its low 31 bits in the candidate word and puts the rest into
# bits 8 and up of the group word; below 2^31 the words are the group and the index as they are)
RT_GROUP_BITS = 8


def decode_group_cand(group_word, cand_word):
    """(group as int32, candidate index as int32 -- int64 when some index needs it --) of the library's two words."""
    group_word = np.asarray(group_word, dtype=np.int32)
    cand_word = np.asarray(cand_word, dtype=np.int32)
    wide = group_word >= (1 << RT_GROUP_BITS)
    if not wide.any():
        return group_word, cand_word
    group = np.where(group_word < 0, group_word, group_word & ((1 << RT_GROUP_BITS) - 1)).astype(np.int32)
    high = np.where(group_word < 0, 0, group_word.astype(np.int64) >> RT_GROUP_BITS)
    return group, (high << 31) | cand_word.astype(np.int64)


def encode_group_cand(group, cand):
    """The library's two int32 words of (group, candidate index) pairs."""
    group = np.asarray(group)
    cand = np.asarray(cand).astype(np.int64)
    high = np.where(group < 0, 0, cand >> 31)
    group_word = np.where(group < 0, group, group.astype(np.int64) | (high << RT_GROUP_BITS))
    cand_word = np.where(group < 0, cand, cand & 0x7FFFFFFF)
    return np.ascontiguousarray(group_word, dtype=np.int32), np.ascontiguousarray(cand_word, dtype=np.int32)


# ---- entry points that need no scene ---------------------------------------------------------------------

def polygon_contains_points(polygon, points, check_in_plane=False, plane_tol=1e-8):
    from .geometry import scene_arrays_of

    scene = scene_arrays_of([polygon]).device_scene()
    return scene.contains_points(0, points, check_in_plane, plane_tol)


def solve_reflections(origin, destinations, planes, device=None):
    lib = require_gpu()
    planes = _f64(planes).reshape(-1, 4)
    dests = _f64(destinations).reshape(-1, 3)
    origin = _f64(origin).reshape(3)
    k = len(planes)
    out = np.zeros((len(dests), k, 3), dtype=np.float64)
    ok = np.zeros(len(dests), dtype=np.uint8)
    _check(lib.rt_solve_reflections(default_device() if device is None else device, _ptr(origin), _ptr(dests),
                                    len(dests), _ptr(planes), k, _ptr(out), _ptr(ok)), "rt_solve_reflections")
    return out, ok.astype(bool)


def solve_reflections_and_diffraction(origin, destinations, planes, edge, device=None):
    lib = require_gpu()
    planes = _f64(planes).reshape(-1, 4)
    dests = _f64(destinations).reshape(-1, 3)
    origin = _f64(origin).reshape(3)
    edge = _f64(edge).reshape(2, 3)
    k = len(planes)
    out = np.zeros((len(dests), k + 1, 3), dtype=np.float64)
    ok = np.zeros(len(dests), dtype=np.uint8)
    _check(lib.rt_solve_reflections_diffraction(default_device() if device is None else device, _ptr(origin),
                                                _ptr(dests), len(dests), _ptr(planes) if k else None, k, _ptr(edge),
                                                _ptr(out), _ptr(ok)), "rt_solve_reflections_diffraction")
    return out, ok.astype(bool)


def solve_like_reference(origin, destinations, planes, edge=None, device=None):
    """The reference's own solver call (SciPy hybr, restated on the GPU) on the system of ``planes`` (k, 4) and ``edge``
    (2, 3) or None for every destination: returns (x [n, k (+1), 3], success bool [n], nfev int32 [n])."""
    lib = require_gpu()
    planes = _f64(planes).reshape(-1, 4)
    dests = _f64(destinations).reshape(-1, 3)
    origin = _f64(origin).reshape(3)
    e = None if edge is None else _f64(edge).reshape(2, 3)
    k = len(planes)
    out = np.zeros((len(dests), k + (0 if e is None else 1), 3), dtype=np.float64)
    ok = np.zeros(len(dests), dtype=np.uint8)
    nfev = np.zeros(len(dests), dtype=np.int32)
    _check(lib.rt_solve_like_reference(default_device() if device is None else device, _ptr(origin), _ptr(dests), len(dests),
                                       _ptr(planes) if k else None, k, _ptr(e), _ptr(out), _ptr(ok), _ptr(nfev)),
           "rt_solve_like_reference")
    return out, ok.astype(bool), nfev


def format_array_rows(values):
    """Strings of the rows of a 2-D real or complex array as the reference's CSV holds small arrays
    (``numpy.array2string(row, separator=",")`` without spaces); None for a row with a non-finite value."""
    a = np.asarray(values)
    if a.ndim != 2 or not 1 <= a.shape[1] <= 16:
        raise ValueError("format_array_rows: a 2-D array with 1 to 16 columns")
    is_complex = a.dtype.kind == "c"
    a = np.ascontiguousarray(a, dtype=np.complex128 if is_complex else np.float64)
    if len(a) == 0:
        return []
    capacity = 64 * a.shape[0] * a.shape[1] * (2 if is_complex else 1) + 8 * a.shape[0]
    buf = ctypes.create_string_buffer(capacity)
    n = load().rt_format_array_rows(a.ctypes.data, a.shape[0], a.shape[1], int(is_complex), buf, capacity)
    if n < 0:
        raise RtError("rt_format_array_rows failed")
    rows = buf.raw[:n].decode("ascii").split("\n")[:-1]
    return [r if r else None for r in rows]


def format_paths_json(keys, path_rx, path_order, n_points, points, n_chain, chain, edge_key=None, edge_points=None):
    """The characters ``json.dumps`` gives for a container of solved paths ``{receiver: {order: [item, ...]}}`` (see
    rt_format_paths_json): paths grouped by receiver in the order of ``keys`` and by order within a receiver;
    ``points`` [P, m, 3] with the first ``n_points[i]`` used, ``chain`` [P, k] with the first ``n_chain[i]`` used;
    ``edge_key`` [P, 2] and ``edge_points`` [P, 2, 3] for diffraction items."""
    keys = np.ascontiguousarray(keys, dtype=np.int64)
    path_rx = np.ascontiguousarray(path_rx, dtype=np.int64)
    n = len(path_rx)
    path_order = np.ascontiguousarray(path_order, dtype=np.int32)
    n_points = np.ascontiguousarray(n_points, dtype=np.int32)
    n_chain = np.ascontiguousarray(n_chain, dtype=np.int32)
    points = np.ascontiguousarray(points, dtype=np.float64).reshape(n, -1) if n else np.zeros((0, 3))
    chain = np.ascontiguousarray(chain, dtype=np.int32).reshape(n, -1) if n else np.zeros((0, 1), dtype=np.int32)
    if chain.shape[1] == 0:
        chain = np.zeros((n, 1), dtype=np.int32)
    ek = ep = None
    if edge_key is not None:
        ek = np.ascontiguousarray(edge_key, dtype=np.int32).reshape(n, 2)
        ep = np.ascontiguousarray(edge_points, dtype=np.float64).reshape(n, 6)
    capacity = 64 + 16 * len(keys) + int(n_points.sum()) * 84 + int(n_chain.sum()) * 13 + n * (140 if ek is not None else 32)
    buf = np.empty(capacity, dtype=np.uint8)
    got = load().rt_format_paths_json(n, path_rx.ctypes.data, path_order.ctypes.data, n_points.ctypes.data, points.ctypes.data,
                                      points.shape[1], n_chain.ctypes.data, chain.ctypes.data, chain.shape[1],
                                      ek.ctypes.data if ek is not None else None, ep.ctypes.data if ep is not None else None,
                                      len(keys), keys.ctypes.data, buf.ctypes.data, capacity)
    if got < 0:
        raise RtError(f"rt_format_paths_json failed ({got})")
    return buf[:got].tobytes().decode("ascii")


def channel_stats(group_off, power, time, el_deg, az_deg, xyz, is_los, n_strongest=1, power_percent=0.99, device=None):
    """Per-receiver channel statistics of a result table whose rows are grouped by receiver (see rt_channel_stats):
    returns an (n_groups, 8) array [delay, rms_az, rms_el, percent, has_los, win_az, win_el, n_paths]."""
    lib = require_gpu()
    off = np.ascontiguousarray(group_off, dtype=np.int64)
    n_groups = len(off) - 1
    n = int(off[-1]) if n_groups >= 0 else 0
    arrays = [_f64(power).reshape(-1), _f64(time).reshape(-1), _f64(el_deg).reshape(-1), _f64(az_deg).reshape(-1),
              _f64(xyz).reshape(-1, 3)]
    los = np.ascontiguousarray(is_los, dtype=np.uint8).reshape(-1)
    if any(len(a) != n for a in arrays) or len(los) != n:
        raise ValueError("channel_stats: the per-path arrays must have group_off[-1] entries")
    out = np.zeros((max(n_groups, 0), 8), dtype=np.float64)
    if n_groups > 0:
        ptrs = [_ptr(a) if n else None for a in arrays]
        _check(lib.rt_channel_stats(default_device() if device is None else device, n_groups, _ptr(off), *ptrs,
                                    _ptr(los) if n else None, int(n_strongest), float(power_percent), _ptr(out), 0),
               "rt_channel_stats")
    return out


def radiomap(values, triangles, transform, query, fill_value, ring=None, interpolate=True, device=None):
    """Linear interpolation of per-receiver values over a triangulation at the query points, and (with a ring) which
    query points lie strictly inside it (see rt_radiomap): returns (values or None, inside or None)."""
    lib = require_gpu()
    query = _f64(query).reshape(-1, 2)
    q = len(query)
    out = inside = None
    vals = tris = tr = None
    if interpolate:
        vals = _f64(values).reshape(-1)
        tris = np.ascontiguousarray(triangles, dtype=np.int32).reshape(-1, 3)
        tr = _f64(transform).reshape(-1, 6)
        if len(tr) != len(tris):
            raise ValueError("radiomap: one transform per triangle")
        out = np.full(q, float(fill_value), dtype=np.float64)
    if ring is not None:
        ring = _f64(ring).reshape(-1, 2)
        inside = np.zeros(q, dtype=np.uint8)
    if q:
        _check(lib.rt_radiomap(default_device() if device is None else device, len(vals) if interpolate else 0,
                               _ptr(vals), len(tris) if interpolate else 0, _ptr(tris), _ptr(tr), q, _ptr(query),
                               float(fill_value), _ptr(out), len(ring) if ring is not None else 0, _ptr(ring),
                               _ptr(inside)), "rt_radiomap")
    return out, (inside.astype(bool) if inside is not None else None)
