"""Receiver-parallel driver with the reference's entry point (``electromagnetism_fun/multithread_solve.py:34-112``).

The reference maps one receiver per worker process with ``pathos`` and pickles the whole place to each of them.  Here
the parallel axis is the same — receivers — but the workers are the GPUs of one node: one process per GPU under
``torch.distributed`` (RCCL), each holding the full scene and solving a contiguous block of receivers; the per-path
records are combined with all-gathers (counts first, then padded payloads) and rank 0 writes the reference's two
result files.  Without an initialised process group it runs on one GPU.
"""
import os
import time

import numpy as np

from . import electromagnetism_utils
from .electromagnetism import my_field_computation
from .ray_tracing import RayTracingProblem


def shard_bounds(n_items, world_size, rank):
    """Contiguous block [begin, end) of rank ``rank`` when ``n_items`` are split as evenly as possible."""
    base, extra = divmod(int(n_items), int(world_size))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def _dist():
    try:
        import torch.distributed as dist
    except ImportError:
        return None
    return dist if dist.is_available() and dist.is_initialized() else None


def _device_for_collectives(dist):
    import torch

    if dist.get_backend() == "nccl":
        return torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    return torch.device("cpu")


def all_gather_rows(local, dist=None):
    """All-gather of 2-D float64 arrays with different row counts: counts first, then one padded all-gather.
    Returns the rows of all ranks, rank after rank."""
    import torch

    dist = dist or _dist()
    local = np.ascontiguousarray(local, dtype=np.float64)
    assert local.ndim == 2
    if dist is None:
        return local
    dev = _device_for_collectives(dist)
    world = dist.get_world_size()
    count = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count)
    counts = [int(c.item()) for c in counts]
    width = local.shape[1]
    pad = max(max(counts), 1)
    buf = torch.zeros((pad, width), dtype=torch.float64, device=dev)
    if local.shape[0]:
        buf[:local.shape[0]] = torch.from_numpy(local).to(dev)
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)
    return np.concatenate([b[:c].cpu().numpy() for b, c in zip(bufs, counts)], axis=0)


def _encode_paths(problem, rx_ids):
    """One row per path of the listed receivers: [rx, kind (0 LOS, 1 R, 2 D), order, n_points, points (6*3), chain (4),
    edge key (2), edge (6)] — fixed width so a single collective moves them."""
    rows = []
    for r in rx_ids:
        for order, lst in problem.reflections[r].items():
            for lines, chain in lst:
                row = np.full(4 + 18 + 4 + 2 + 6, np.nan)
                row[:4] = [r, 1, order, len(lines)]
                row[4:4 + lines.size] = lines.reshape(-1)
                row[22:22 + len(chain)] = chain
                rows.append(row)
        for order, lst in problem.diffractions[r].items():
            for lines, chain, key, edge in lst:
                row = np.full(4 + 18 + 4 + 2 + 6, np.nan)
                row[:4] = [r, 2, order, len(lines)]
                row[4:4 + lines.size] = lines.reshape(-1)
                row[22:22 + len(chain)] = chain
                row[26:28] = key
                row[28:34] = np.asarray(edge).reshape(-1)
                rows.append(row)
    return np.array(rows, dtype=np.float64).reshape(-1, 34)


def _decode_paths(problem, rows):
    for row in rows:
        r, kind, order, n_pts = (int(v) for v in row[:4])
        lines = row[4:4 + 3 * n_pts].reshape(n_pts, 3).copy()
        n_chain = order if kind == 1 else order - 1
        chain = [int(v) for v in row[22:22 + n_chain]]
        if kind == 1:
            problem.reflections[r][order].append((lines, chain))
        else:
            problem.diffractions[r][order].append((lines, chain, (int(row[26]), int(row[27])), row[28:34].reshape(2, 3).copy()))


def gather_visibility_rows(rows_of, n, dist=None):
    """Visibility matrix of an n-polygon scene from rows computed rank by rank: rank r computes rows r, r + W, r + 2W ...
    (``rows_of(ids)`` -> bool [len(ids), n], unsymmetrised; interleaved because neighbouring rows cost alike), one
    all-gather of the byte rows, then matrix = rows | rows^T (geometry.py:1036-1048)."""
    import torch

    dist = dist or _dist()
    world = dist.get_world_size() if dist else 1
    rank = dist.get_rank() if dist else 0
    mine = np.arange(rank, n, world, dtype=np.int32)
    local = np.ascontiguousarray(rows_of(mine), dtype=np.uint8).reshape(len(mine), n)
    matrix = np.zeros((n, n), dtype=bool)
    if dist is None:
        matrix[mine] = local.astype(bool)
    else:
        dev = _device_for_collectives(dist)
        pad = (n + world - 1) // world
        buf = torch.zeros((pad, n), dtype=torch.uint8, device=dev)
        buf[:len(mine)] = torch.from_numpy(local).to(dev)
        bufs = [torch.zeros_like(buf) for _ in range(world)]
        dist.all_gather(bufs, buf)
        for r, b in enumerate(bufs):
            ids = np.arange(r, n, world)
            matrix[ids] = b[:len(ids)].cpu().numpy().astype(bool)
    return matrix | matrix.T


def sharded_visibility_matrix(place, strict=False, emitter=None, n_screens=6):
    """``place.get_visibility_matrix()`` with the rows dealt to the ranks of the process group; the result is cached on
    the place like the reference's (geometry.py:1467-1483), on every rank.  With ``emitter`` every rank also paints the
    emitter's screens (ray_tracing.py:210-228) beside its rows and leaves them on the place for the problem's
    ``precompute``."""
    from . import geometry as geom

    if place.visibility_matrix is None:
        polygons = place.get_polygons_list()
        scene = geom.scene_arrays_of(polygons).device_scene()
        if emitter is None:
            rows_of = lambda ids: scene.visibility_rows(ids, strict)
        else:
            emitter = np.asarray(emitter, dtype=float)
            cube = geom.Cube.by_point_and_side_length(emitter, 2 * 0.1)
            screens = [(cube.polygons[i].points, cube.polygons[i].get_parametric(), -1) for i in range(n_screens)]

            def rows_of(ids):
                rows, screen_vis = scene.visibility_with_faces(screens, row_ids=ids, strict=strict)
                place._screens_ahead = ((emitter.tobytes(), n_screens), screen_vis)
                return rows
        place.visibility_matrix = gather_visibility_rows(rows_of, len(polygons))
    return place.visibility_matrix


def solve_sharded(problem, order):
    """Solves this rank's block of receivers, then gives every rank every receiver's paths."""
    dist = _dist()
    n_rx = len(problem.receivers)
    world = dist.get_world_size() if dist else 1
    rank = dist.get_rank() if dist else 0
    begin, end = shard_bounds(n_rx, world, rank)
    mine = list(range(begin, end))
    if mine:
        problem.solve(max_order=order, receivers_indexs=mine)
    attempted = all_gather_rows(np.array([[float(problem.path_solves_attempted)]]), dist)
    if dist is not None:
        rows = all_gather_rows(_encode_paths(problem, mine), dist)
        for r in mine:  # replaced by the gathered copy below, so every rank holds the same containers
            problem.reflections[r].clear()
            problem.diffractions[r].clear()
        _decode_paths(problem, rows)
    if order >= 2:
        problem.solved_receivers = problem.receivers
    return float(attempted.sum())


def multithread_solve_place(place, tx, geometry, N_CPU=16, order=3, save_name=None):
    """Solves every receiver of ``place`` and computes the fields; returns ``(solved_em_path, solved_rays_path)``.
    ``N_CPU`` is accepted for compatibility: the degree of parallelism is the number of ranks of the process group."""
    if save_name is None:
        save_name = f"{geometry}_{order}_launch"
    dist = _dist()
    rank = dist.get_rank() if dist else 0
    t0 = time.time()
    if dist is not None and dist.get_world_size() > 1:
        # the one precompute step worth sharing out (every rank needs the whole matrix); the emitter's screens are
        # painted beside this rank's rows
        sharded_visibility_matrix(place, emitter=tx)
    problem = RayTracingProblem(tx, place)
    solve_sharded(problem, order)
    out_dir = electromagnetism_utils.results_dir()
    solved_rays_path = os.path.join(out_dir, f"{save_name}_ray_solved.json")
    solved_em_path = os.path.join(out_dir, f"{save_name}_em_solved.csv")
    if rank == 0:
        problem.to_json(solved_rays_path)
        my_field_computation(problem, solved_em_path)
        print(f"solved {len(problem.receivers)} receivers at order {order} in {time.time() - t0:.2f} s")
    if dist is not None:
        dist.barrier()
    return solved_em_path, solved_rays_path
