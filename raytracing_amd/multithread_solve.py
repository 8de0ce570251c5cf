"""Receiver-parallel driver with the reference's entry point (``electromagnetism_fun/multithread_solve.py:34-112``).

The reference maps one receiver per worker process with ``pathos`` and pickles the whole place to each of them.  Here
the parallel axis is the same — receivers — but the workers are the GPUs of one node: one process per GPU under
``torch.distributed`` (RCCL), each holding the full scene.  A rank solves its contiguous block of receivers AND computes
the fields of that block's paths (the reference computes the fields serially in the parent, multithread_solve.py:97),
everything staying on its GPU (``device_pipeline.DevicePipeline``); the per-path rows — path and field results together —
are combined with all-gathers (counts first, then padded payloads) and rank 0 writes the reference's two result files.
A failure on rank 0 (a physics assertion of the field step, a file that cannot be written) is told to every rank before
anyone waits at a barrier, and every rank raises.  Without an initialised process group it runs on one GPU.
"""
import os
import time

import numpy as np

from . import _lib
from . import electromagnetism as em
from . import electromagnetism_utils
from . import geometry as geom
from . import materials_properties as mat
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


def gather_dealt_rows(rows_of, n_rows, n_cols, dist=None):
    """Byte rows computed rank by rank, on every rank: rank r computes rows r, r + W, r + 2W ... (``rows_of(ids)`` ->
    uint8 / bool [len(ids), n_cols]; interleaved because neighbouring rows cost alike), one all-gather of the padded
    blocks puts them back in order.  Returns bool [n_rows, n_cols]."""
    return _gather_dealt_rows_tensor(rows_of, n_rows, n_cols, dist or _dist()).cpu().numpy().astype(bool)


def _gather_dealt_rows_tensor(rows_of, n_rows, n_cols, dist):
    """``gather_dealt_rows`` as a uint8 torch tensor [n_rows, n_cols] on the device the collective works on (the GPU
    for RCCL, the host for gloo or a single rank).  Rank r's k-th row is row k * W + r of the whole: the gathered
    blocks [W, pad, n_cols] with their first two axes swapped are the rows in order."""
    import torch

    world = dist.get_world_size() if dist else 1
    rank = dist.get_rank() if dist else 0
    mine = np.arange(rank, n_rows, world, dtype=np.int32)
    local = np.ascontiguousarray(rows_of(mine), dtype=np.uint8).reshape(len(mine), n_cols)
    if dist is None:
        return torch.from_numpy(local)
    dev = _device_for_collectives(dist)
    pad = (n_rows + world - 1) // world
    buf = torch.zeros((pad, n_cols), dtype=torch.uint8, device=dev)
    buf[:len(mine)] = torch.from_numpy(local).to(dev)
    gathered = torch.empty((world, pad, n_cols), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(gathered.view(world * pad, n_cols), buf)
    return dealt_rows_in_order(gathered, n_rows)


def gather_visibility_rows(rows_of, n, dist=None):
    """Visibility matrix of an n-polygon scene from rows computed rank by rank (``gather_dealt_rows``), then
    matrix = rows | rows^T (geometry.py:1036-1048)."""
    rows = _gather_dealt_rows_tensor(rows_of, n, n, dist or _dist())
    return (rows | rows.T).cpu().numpy().astype(bool)


def paint_dealt_block_on_device(scene, screens, strict, world, rank):
    """This rank's share of the list [matrix rows 0..N-1, screens] dealt out interleaved (items rank, rank + W, ...),
    painted into one zero-padded uint8 tensor [ceil((N + len(screens)) / W), N] on the scene's GPU: its matrix rows
    first, then its screens (the screens are the last items of the list), which is the order the items have."""
    import torch

    n = scene.n
    n_items = n + len(screens)
    ids = np.arange(rank, n_items, world, dtype=np.int64)
    row_ids = ids[ids < n].astype(np.int32)
    mine = [screens[int(i) - n] for i in ids[ids >= n]]
    pad = (n_items + world - 1) // world
    dev = torch.device("cuda", scene.device)
    block = torch.zeros((pad, n), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)  # the library writes on a stream of its own
    scene.visibility_with_faces(mine, row_ids=row_ids, strict=strict, rows_out=block[:len(row_ids)],
                                faces_out=block[len(row_ids):len(ids)])
    return block


def dealt_rows_in_order(gathered, n_items):
    """The items of the dealt list in order, from the gathered blocks [W, pad, n_cols]: rank r's k-th item is item
    k * W + r, so the blocks with their first two axes swapped are the items one after the other."""
    world, pad, n_cols = gathered.shape
    return gathered.permute(1, 0, 2).reshape(world * pad, n_cols)[:n_items]


def sharded_visibility(scene, screens, strict=False, dist=None, on_device=False):
    """The visibility matrix of ``scene`` and the visibility vectors of the emitter's ``screens`` (as
    ``Scene.faces_visibility`` takes them) with the work dealt to the ranks: the matrix rows AND the screens are rows
    of one list, dealt out interleaved, each rank painting its matrix rows and its screens side by side; one
    all-gather.  A screen of a large scene costs as much as hundreds of matrix rows, so the screens must not be
    repeated on every rank.  Returns (matrix bool [N, N], screen vectors bool [len(screens), N]); with ``on_device`` the
    matrix is a byte tensor on the scene's GPU instead, which ``Scene.enumerate`` reads there (the matrix of a large
    scene takes longer to bring to the host than a sixth of its painting, and nothing on the host needs it)."""
    n = scene.n
    dist = dist or _dist()
    # (a group of one rank over RCCL goes through the collective all the same: a rehearsal of that branch on one GPU)
    if dist is None or (dist.get_world_size() == 1 and dist.get_backend() != "nccl"):
        # one rank paints everything: matrix (symmetrised on the device) and screens side by side
        if on_device:
            import torch

            matrix = torch.empty((n, n), dtype=torch.uint8, device=torch.device("cuda", scene.device))
            return scene.visibility_with_faces(screens, strict=strict, matrix_out=matrix)
        return scene.visibility_with_faces(screens, strict=strict)

    def rows_of(ids):
        ids = np.asarray(ids, dtype=np.int64)
        row_ids = ids[ids < n].astype(np.int32)
        mine = [screens[int(i) - n] for i in ids[ids >= n]]
        if mine:
            rows, faces = scene.visibility_with_faces(mine, row_ids=row_ids, strict=strict)
        else:
            rows, faces = scene.visibility_rows(row_ids, strict), np.zeros((0, n), dtype=bool)
        out = np.zeros((len(ids), n), dtype=np.uint8)
        out[ids < n] = rows
        out[ids >= n] = faces
        return out

    # put in order and symmetrised where the collective left the rows (on the GPU for RCCL: the host took 0.2 s to
    # scatter 58 MB of interleaved rows and OR them with their transpose, more than an eighth of the painting takes)
    if dist.get_backend() == "nccl":
        # painted straight into the buffer the collective sends: the rows never visit the host
        import torch

        world = dist.get_world_size()
        block = paint_dealt_block_on_device(scene, screens, strict, world, dist.get_rank())
        gathered = torch.empty((world,) + tuple(block.shape), dtype=torch.uint8, device=block.device)
        dist.all_gather_into_tensor(gathered.view(world * block.shape[0], n), block)
        rows = dealt_rows_in_order(gathered, n + len(screens))
    else:
        rows = _gather_dealt_rows_tensor(rows_of, n + len(screens), n, dist)
    matrix = rows[:n]
    matrix = matrix | matrix.T
    if on_device:
        import torch

        matrix = matrix.to(torch.device("cuda", scene.device)).contiguous()
        torch.cuda.synchronize(matrix.device)  # the library reads it on a stream of its own
        return matrix, rows[n:].cpu().numpy().astype(bool)
    return matrix.cpu().numpy().astype(bool), rows[n:].cpu().numpy().astype(bool)


def sharded_visibility_matrix(place, strict=False, emitter=None, n_screens=6):
    """``place.get_visibility_matrix()`` with the rows dealt to the ranks of the process group; the result is cached on
    the place like the reference's (geometry.py:1467-1483), on every rank.  With ``emitter`` the emitter's screens
    (ray_tracing.py:210-228) are dealt out with the rows and left on the place for the problem's ``precompute``."""
    if place.visibility_matrix is None:
        polygons = place.get_polygons_list()
        scene = geom.scene_arrays_of(polygons).device_scene()
        if emitter is None:
            place.visibility_matrix = gather_visibility_rows(lambda ids: scene.visibility_rows(ids, strict), len(polygons))
        else:
            emitter = np.asarray(emitter, dtype=float)
            cube = geom.Cube.by_point_and_side_length(emitter, 2 * 0.1)
            screens = [(cube.polygons[i].points, cube.polygons[i].get_parametric(), -1) for i in range(n_screens)]
            place.visibility_matrix, screen_vis = sharded_visibility(scene, screens, strict)
            place._screens_ahead = ((emitter.tobytes(), n_screens), screen_vis)
    return place.visibility_matrix


# columns of a path row (float64, every integer exactly representable): what a rank contributes per path
_COLS = dict(rx=(0, 1), group=(1, 2), cand=(2, 3), points=(3, 15), kind=(15, 16), chain=(16, 20), edge_polys=(20, 22),
             edge_points=(22, 28), field=(28, 34), power=(34, 35), angles=(35, 39), time=(39, 40), status=(40, 41))
PATH_ROW_WIDTH = 41


def pack_path_rows(host, rx_offset):
    """Per-path arrays of one rank's block (``PathBlock.to_host()``, one frequency) as one float64 matrix; receiver
    indices become global."""
    n = len(host["rx"])
    rows = np.zeros((n, PATH_ROW_WIDTH))
    for name, (a, b) in _COLS.items():
        rows[:, a:b] = np.asarray(host[name], dtype=np.float64).reshape(n, b - a)
    rows[:, 0] += rx_offset
    return rows


def unpack_path_rows(rows):
    rows = np.asarray(rows, dtype=np.float64).reshape(-1, PATH_ROW_WIDTH)
    out = {}
    for name, (a, b) in _COLS.items():
        v = rows[:, a:b]
        if name in ("rx", "group", "cand", "kind", "chain", "edge_polys", "status"):
            v = np.rint(v).astype(np.int64)
        out[name] = v[:, 0] if b - a == 1 else v
    out["points"] = out["points"].reshape(-1, _lib.RT_MAX_ORDER, 3)
    field = out["field"]
    out["field"] = field[:, 0::2] + 1j * field[:, 1::2]
    return out


def solve_block_on_device(problem, order, begin, end, frequency=None):
    """Lines of sight, solve and field step for receivers [begin, end) of ``problem`` on this rank's GPU: returns
    (path rows of the block, candidates handle, path-solves attempted).  The place gets its tree crowns first, as
    my_field_computation gives them (electromagnetism.py:745); they only attenuate."""
    import torch

    from .device_pipeline import DevicePipeline

    scene = problem._solving_scene()
    cand = scene.enumerate(problem.visibility_matrix, problem.emitter_visibility, problem._edge_pairs,
                           problem._edge_array, order, with_diffraction=True)
    problem.place.add_tree_crowns()
    all_polygons = problem.place.get_polygons_list()
    n_late = len(all_polygons) - len(problem.polygons)
    field_scene = geom.scene_arrays_of(all_polygons).device_scene() if n_late > 0 else scene
    tx = np.asarray(problem.emitter, dtype=float).reshape(3)
    pipe = DevicePipeline(scene, cand, tx, capacity=max(4096, 64 * (end - begin)), field_scene=field_scene,
                          n_late_polygons=max(n_late, 0))
    receivers = np.asarray(problem.receivers, dtype=float).reshape(-1, 3)
    d_rx = torch.tensor(receivers[begin:end], dtype=torch.float64, device=pipe.device).reshape(-1, 3)
    block = pipe.run(d_rx, [mat.FREQUENCY if frequency is None else frequency], tx_gain=mat.TX_GAIN, rx_gain=mat.RX_GAIN,
                     p_in=mat.P_IN, isotropic=mat.ISOTROPIC_ANTENNA, alpha=mat.ALPHA)
    rows = pack_path_rows(block.to_host(), begin)
    return rows, cand, block.n_attempted


def fill_problem_and_table(problem, arrays, cand, order):
    """Rank 0's part: the reference's containers (ray_tracing.py:314, 350, 376) and result table
    (electromagnetism.py:577-607) from the gathered rows, which are sorted by (receiver, group, candidate)."""
    paths = arrays["group"] >= 0
    problem.store_paths(arrays["rx"][paths], arrays["group"][paths], arrays["cand"][paths], arrays["points"][paths], cand)
    if order >= 2:
        problem.solved_receivers = problem.receivers
    if problem.solved_receivers is None:  # my_field_computation iterates over it (electromagnetism.py:763)
        raise TypeError("object of type 'NoneType' has no len(): solve(max_order >= 2) must run first")
    labels = em.path_labels(arrays["kind"])
    em._raise_on_status(arrays["status"], labels)
    receivers = np.asarray(problem.receivers, dtype=float).reshape(-1, 3)
    return em.table_from_columns(len(receivers), receivers, arrays["rx"], labels, arrays["angles"], arrays["time"],
                                 arrays["field"], arrays["power"])


def raise_together(error, dist=None):
    """Every rank learns whether any rank failed; when one did, every rank raises (the failing rank its own error).
    Called before anything that waits for the other ranks, so that a failure on one rank cannot leave the others
    waiting in a collective."""
    dist = dist or _dist()
    if dist is None:
        if error is not None:
            raise error
        return
    flags = all_gather_rows(np.array([[0.0 if error is None else 1.0]]), dist)
    if error is not None:
        raise error
    if flags.any():
        bad = [int(r) for r in np.flatnonzero(flags[:, 0])]
        raise RuntimeError(f"multithread_solve_place failed on rank {bad}; see that rank's error")


def multithread_solve_place(place, tx, geometry, N_CPU=16, order=3, save_name=None):
    """Solves every receiver of ``place`` and computes the fields; returns ``(solved_em_path, solved_rays_path)``.
    ``N_CPU`` is accepted for compatibility: the degree of parallelism is the number of ranks of the process group."""
    if save_name is None:
        save_name = f"{geometry}_{order}_launch"
    dist = _dist()
    rank = dist.get_rank() if dist else 0
    world = dist.get_world_size() if dist else 1
    t0 = time.time()
    out_dir = electromagnetism_utils.results_dir()
    solved_rays_path = os.path.join(out_dir, f"{save_name}_ray_solved.json")
    solved_em_path = os.path.join(out_dir, f"{save_name}_em_solved.csv")
    error, rows, cand, problem = None, np.zeros((0, PATH_ROW_WIDTH)), None, None
    try:
        if world > 1:
            # the one precompute step worth sharing out (every rank needs the whole matrix); the emitter's screens are
            # painted beside this rank's rows
            sharded_visibility_matrix(place, emitter=tx)
    except Exception as err:  # noqa: BLE001 - told to the other ranks below
        error = err
    raise_together(error, dist)  # the matrix gather is a collective: nobody goes on alone
    try:
        problem = RayTracingProblem(tx, place)
        begin, end = shard_bounds(len(problem.receivers), world, rank)
        if order >= 1:
            rows, cand, attempted = solve_block_on_device(problem, order, begin, end)
            problem.path_solves_attempted += attempted
    except Exception as err:  # noqa: BLE001
        error = err
    raise_together(error, dist)
    rows = all_gather_rows(rows, dist)  # blocks are contiguous and ascending: the rows stay sorted by receiver
    try:
        if rank == 0:
            if cand is not None:
                df = fill_problem_and_table(problem, unpack_path_rows(rows), cand, order)
            else:
                raise TypeError("object of type 'NoneType' has no len(): solve(max_order >= 2) must run first")
            problem.to_json(solved_rays_path)
            electromagnetism_utils.save_df(df, solved_em_path)
            print(f"solved {len(problem.receivers)} receivers at order {order} in {time.time() - t0:.2f} s")
    except Exception as err:  # noqa: BLE001
        error = err
    finally:
        if cand is not None:
            cand.close()
    raise_together(error, dist)
    return solved_em_path, solved_rays_path
