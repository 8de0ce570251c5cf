"""The receiver-sharding path under a real 2-rank process group (gloo, CPU): block bounds, variable-length all-gather,
path record round trip.  The solve itself needs a GPU and is covered by the gpu tests."""
import os
from collections import defaultdict

import numpy as np
import pytest


def test_shard_bounds_cover_everything():
    from raytracing_amd.multithread_solve import shard_bounds

    for n in (0, 1, 7, 16, 1024, 10000):
        for world in (1, 2, 3, 8):
            blocks = [shard_bounds(n, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in blocks]
            assert max(sizes) - min(sizes) <= 1


class FakeProblem:
    def __init__(self, n_rx):
        self.receivers = np.zeros((n_rx, 3))
        self.reflections = {r: defaultdict(list) for r in range(n_rx)}
        self.diffractions = {r: defaultdict(list) for r in range(n_rx)}


def make_paths(problem, rx_ids):
    for r in rx_ids:
        rng = np.random.default_rng(r)
        for k in range(r % 3):
            problem.reflections[r][1].append((rng.normal(size=(3, 3)), [int(rng.integers(0, 50))]))
        if r % 2:
            problem.reflections[r][2].append((rng.normal(size=(4, 3)), [3, 7]))
            problem.diffractions[r][2].append((rng.normal(size=(4, 3)), [5], (2, 9), rng.normal(size=(2, 3))))
        problem.diffractions[r][1].append((rng.normal(size=(3, 3)), [], (1, r + 2), rng.normal(size=(2, 3))))


def _worker(rank, world, port, n_rx, out_dir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from raytracing_amd import multithread_solve as mts

    begin, end = mts.shard_bounds(n_rx, world, rank)
    problem = FakeProblem(n_rx)
    make_paths(problem, range(begin, end))
    rows = mts.all_gather_rows(mts._encode_paths(problem, range(begin, end)))
    merged = FakeProblem(n_rx)
    mts._decode_paths(merged, rows)
    counts = mts.all_gather_rows(np.array([[float(end - begin)]]))
    np.save(os.path.join(out_dir, f"rows_{rank}.npy"), rows)
    ok = counts.sum() == n_rx
    want = FakeProblem(n_rx)
    make_paths(want, range(n_rx))
    for r in range(n_rx):
        for order in want.reflections[r]:
            a, b = merged.reflections[r][order], want.reflections[r][order]
            ok = ok and len(a) == len(b) and all(np.array_equal(x[0], y[0]) and x[1] == y[1] for x, y in zip(a, b))
        for order in want.diffractions[r]:
            a, b = merged.diffractions[r][order], want.diffractions[r][order]
            ok = ok and len(a) == len(b) and all(np.array_equal(x[0], y[0]) and x[1] == y[1] and x[2] == y[2] and
                                                 np.array_equal(x[3], y[3]) for x, y in zip(a, b))
    with open(os.path.join(out_dir, f"ok_{rank}"), "w") as f:
        f.write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_rx", [5, 16])
def test_two_rank_gather_of_path_records(tmp_path, n_rx):
    import torch.multiprocessing as mp

    port = 29500 + (os.getpid() % 2000) + n_rx
    mp.spawn(_worker, args=(2, port, n_rx, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "ok_0").read() == "1" and open(tmp_path / "ok_1").read() == "1"
    np.testing.assert_array_equal(np.load(tmp_path / "rows_0.npy"), np.load(tmp_path / "rows_1.npy"))


def test_single_process_gather_is_identity():
    from raytracing_amd.multithread_solve import all_gather_rows

    a = np.arange(12.0).reshape(4, 3)
    np.testing.assert_array_equal(all_gather_rows(a), a)


def _stub_rows(n):
    """A deterministic unsymmetrised 'sees' relation standing in for the GPU rows."""
    rng = np.random.default_rng(42)
    full = rng.random((n, n)) < 0.2
    np.fill_diagonal(full, False)
    return full


def _vis_worker(rank, world, port, n, out_dir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from raytracing_amd import multithread_solve as mts

    full = _stub_rows(n)
    asked = []

    def rows_of(ids):
        asked.extend(int(i) for i in ids)
        return full[ids]

    matrix = mts.gather_visibility_rows(rows_of, n)
    np.save(os.path.join(out_dir, f"vis_{rank}.npy"), matrix)
    np.save(os.path.join(out_dir, f"asked_{rank}.npy"), np.array(asked))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [1, 7, 64])
def test_two_rank_visibility_rows(tmp_path, n):
    """Rows dealt to two ranks (interleaved), all-gathered and symmetrised: every rank ends with rows | rows^T and each
    row was computed by exactly one rank."""
    import torch.multiprocessing as mp

    from raytracing_amd import multithread_solve as mts

    port = 31500 + (os.getpid() % 2000) + n
    mp.spawn(_vis_worker, args=(2, port, n, str(tmp_path)), nprocs=2, join=True)
    full = _stub_rows(n)
    want = full | full.T
    for r in range(2):
        np.testing.assert_array_equal(np.load(tmp_path / f"vis_{r}.npy"), want)
    asked = np.concatenate([np.load(tmp_path / f"asked_{r}.npy") for r in range(2)])
    assert sorted(asked.tolist()) == list(range(n))
    # without a process group: one rank does every row
    np.testing.assert_array_equal(mts.gather_visibility_rows(lambda ids: full[ids], n), want)
