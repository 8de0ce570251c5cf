"""The receiver-sharding path under a real 2-rank process group (gloo, CPU): block bounds, variable-length all-gather of
the per-path rows, all-gather of the fixed-size receiver records, failure propagation, visibility rows.  The solve
itself needs a GPU and is covered by the gpu tests."""
import os

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


def fake_block(begin, end):
    """Per-path arrays as PathBlock.to_host() gives them for receivers [begin, end): a receiver r has r % 4 paths."""
    from raytracing_amd import _lib

    rx = np.concatenate([np.full(r % 4, r - begin) for r in range(begin, end)] + [np.zeros(0)]).astype(np.int32)
    n = len(rx)
    rng = np.random.default_rng(begin * 1000 + end)
    return dict(rx=rx, group=rng.integers(-1, 4, n).astype(np.int32), cand=rng.integers(0, 1 << 30, n).astype(np.int32),
                points=rng.normal(size=(n, _lib.RT_MAX_ORDER, 3)), kind=rng.integers(0, 260, n).astype(np.int32),
                chain=rng.integers(-1, 7000, (n, _lib.RT_MAX_ORDER)).astype(np.int32),
                edge_polys=rng.integers(-1, 7000, (n, 2)).astype(np.int32), edge_points=rng.normal(size=(n, 6)),
                field=rng.normal(size=(n, 1, 6)), power=rng.random((n, 1)), angles=rng.normal(size=(n, 4)),
                time=rng.random(n), status=rng.integers(0, 11, n).astype(np.int32))


def check_rows(arrays, n_rx, world):
    """The gathered rows are the blocks of all ranks one after the other, receiver indices global."""
    from raytracing_amd import multithread_solve as mts

    ok, at = True, 0
    for r in range(world):
        begin, end = mts.shard_bounds(n_rx, world, r)
        want = fake_block(begin, end)
        k = len(want["rx"])
        ok = ok and np.array_equal(arrays["rx"][at:at + k], want["rx"] + begin)
        for name in ("group", "cand", "kind", "chain", "edge_polys", "status"):
            ok = ok and np.array_equal(arrays[name][at:at + k], want[name].reshape(arrays[name][at:at + k].shape))
        for name in ("points", "edge_points", "angles", "time"):
            ok = ok and np.array_equal(arrays[name][at:at + k], want[name])
        ok = ok and np.array_equal(arrays["power"][at:at + k], want["power"][:, 0])
        f = want["field"][:, 0]
        ok = ok and np.array_equal(arrays["field"][at:at + k], f[:, 0::2] + 1j * f[:, 1::2])
        at += k
    return ok and at == len(arrays["rx"])


def _worker(rank, world, port, n_rx, out_dir):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from raytracing_amd import device_pipeline as dp
    from raytracing_amd import multithread_solve as mts

    begin, end = mts.shard_bounds(n_rx, world, rank)
    rows = mts.all_gather_rows(mts.pack_path_rows(fake_block(begin, end), begin))
    ok = check_rows(mts.unpack_path_rows(rows), n_rx, world)
    counts = mts.all_gather_rows(np.array([[float(end - begin)]]))
    ok = ok and counts.sum() == n_rx
    np.save(os.path.join(out_dir, f"rows_{rank}.npy"), rows)
    # the fixed-size receiver records: every rank ends with the records of all receivers, in receiver order
    n_freq = 3
    width = dp.record_bytes(n_freq)
    local = np.zeros((end - begin, width), dtype=np.uint8)
    db = np.arange(begin, end, dtype=np.float64)[:, None] * 10.0 - np.arange(n_freq)[None, :]
    db[::3] = -np.inf
    ints = (np.arange(begin, end, dtype=np.int32)[:, None] * 100 + np.arange(dp.N_INT_COLUMNS, dtype=np.int32)[None, :])
    local[:, :8 * n_freq] = db.view(np.uint8).reshape(end - begin, -1)
    local[:, 8 * n_freq:8 * n_freq + 4 * dp.N_INT_COLUMNS] = np.ascontiguousarray(ints).view(np.uint8).reshape(end - begin, -1)
    gathered = dp.all_gather_records(torch.from_numpy(local), n_rx, dist)
    rec = dp.unpack_records(gathered.numpy(), n_freq)
    want_db = np.arange(n_rx, dtype=np.float64)[:, None] * 10.0 - np.arange(n_freq)[None, :]
    for r in range(world):
        b, _ = mts.shard_bounds(n_rx, world, r)
        want_db[b::3][: len(range(b, mts.shard_bounds(n_rx, world, r)[1], 3))] = -np.inf
    ok = ok and np.array_equal(rec["power_db"], want_db)
    ok = ok and np.array_equal(rec["counts"][:, 2], np.arange(n_rx) * 100 + 2)
    ok = ok and np.array_equal(rec["n_paths"], np.arange(n_rx) * 100 + dp.N_TYPES)
    ok = ok and np.array_equal(rec["worst_status"], np.arange(n_rx) * 100 + dp.N_TYPES + 2)
    with open(os.path.join(out_dir, f"ok_{rank}"), "w") as f:
        f.write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_rx", [5, 16])
def test_two_rank_gather_of_path_rows_and_receiver_records(tmp_path, n_rx):
    import torch.multiprocessing as mp

    port = 29500 + (os.getpid() % 2000) + n_rx
    mp.spawn(_worker, args=(2, port, n_rx, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "ok_0").read() == "1" and open(tmp_path / "ok_1").read() == "1"
    np.testing.assert_array_equal(np.load(tmp_path / "rows_0.npy"), np.load(tmp_path / "rows_1.npy"))


def _failing_worker(rank, world, port, out_dir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from raytracing_amd import multithread_solve as mts

    mts.raise_together(None)  # nobody failed: nobody raises
    outcome = "none"
    try:
        mts.raise_together(AssertionError("incident angle should be in 0,90 deg") if rank == 0 else None)
    except AssertionError as err:
        outcome = "own:" + str(err)
    except RuntimeError as err:
        outcome = "other:" + str(err)
    with open(os.path.join(out_dir, f"outcome_{rank}"), "w") as f:
        f.write(outcome)
    dist.barrier()  # reached by both: the failure did not leave anyone behind in a collective
    dist.destroy_process_group()


def test_a_failure_on_one_rank_raises_on_every_rank(tmp_path):
    """Rank 0 fails in its part (as when the field step meets one of the reference's physics assertions): it raises its
    own error, the other rank raises too instead of waiting at a barrier for ever."""
    import torch.multiprocessing as mp

    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_failing_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "outcome_0").read().startswith("own:incident angle")
    assert open(tmp_path / "outcome_1").read().startswith("other:") and "rank [0]" in open(tmp_path / "outcome_1").read()


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


@pytest.mark.parametrize("world,n_items", [(1, 7), (2, 7), (3, 7), (4, 8), (8, 7621), (5, 3)])
def test_dealt_blocks_come_back_in_order(world, n_items):
    """Items dealt out interleaved (rank r holds items r, r + W, ...) in zero-padded blocks of equal length, stacked as an
    all-gather leaves them: ``dealt_rows_in_order`` gives the items 0 .. n_items - 1 one after the other."""
    import torch

    from raytracing_amd import multithread_solve as mts

    n_cols = 3
    pad = (n_items + world - 1) // world
    blocks = torch.zeros((world, pad, n_cols), dtype=torch.int64)
    for r in range(world):
        ids = torch.from_numpy(np.arange(r, n_items, world))  # (empty for a rank beyond the last item)
        blocks[r, :len(ids)] = ids[:, None] * 10 + torch.arange(n_cols)[None, :]
    rows = mts.dealt_rows_in_order(blocks, n_items)
    want = torch.arange(n_items)[:, None] * 10 + torch.arange(n_cols)[None, :]
    assert rows.shape == (n_items, n_cols) and torch.equal(rows, want)
